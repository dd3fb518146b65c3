/*
 * gs_oracle.c -- CPU restatement of the 3D-Gaussian-Splatting rasterizer GSWorld renders through.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product path (gsworld_amd/) never does.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in third-party CUDA code that is NOT under
 * /root/reference (empty submodule, /root/reference/.gitmodules:1-3 ->
 * graphdeco-inria/gaussian-splatting -> diff-gaussian-rasterization (branch dr_aa), simple-knn,
 * fused-ssim; no commit pins recoverable; GSWorld's only edit is the near-plane constant,
 * /root/reference/README.md:33 "from 0.2f to 0.05f in cuda_rasterizer/auxiliary.h").
 * The reference has no tests or golden vectors for this path.  What follows
 * restates the *published* algorithm of those packages, by upstream file name:
 *   cuda_rasterizer/auxiliary.h   in_frustum, getRect, ndc2Pix, transformPoint4x3/4x4, SH constants
 *   cuda_rasterizer/forward.cu    computeCov3D, computeCov2D, computeColorFromSH, preprocessCUDA, renderCUDA
 *   cuda_rasterizer/rasterizer_impl.cu  getHigherMsb, duplicateWithKeys, identifyTileRanges, sort range
 *   cuda_rasterizer/backward.cu   renderCUDA (bwd), computeCov2DCUDA, preprocessCUDA (bwd)
 *   simple-knn/simple_knn.cu      distCUDA2  (exact 3-NN mean squared distance)
 *   fused-ssim/ssim.cu            fusedssim / fusedssim_backward
 * and is anchored on the reference's call sites:
 *   /root/reference/gsworld/mani_skill/utils/wrappers/gs_world_wrapper.py:232-275 (render call, bg=0,
 *   uint8 conversion) and :277-325 (camera construction).
 *
 * Canonical floating-point order.  Where bit-exactness matters (depth keys, radii, tile rects) the
 * result depends on FMA contraction choices nvcc made, which cannot be known here.  This file therefore
 * DEFINES the canonical order: every fused multiply-add is an explicit fmaf(), everything else is a
 * single IEEE-754 binary32 operation, and the file must be compiled with -ffp-contract=off.  The HIP
 * kernels follow the same order and are tested bit-for-bit against this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Exposure builds (oracle/Makefile `variants`, tools/fma_exposure.py): the SAME file with the
 * contraction choice changed, to MEASURE how much of the output depends on it.  GSO_FMA_NONE: nothing fuses (nvcc
 * -fmad=false); GSO_FMA_COMPILER: every a * b + c is left to the compiler under -ffp-contract=fast (the closest thing
 * here to "whatever nvcc -fmad=true picked").  Neither is the canonical build and no test uses them as a checker. */
#if defined(GSO_FMA_NONE) || defined(GSO_FMA_COMPILER)
#undef fmaf
#define fmaf(a, b, c) ((a) * (b) + (c))
#endif

#define GSO_BLOCK_X 16
#define GSO_BLOCK_Y 16
#define GSO_NEAR_DEFAULT 0.05f /* /root/reference/README.md:33 (stock upstream: 0.2f) */

typedef struct {
    int32_t image_height, image_width;
    float tanfovx, tanfovy;
    float scale_modifier;
    int32_t sh_degree;   /* active degree D (0..3) */
    int32_t sh_coeffs;   /* M: coefficients stored per Gaussian (16 for GSWorld) */
    int32_t prefiltered;
    int32_t antialiasing;
    float near_plane;    /* 0.05f for GSWorld */
} GsoSettings;

static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

void gso_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int gso_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* rasterizer_impl.cu getHigherMsb: smallest msb with (n >> msb) == 0, found by bisection from 16. */
uint32_t gso_higher_msb(uint32_t n) {
    uint32_t msb = 32u / 2u, step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

/* auxiliary.h transformPoint4x3 / 4x4; matrices are stored so that M[r][c] = m[c*4+r]. */
static inline void xform4x3(const float *m, float px, float py, float pz, float *o) {
    o[0] = fmaf(m[8], pz, fmaf(m[4], py, m[0] * px)) + m[12];
    o[1] = fmaf(m[9], pz, fmaf(m[5], py, m[1] * px)) + m[13];
    o[2] = fmaf(m[10], pz, fmaf(m[6], py, m[2] * px)) + m[14];
}
static inline void xform4x4(const float *m, float px, float py, float pz, float *o) {
    xform4x3(m, px, py, pz, o);
    o[3] = fmaf(m[11], pz, fmaf(m[7], py, m[3] * px)) + m[15];
}

/* auxiliary.h ndc2Pix: the upstream expression mixes double literals with a float argument, so the
 * arithmetic is binary64 and only the result is rounded to float. */
static inline float ndc2pix(float v, int S) {
    return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5);
}

/* forward.cu computeCov3D: Sigma = R diag((mod*s)^2) R^T, quaternion (r,x,y,z) used as given. */
static void cov3d_from_scale_rot(const float *scale, float mod, const float *q, float *cov6) {
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    float R[3][3];
    R[0][0] = fmaf(-2.f, fmaf(z, z, y * y), 1.f);
    R[0][1] = 2.f * fmaf(-r, z, x * y);
    R[0][2] = 2.f * fmaf(r, y, x * z);
    R[1][0] = 2.f * fmaf(r, z, x * y);
    R[1][1] = fmaf(-2.f, fmaf(z, z, x * x), 1.f);
    R[1][2] = 2.f * fmaf(-r, x, y * z);
    R[2][0] = 2.f * fmaf(-r, y, x * z);
    R[2][1] = 2.f * fmaf(r, x, y * z);
    R[2][2] = fmaf(-2.f, fmaf(y, y, x * x), 1.f);
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    float M[3][3]; /* M[k][j] = s_k * R[j][k]  (S * R^T) */
    for (int k = 0; k < 3; k++)
        for (int j = 0; j < 3; j++) M[k][j] = s[k] * R[j][k];
#define SIG(i, j) fmaf(M[2][i], M[2][j], fmaf(M[1][i], M[1][j], M[0][i] * M[0][j]))
    cov6[0] = SIG(0, 0); cov6[1] = SIG(0, 1); cov6[2] = SIG(0, 2);
    cov6[3] = SIG(1, 1); cov6[4] = SIG(1, 2); cov6[5] = SIG(2, 2);
#undef SIG
}

/* forward.cu computeCov2D (EWA splatting): cov = (J W) Sigma (J W)^T, returns xx, xy, yy. */
static void cov2d_ewa(const float *t_in, float fx, float fy, float tanfovx, float tanfovy,
                      const float *c6, const float *view, float *out3) {
    float tx = t_in[0], ty = t_in[1];
    const float tz = t_in[2];
    const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
    const float txtz = tx / tz, tytz = ty / tz;
    tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
    ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
    const float J00 = fx / tz, J02 = -(fx * tx) / (tz * tz);
    const float J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
    float A[2][3];
    for (int j = 0; j < 3; j++) {
        /* W[i][j] = view[j*4+i] */
        A[0][j] = fmaf(J02, view[j * 4 + 2], J00 * view[j * 4 + 0]);
        A[1][j] = fmaf(J12, view[j * 4 + 2], J11 * view[j * 4 + 1]);
    }
    const float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    float B[2][3];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++)
            B[i][j] = fmaf(A[i][2], S[2][j], fmaf(A[i][1], S[1][j], A[i][0] * S[0][j]));
    out3[0] = fmaf(B[0][2], A[0][2], fmaf(B[0][1], A[0][1], B[0][0] * A[0][0]));
    out3[1] = fmaf(B[0][2], A[1][2], fmaf(B[0][1], A[1][1], B[0][0] * A[1][0]));
    out3[2] = fmaf(B[1][2], A[1][2], fmaf(B[1][1], A[1][1], B[1][0] * A[1][0]));
}

/* SH basis for a unit direction: b[0..(deg+1)^2).  forward.cu computeColorFromSH coefficient signs. */
static void sh_basis(int deg, float x, float y, float z, float *b) {
    b[0] = SH_C0;
    if (deg > 0) {
        b[1] = -(SH_C1 * y);
        b[2] = SH_C1 * z;
        b[3] = -(SH_C1 * x);
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = SH_C2[0] * xy;
            b[5] = SH_C2[1] * yz;
            b[6] = SH_C2[2] * (fmaf(2.f, zz, -xx) - yy);
            b[7] = SH_C2[3] * xz;
            b[8] = SH_C2[4] * (xx - yy);
            if (deg > 2) {
                b[9] = (SH_C3[0] * y) * fmaf(3.f, xx, -yy);
                b[10] = (SH_C3[1] * xy) * z;
                b[11] = (SH_C3[2] * y) * (fmaf(4.f, zz, -xx) - yy);
                b[12] = (SH_C3[3] * z) * fmaf(-3.f, yy, fmaf(-3.f, xx, 2.f * zz));
                b[13] = (SH_C3[4] * x) * (fmaf(4.f, zz, -xx) - yy);
                b[14] = (SH_C3[5] * z) * (xx - yy);
                b[15] = (SH_C3[6] * x) * fmaf(-3.f, yy, xx);
            }
        }
    }
}

/* forward.cu preprocessCUDA (forward).  All per-Gaussian outputs are written for every index; culled
 * Gaussians get radii = 0 and tiles_touched = 0 (other fields are left as passed in / zero).
 * rects: (min.x, min.y, max.x, max.y) per Gaussian, the getRect result (int32). */
/* float -> int32 as a GPU converts it (CUDA cvt.rzi.s32.f32, AMD v_cvt_i32_f32): truncation, SATURATING at the
 * int32 range, NaN -> 0.  A C cast is undefined out of range (x86 gives INT_MIN); absurdly large splats -- radius
 * beyond 2^31 pixels -- must still take the same rect on both sides. */
static inline int32_t f2i_sat(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int32_t)x;
}

void gso_preprocess(const GsoSettings *st, int P, const float *means3D, const float *shs,
                    const float *colors_precomp, const float *opacities, const float *scales,
                    const float *rotations, const float *cov3D_precomp, const float *viewmatrix,
                    const float *projmatrix, const float *campos,
                    /* outputs */ float *depths, int32_t *radii, float *means2D, float *cov3D,
                    float *conic_opacity, float *rgb, uint8_t *clamped, uint32_t *tiles_touched,
                    int32_t *rects) {
    const int W = st->image_width, H = st->image_height;
    const int gx = (W + GSO_BLOCK_X - 1) / GSO_BLOCK_X, gy = (H + GSO_BLOCK_Y - 1) / GSO_BLOCK_Y;
    const float fx = (float)W / (2.0f * st->tanfovx), fy = (float)H / (2.0f * st->tanfovy);
    const int D = st->sh_degree, M = st->sh_coeffs;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        radii[i] = 0;
        tiles_touched[i] = 0;
        rects[4 * i + 0] = rects[4 * i + 1] = rects[4 * i + 2] = rects[4 * i + 3] = 0;
        const float px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
        float pv[3];
        xform4x3(viewmatrix, px, py, pz, pv);
        if (pv[2] <= st->near_plane) continue; /* in_frustum (prefiltered trap not restated) */
        float ph[4];
        xform4x4(projmatrix, px, py, pz, ph);
        const float p_w = 1.0f / (ph[3] + 0.0000001f);
        const float pprx = ph[0] * p_w, ppry = ph[1] * p_w;
        float c6_local[6];
        const float *c6;
        if (cov3D_precomp) {
            c6 = cov3D_precomp + 6 * (size_t)i;
        } else {
            cov3d_from_scale_rot(scales + 3 * (size_t)i, st->scale_modifier, rotations + 4 * (size_t)i,
                                 c6_local);
            c6 = c6_local;
        }
        for (int k = 0; k < 6; k++) cov3D[6 * (size_t)i + k] = c6[k];
        float cov[3];
        cov2d_ewa(pv, fx, fy, st->tanfovx, st->tanfovy, c6, viewmatrix, cov);
        const float h_var = 0.3f;
        const float det_cov = fmaf(-cov[1], cov[1], cov[0] * cov[2]);
        cov[0] += h_var;
        cov[2] += h_var;
        const float det = fmaf(-cov[1], cov[1], cov[0] * cov[2]);
        float h_scale = 1.0f;
        if (st->antialiasing) h_scale = sqrtf(fmaxf(0.000025f, det_cov / det));
        if (det == 0.0f) continue;
        const float det_inv = 1.f / det;
        const float conic[3] = {cov[2] * det_inv, -cov[1] * det_inv, cov[0] * det_inv};
        const float mid = 0.5f * (cov[0] + cov[2]);
        const float root = sqrtf(fmaxf(0.1f, fmaf(mid, mid, -det)));
        const float lambda1 = mid + root, lambda2 = mid - root;
        const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        const float pix_x = ndc2pix(pprx, W), pix_y = ndc2pix(ppry, H);
        const int ir = f2i_sat(my_radius); /* getRect takes the radius as int */
        int rminx = f2i_sat((pix_x - (float)ir) / (float)GSO_BLOCK_X);
        int rminy = f2i_sat((pix_y - (float)ir) / (float)GSO_BLOCK_Y);
        int rmaxx = f2i_sat((pix_x + (float)ir + (float)(GSO_BLOCK_X - 1)) / (float)GSO_BLOCK_X);
        int rmaxy = f2i_sat((pix_y + (float)ir + (float)(GSO_BLOCK_Y - 1)) / (float)GSO_BLOCK_Y);
        rminx = rminx < 0 ? 0 : (rminx > gx ? gx : rminx);
        rminy = rminy < 0 ? 0 : (rminy > gy ? gy : rminy);
        rmaxx = rmaxx < 0 ? 0 : (rmaxx > gx ? gx : rmaxx);
        rmaxy = rmaxy < 0 ? 0 : (rmaxy > gy ? gy : rmaxy);
        if ((rmaxx - rminx) * (rmaxy - rminy) == 0) continue;
        if (colors_precomp) {
            for (int ch = 0; ch < 3; ch++) {
                rgb[3 * (size_t)i + ch] = colors_precomp[3 * (size_t)i + ch];
                clamped[3 * (size_t)i + ch] = 0;
            }
        } else {
            float dx = px - campos[0], dy = py - campos[1], dz = pz - campos[2];
            const float len = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
            dx = dx / len; dy = dy / len; dz = dz / len;
            float b[16];
            sh_basis(D, dx, dy, dz, b);
            const int nb = (D + 1) * (D + 1);
            const float *sh = shs + (size_t)i * M * 3;
            for (int ch = 0; ch < 3; ch++) {
                float c = b[0] * sh[ch];
                for (int k = 1; k < nb; k++) c = fmaf(b[k], sh[3 * k + ch], c);
                c += 0.5f;
                clamped[3 * (size_t)i + ch] = (c < 0.f);
                rgb[3 * (size_t)i + ch] = fmaxf(c, 0.f);
            }
        }
        depths[i] = pv[2];
        radii[i] = ir;
        means2D[2 * (size_t)i] = pix_x;
        means2D[2 * (size_t)i + 1] = pix_y;
        conic_opacity[4 * (size_t)i + 0] = conic[0];
        conic_opacity[4 * (size_t)i + 1] = conic[1];
        conic_opacity[4 * (size_t)i + 2] = conic[2];
        conic_opacity[4 * (size_t)i + 3] = opacities[i] * h_scale;
        tiles_touched[i] = (uint32_t)((rmaxy - rminy) * (rmaxx - rminx));
        rects[4 * i + 0] = rminx; rects[4 * i + 1] = rminy;
        rects[4 * i + 2] = rmaxx; rects[4 * i + 3] = rmaxy;
    }
}

/* auxiliary.h in_frustum via rasterizer_impl.cu checkFrustum (markVisible). */
void gso_mark_visible(int P, const float *means3D, const float *viewmatrix, float near_plane,
                      uint8_t *present) {
    for (int i = 0; i < P; i++) {
        float pv[3];
        xform4x3(viewmatrix, means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2], pv);
        present[i] = pv[2] > near_plane;
    }
}

/* rasterizer_impl.cu: InclusiveSum over tiles_touched (uint32); returns num_rendered. */
int64_t gso_inclusive_sum(int P, const uint32_t *tiles_touched, uint32_t *offsets) {
    uint32_t acc = 0;
    for (int i = 0; i < P; i++) {
        acc += tiles_touched[i];
        offsets[i] = acc;
    }
    return P ? (int64_t)acc : 0;
}

/* stable LSD radix sort of (u64 key, u32 value) on the low `bits` bits -- what SortPairs does. */
static void radix_sort_pairs(uint64_t *keys, uint32_t *vals, int64_t n, int bits) {
    if (n <= 1) return;
    uint64_t *k2 = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n);
    uint32_t *v2 = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n);
    uint64_t *ks = keys, *kd = k2;
    uint32_t *vs = vals, *vd = v2;
    for (int shift = 0; shift < bits; shift += 8) {
        int64_t hist[257];
        memset(hist, 0, sizeof(hist));
        const int nb = (bits - shift) < 8 ? (bits - shift) : 8;
        const uint64_t mask = ((uint64_t)1 << nb) - 1;
        for (int64_t i = 0; i < n; i++) hist[((ks[i] >> shift) & mask) + 1]++;
        for (int d = 0; d < 256; d++) hist[d + 1] += hist[d];
        for (int64_t i = 0; i < n; i++) {
            const int64_t p = hist[(ks[i] >> shift) & mask]++;
            kd[p] = ks[i];
            vd[p] = vs[i];
        }
        uint64_t *tk = ks; ks = kd; kd = tk;
        uint32_t *tv = vs; vs = vd; vd = tv;
    }
    if (ks != keys) {
        memcpy(keys, ks, sizeof(uint64_t) * (size_t)n);
        memcpy(vals, vs, sizeof(uint32_t) * (size_t)n);
    }
    free(k2);
    free(v2);
}

/* rasterizer_impl.cu duplicateWithKeys + SortPairs + identifyTileRanges.
 * keys/values have num_rendered entries; ranges has 2*tiles entries (start,end), zero when empty.
 * keys_unsorted (optional, may be NULL) receives the emission-order keys. */
void gso_bin(const GsoSettings *st, int P, const float *depths, const int32_t *radii,
             const int32_t *rects, const uint32_t *offsets, int64_t num_rendered,
             uint64_t *keys, uint32_t *values, uint64_t *keys_unsorted, uint32_t *ranges) {
    const int W = st->image_width, H = st->image_height;
    const int gx = (W + GSO_BLOCK_X - 1) / GSO_BLOCK_X, gy = (H + GSO_BLOCK_Y - 1) / GSO_BLOCK_Y;
    for (int i = 0; i < P; i++) {
        if (radii[i] <= 0) continue;
        uint32_t off = (i == 0) ? 0 : offsets[i - 1];
        uint32_t dbits;
        memcpy(&dbits, &depths[i], 4);
        for (int y = rects[4 * i + 1]; y < rects[4 * i + 3]; y++)
            for (int x = rects[4 * i + 0]; x < rects[4 * i + 2]; x++) {
                uint64_t key = (uint64_t)(y * gx + x);
                key <<= 32;
                key |= dbits;
                keys[off] = key;
                values[off] = (uint32_t)i;
                off++;
            }
    }
    if (keys_unsorted) memcpy(keys_unsorted, keys, sizeof(uint64_t) * (size_t)num_rendered);
    const int bits = 32 + (int)gso_higher_msb((uint32_t)(gx * gy));
    radix_sort_pairs(keys, values, num_rendered, bits);
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)(gx * gy));
    for (int64_t i = 0; i < num_rendered; i++) {
        const uint32_t tile = (uint32_t)(keys[i] >> 32);
        if (i == 0) ranges[2 * tile] = 0;
        else {
            const uint32_t prev = (uint32_t)(keys[i - 1] >> 32);
            if (tile != prev) {
                ranges[2 * prev + 1] = (uint32_t)i;
                ranges[2 * tile] = (uint32_t)i;
            }
        }
        if (i == num_rendered - 1) ranges[2 * tile + 1] = (uint32_t)num_rendered;
    }
}

/* forward.cu renderCUDA: front-to-back alpha compositing per pixel.  The tile-cooperative batching of
 * the CUDA kernel has no arithmetic effect, so each pixel simply walks its tile's range.
 * exp() is libm expf.  The borderline map (optional) counts, per pixel, decisions that sit within a relative
 * band of a threshold (`border_eps` around alpha = 1/255, `border_eps_T` around T = 1e-4) and could therefore
 * flip under an exp() that differs in the last ulps (the GPU uses the hardware exp2 unit). */
void gso_render(const GsoSettings *st, const uint32_t *ranges, const uint32_t *point_list,
                const float *means2D, const float *conic_opacity, const float *rgb,
                const float *depths, const float *bg, float *out_color, float *out_invdepth,
                float *final_T, uint32_t *n_contrib, float border_eps, float border_eps_T,
                uint32_t *borderline) {
    const int W = st->image_width, H = st->image_height;
    const int gx = (W + GSO_BLOCK_X - 1) / GSO_BLOCK_X;
#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py < H; py++) {
        for (int px = 0; px < W; px++) {
            const int tile = (py / GSO_BLOCK_Y) * gx + (px / GSO_BLOCK_X);
            const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            const float pfx = (float)px, pfy = (float)py;
            float T = 1.0f, C[3] = {0.f, 0.f, 0.f}, Dacc = 0.f;
            uint32_t contributor = 0, last = 0, nborder = 0;
            for (uint32_t j = r0; j < r1; j++) {
                contributor++;
                const uint32_t g = point_list[j];
                const float dx = means2D[2 * g] - pfx, dy = means2D[2 * g + 1] - pfy;
                const float *co = conic_opacity + 4 * (size_t)g;
                const float q = fmaf(co[2] * dy, dy, (co[0] * dx) * dx);
                const float power = fmaf(-(co[1] * dx), dy, -0.5f * q);
                if (power > 0.0f) continue;
                const float a_raw = co[3] * expf(power);
                const float alpha = fminf(0.99f, a_raw);
                if (borderline && fabsf(a_raw - (1.0f / 255.0f)) <= border_eps * (1.0f / 255.0f)) nborder++;
                if (alpha < 1.0f / 255.0f) continue;
                const float test_T = T * (1.0f - alpha);
                if (borderline && fabsf(test_T - 0.0001f) <= border_eps_T * 0.0001f) nborder++;
                if (test_T < 0.0001f) break; /* done: this instance is NOT added */
#ifdef GSO_ASSOC_UPSTREAM
                /* exposure build (tools/fma_exposure.py, never a checker): upstream writes `C[ch] += features * alpha * T`,
                 * which parses as ((features * alpha) * T) + C -- the canonical order below forms the weight alpha * T
                 * first (one multiply instead of four per contribution in the kernels).  A reassociation of roundings,
                 * up to 1 ulp per contribution; what it moves is counted in profiles/fma_exposure.json. */
                C[0] = fmaf(rgb[3 * (size_t)g + 0] * alpha, T, C[0]);
                C[1] = fmaf(rgb[3 * (size_t)g + 1] * alpha, T, C[1]);
                C[2] = fmaf(rgb[3 * (size_t)g + 2] * alpha, T, C[2]);
                Dacc = fmaf((1.0f / depths[g]) * alpha, T, Dacc);
#else
                const float w = alpha * T;
                C[0] = fmaf(rgb[3 * (size_t)g + 0], w, C[0]);
                C[1] = fmaf(rgb[3 * (size_t)g + 1], w, C[1]);
                C[2] = fmaf(rgb[3 * (size_t)g + 2], w, C[2]);
                Dacc = fmaf(1.0f / depths[g], w, Dacc);
#endif
                T = test_T;
                last = contributor;
            }
            const size_t pid = (size_t)py * W + px;
            final_T[pid] = T;
            n_contrib[pid] = last;
            for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * H * W + pid] = fmaf(T, bg[ch], C[ch]);
            out_invdepth[pid] = Dacc;
            if (borderline) borderline[pid] = nborder;
        }
    }
}

/* Brute-force cross-check (SURVEY.md 8c KAT 8): per pixel, ALL Gaussians that pass preprocess and whose
 * tile rect contains the pixel's tile, ordered by (depth bits, index), composited in binary64. */
typedef struct { uint32_t d; uint32_t g; } GsoDG;
static int cmp_dg(const void *a, const void *b) {
    const GsoDG *x = (const GsoDG *)a, *y = (const GsoDG *)b;
    if (x->d != y->d) return x->d < y->d ? -1 : 1;
    return x->g < y->g ? -1 : (x->g > y->g);
}
void gso_render_bruteforce(const GsoSettings *st, int P, const int32_t *radii, const int32_t *rects,
                           const float *means2D, const float *conic_opacity, const float *rgb,
                           const float *depths, const float *bg, double *out_color,
                           double *out_invdepth) {
    const int W = st->image_width, H = st->image_height;
    GsoDG *order = (GsoDG *)malloc(sizeof(GsoDG) * (size_t)(P > 0 ? P : 1));
    int n = 0;
    for (int i = 0; i < P; i++)
        if (radii[i] > 0) {
            memcpy(&order[n].d, &depths[i], 4);
            order[n].g = (uint32_t)i;
            n++;
        }
    qsort(order, (size_t)n, sizeof(GsoDG), cmp_dg);
#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            const int tx = px / GSO_BLOCK_X, ty = py / GSO_BLOCK_Y;
            double T = 1.0, C[3] = {0, 0, 0}, Dacc = 0;
            for (int k = 0; k < n; k++) {
                const uint32_t g = order[k].g;
                if (tx < rects[4 * g] || tx >= rects[4 * g + 2] || ty < rects[4 * g + 1] ||
                    ty >= rects[4 * g + 3])
                    continue;
                const double dx = (double)means2D[2 * g] - px, dy = (double)means2D[2 * g + 1] - py;
                const float *co = conic_opacity + 4 * (size_t)g;
                const double power = -0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0) continue;
                double alpha = co[3] * exp(power);
                if (alpha > 0.99) alpha = 0.99;
                if (alpha < 1.0 / 255.0) continue;
                const double test_T = T * (1 - alpha);
                if (test_T < 0.0001) break;
                for (int ch = 0; ch < 3; ch++) C[ch] += rgb[3 * (size_t)g + ch] * alpha * T;
                Dacc += alpha * T / depths[g];
                T = test_T;
            }
            const size_t pid = (size_t)py * W + px;
            for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * H * W + pid] = C[ch] + T * bg[ch];
            out_invdepth[pid] = Dacc;
        }
    free(order);
}

/* ------------------------------------------------------------------------------------------------
 * simple-knn distCUDA2: mean of the squared distances to the 3 nearest OTHER points (self excluded by
 * index, duplicates give 0).  The Morton/box machinery upstream only accelerates the search; the
 * result is the exact 3-NN, so the restatement is the O(N^2) definition with the same float
 * expression for the distance and the same ascending 3-best accumulation.
 * ------------------------------------------------------------------------------------------------ */
void gso_knn_dist2(int P, const float *pts, float *out) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        for (int j = 0; j < P; j++) {
            if (j == i) continue;
            const float dx = x - pts[3 * j], dy = y - pts[3 * j + 1], dz = z - pts[3 * j + 2];
            float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            for (int k = 0; k < 3; k++)
                if (best[k] > d) { float t = best[k]; best[k] = d; d = t; }
        }
        out[i] = (best[0] + best[1] + best[2]) / 3.0f;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Backward pass (backward.cu: renderCUDA, computeCov2DCUDA, preprocessCUDA).
 * The CUDA kernels accumulate with float atomics in a non-deterministic order; this restatement keeps the
 * same per-(pixel, instance) float32 expressions and sums them in binary64, so it is the order-free value
 * the HIP path is compared against (tolerance stated in tests/test_backward_gpu.py).
 * Conventions kept from upstream: dL_dconic.y holds HALF of the off-diagonal derivative (the 2D-covariance
 * stage multiplies it back); the 0.99 alpha cap is transparent to the gradient; dL_dscale is the derivative
 * w.r.t. (scale_modifier * scale) and is not multiplied by scale_modifier.
 * ------------------------------------------------------------------------------------------------ */
void gso_render_backward(const GsoSettings *st, int P, const uint32_t *ranges, const uint32_t *point_list,
                         const float *means2D, const float *conic_opacity, const float *rgb,
                         const float *depths, const float *bg, const float *final_T,
                         const uint32_t *n_contrib, const float *dL_dpix, const float *dL_dinvdepth_pix,
                         /* out, all zero-initialised here */ double *dL_dmean2D /*2P*/,
                         double *dL_dconic /*3P: xx, xy(half), yy*/, double *dL_dopacity /*P*/,
                         double *dL_dcolors /*3P*/, double *dL_dinvdepths /*P*/) {
    const int W = st->image_width, H = st->image_height;
    const int gx = (W + GSO_BLOCK_X - 1) / GSO_BLOCK_X;
    memset(dL_dmean2D, 0, sizeof(double) * 2 * (size_t)P);
    memset(dL_dconic, 0, sizeof(double) * 3 * (size_t)P);
    memset(dL_dopacity, 0, sizeof(double) * (size_t)P);
    memset(dL_dcolors, 0, sizeof(double) * 3 * (size_t)P);
    memset(dL_dinvdepths, 0, sizeof(double) * (size_t)P);
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            const int tile = (py / GSO_BLOCK_Y) * gx + (px / GSO_BLOCK_X);
            const uint32_t r0 = ranges[2 * tile];
            const size_t pid = (size_t)py * W + px;
            const float T_final = final_T[pid];
            float T = T_final;
            const uint32_t last_contributor = n_contrib[pid];
            float accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0};
            float accum_invd = 0.f, last_invd = 0.f, last_alpha = 0.f;
            float dLp[3];
            for (int ch = 0; ch < 3; ch++) dLp[ch] = dL_dpix[(size_t)ch * H * W + pid];
            const float dLd = dL_dinvdepth_pix ? dL_dinvdepth_pix[pid] : 0.f;
            const float bg_dot = fmaf(bg[2], dLp[2], fmaf(bg[1], dLp[1], bg[0] * dLp[0]));
            const float pfx = (float)px, pfy = (float)py;
            /* instances [r0, r0 + last_contributor) were examined before the pixel finished; walk them back to front */
            for (uint32_t k = last_contributor; k-- > 0;) {
                const uint32_t g = point_list[r0 + k];
                const float dx = means2D[2 * g] - pfx, dy = means2D[2 * g + 1] - pfy;
                const float *co = conic_opacity + 4 * (size_t)g;
                const float q = fmaf(co[2] * dy, dy, (co[0] * dx) * dx);
                const float power = fmaf(-(co[1] * dx), dy, -0.5f * q);
                if (power > 0.0f) continue;
                const float G = expf(power);
                const float alpha = fminf(0.99f, co[3] * G);
                if (alpha < 1.0f / 255.0f) continue;
                T = T / (1.f - alpha);
                const float dchannel_dcolor = alpha * T;
                float dL_dalpha = 0.f;
                for (int ch = 0; ch < 3; ch++) {
                    const float c = rgb[3 * (size_t)g + ch];
                    accum_rec[ch] = fmaf(last_alpha, last_color[ch], (1.f - last_alpha) * accum_rec[ch]);
                    last_color[ch] = c;
                    dL_dalpha = fmaf(c - accum_rec[ch], dLp[ch], dL_dalpha);
                    dL_dcolors[3 * (size_t)g + ch] += (double)(dchannel_dcolor * dLp[ch]);
                }
                if (dL_dinvdepth_pix) {
                    const float invd = 1.f / depths[g];
                    accum_invd = fmaf(last_alpha, last_invd, (1.f - last_alpha) * accum_invd);
                    last_invd = invd;
                    dL_dalpha = fmaf(invd - accum_invd, dLd, dL_dalpha);
                    dL_dinvdepths[g] += (double)(dchannel_dcolor * dLd);
                }
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha = fmaf(-T_final / (1.f - alpha), bg_dot, dL_dalpha);
                const float dL_dG = co[3] * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = fmaf(-gdy, co[1], -gdx * co[0]);
                const float dG_ddely = fmaf(-gdx, co[1], -gdy * co[2]);
                dL_dmean2D[2 * (size_t)g] += (double)((dL_dG * dG_ddelx) * ddelx_dx);
                dL_dmean2D[2 * (size_t)g + 1] += (double)((dL_dG * dG_ddely) * ddely_dy);
                dL_dconic[3 * (size_t)g] += (double)((-0.5f * gdx) * dx * dL_dG);
                dL_dconic[3 * (size_t)g + 1] += (double)((-0.5f * gdx) * dy * dL_dG);
                dL_dconic[3 * (size_t)g + 2] += (double)((-0.5f * gdy) * dy * dL_dG);
                dL_dopacity[g] += (double)(G * dL_dalpha);
            }
        }
}

/* Per-Gaussian chain: conic -> 2D covariance -> (3D covariance, view-space mean), projected-mean and
 * inverse-depth gradients, SH -> dL_dsh and view direction, 3D covariance -> scale and quaternion.
 * Inputs are the double accumulators of gso_render_backward rounded to float (what the atomics produce). */
void gso_preprocess_backward(const GsoSettings *st, int P, const float *means3D, const int32_t *radii,
                             const float *shs, const uint8_t *clamped, const float *opacities,
                             const float *scales, const float *rotations, const float *cov3D,
                             int cov3D_is_precomp, int colors_are_precomp, const float *viewmatrix,
                             const float *projmatrix, const float *campos, const float *dL_dmean2D,
                             const float *dL_dconic, float *dL_dopacity /* in/out */,
                             const float *dL_dcolors, const float *dL_dinvdepths,
                             /* out */ float *dL_dmeans3D, float *dL_dcov3D, float *dL_dsh, float *dL_dscales,
                             float *dL_drots) {
    const int W = st->image_width, H = st->image_height;
    const float fx = (float)W / (2.0f * st->tanfovx), fy = (float)H / (2.0f * st->tanfovy);
    const int D = st->sh_degree, M = st->sh_coeffs;
    memset(dL_dmeans3D, 0, sizeof(float) * 3 * (size_t)P);
    memset(dL_dcov3D, 0, sizeof(float) * 6 * (size_t)P);
    if (dL_dsh) memset(dL_dsh, 0, sizeof(float) * 3 * (size_t)M * (size_t)P);
    if (dL_dscales) memset(dL_dscales, 0, sizeof(float) * 3 * (size_t)P);
    if (dL_drots) memset(dL_drots, 0, sizeof(float) * 4 * (size_t)P);
    const float *m = viewmatrix;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        if (radii[i] <= 0) continue;
        const float px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
        const float *c6 = cov3D + 6 * (size_t)i;
        /* ---- computeCov2DCUDA ---------------------------------------------------------------- */
        float t[3];
        xform4x3(m, px, py, pz, t);
        const float limx = 1.3f * st->tanfovx, limy = 1.3f * st->tanfovy;
        const float txtz = t[0] / t[2], tytz = t[1] / t[2];
        const float tx = fminf(limx, fmaxf(-limx, txtz)) * t[2];
        const float ty = fminf(limy, fmaxf(-limy, tytz)) * t[2];
        const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
        const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
        const float tz = t[2];
        const float J00 = fx / tz, J02 = -(fx * tx) / (tz * tz), J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
        float A[2][3], Wm[3][3];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) Wm[r][c] = m[c * 4 + r];
        for (int j = 0; j < 3; j++) {
            A[0][j] = fmaf(J02, Wm[2][j], J00 * Wm[0][j]);
            A[1][j] = fmaf(J12, Wm[2][j], J11 * Wm[1][j]);
        }
        const float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
        float B[2][3];
        for (int r = 0; r < 2; r++)
            for (int j = 0; j < 3; j++)
                B[r][j] = fmaf(A[r][2], S[2][j], fmaf(A[r][1], S[1][j], A[r][0] * S[0][j]));
        float a = fmaf(B[0][2], A[0][2], fmaf(B[0][1], A[0][1], B[0][0] * A[0][0]));
        const float b = fmaf(B[0][2], A[1][2], fmaf(B[0][1], A[1][1], B[0][0] * A[1][0]));
        float c = fmaf(B[1][2], A[1][2], fmaf(B[1][1], A[1][1], B[1][0] * A[1][0]));
        const float h_var = 0.3f;
        float dL_da_aa = 0.f, dL_db_aa = 0.f, dL_dc_aa = 0.f;
        if (st->antialiasing) {
            const float det_cov = fmaf(-b, b, a * c);
            a += h_var;
            c += h_var;
            const float det_plus = fmaf(-b, b, a * c);
            const float ratio = det_cov / det_plus;
            const float h_scale = sqrtf(fmaxf(0.000025f, ratio));
            const float dL_dop = dL_dopacity[i];
            const float d_h = dL_dop * opacities[i];
            dL_dopacity[i] = dL_dop * h_scale;
            const float d_root = ratio <= 0.000025f ? 0.f : d_h / (2.f * h_scale);
            /* ratio = ((a-h)(c-h) - b^2) / (a c - b^2) with a, c the post-filter entries */
            const float inv2 = 1.f / (det_plus * det_plus);
            dL_da_aa = d_root * ((c - h_var) * det_plus - det_cov * c) * inv2;
            dL_dc_aa = d_root * ((a - h_var) * det_plus - det_cov * a) * inv2;
            dL_db_aa = d_root * (-2.f * b * det_plus + 2.f * b * det_cov) * inv2;
        } else {
            a += h_var;
            c += h_var;
        }
        const float Lx = dL_dconic[3 * (size_t)i], Ly = dL_dconic[3 * (size_t)i + 1], Lz = dL_dconic[3 * (size_t)i + 2];
        const float denom = fmaf(-b, b, a * c);
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        if (denom2inv != 0.f) {
            dL_da = denom2inv * (-c * c * Lx + 2.f * b * c * Ly + (denom - a * c) * Lz);
            dL_dc = denom2inv * (-a * a * Lz + 2.f * a * b * Ly + (denom - a * c) * Lx);
            dL_db = denom2inv * 2.f * (b * c * Lx - (denom + 2.f * b * b) * Ly + a * b * Lz);
        }
        dL_da += dL_da_aa;
        dL_db += dL_db_aa;
        dL_dc += dL_dc_aa;
        float dS[6]; /* gradient w.r.t. the 6 stored entries (off-diagonals count twice) */
        dS[0] = A[0][0] * A[0][0] * dL_da + A[0][0] * A[1][0] * dL_db + A[1][0] * A[1][0] * dL_dc;
        dS[3] = A[0][1] * A[0][1] * dL_da + A[0][1] * A[1][1] * dL_db + A[1][1] * A[1][1] * dL_dc;
        dS[5] = A[0][2] * A[0][2] * dL_da + A[0][2] * A[1][2] * dL_db + A[1][2] * A[1][2] * dL_dc;
        dS[1] = 2.f * A[0][0] * A[0][1] * dL_da + (A[0][0] * A[1][1] + A[0][1] * A[1][0]) * dL_db +
                2.f * A[1][0] * A[1][1] * dL_dc;
        dS[2] = 2.f * A[0][0] * A[0][2] * dL_da + (A[0][0] * A[1][2] + A[0][2] * A[1][0]) * dL_db +
                2.f * A[1][0] * A[1][2] * dL_dc;
        dS[4] = 2.f * A[0][2] * A[0][1] * dL_da + (A[0][1] * A[1][2] + A[0][2] * A[1][1]) * dL_db +
                2.f * A[1][1] * A[1][2] * dL_dc;
        for (int k = 0; k < 6; k++) dL_dcov3D[6 * (size_t)i + k] = dS[k];
        /* dL/dA = 2 G (A Sigma), G = [[da, db/2],[db/2, dc]] */
        float dA[2][3];
        for (int j = 0; j < 3; j++) {
            dA[0][j] = 2.f * B[0][j] * dL_da + B[1][j] * dL_db;
            dA[1][j] = 2.f * B[1][j] * dL_dc + B[0][j] * dL_db;
        }
        const float dJ00 = Wm[0][0] * dA[0][0] + Wm[0][1] * dA[0][1] + Wm[0][2] * dA[0][2];
        const float dJ02 = Wm[2][0] * dA[0][0] + Wm[2][1] * dA[0][1] + Wm[2][2] * dA[0][2];
        const float dJ11 = Wm[1][0] * dA[1][0] + Wm[1][1] * dA[1][1] + Wm[1][2] * dA[1][2];
        const float dJ12 = Wm[2][0] * dA[1][0] + Wm[2][1] * dA[1][1] + Wm[2][2] * dA[1][2];
        const float itz = 1.f / tz, itz2 = itz * itz, itz3 = itz2 * itz;
        const float dtx = x_grad_mul * -fx * itz2 * dJ02;
        const float dty = y_grad_mul * -fy * itz2 * dJ12;
        float dtz = -fx * itz2 * dJ00 - fy * itz2 * dJ11 + (2.f * fx * tx) * itz3 * dJ02 + (2.f * fy * ty) * itz3 * dJ12;
        if (dL_dinvdepths) dtz -= dL_dinvdepths[i] / (tz * tz);
        float dmean[3];
        for (int j = 0; j < 3; j++) dmean[j] = m[j * 4 + 0] * dtx + m[j * 4 + 1] * dty + m[j * 4 + 2] * dtz;
        /* ---- projected mean (preprocessCUDA bwd) ----------------------------------------------- */
        const float *q = projmatrix;
        float ph[4];
        xform4x4(q, px, py, pz, ph);
        const float m_w = 1.0f / (ph[3] + 0.0000001f);
        const float mul1 = ph[0] * m_w * m_w, mul2 = ph[1] * m_w * m_w;
        const float g2x = dL_dmean2D[2 * (size_t)i], g2y = dL_dmean2D[2 * (size_t)i + 1];
        dmean[0] += (q[0] * m_w - q[3] * mul1) * g2x + (q[1] * m_w - q[3] * mul2) * g2y;
        dmean[1] += (q[4] * m_w - q[7] * mul1) * g2x + (q[5] * m_w - q[7] * mul2) * g2y;
        dmean[2] += (q[8] * m_w - q[11] * mul1) * g2x + (q[9] * m_w - q[11] * mul2) * g2y;
        /* ---- SH -> colour (computeColorFromSH bwd) ------------------------------------------------ */
        if (!colors_are_precomp && dL_dsh) {
            const float ox = px - campos[0], oy = py - campos[1], oz = pz - campos[2];
            const float len = sqrtf(fmaf(oz, oz, fmaf(oy, oy, ox * ox)));
            const float x = ox / len, y = oy / len, z = oz / len;
            float dRGB[3];
            for (int ch = 0; ch < 3; ch++) dRGB[ch] = clamped[3 * (size_t)i + ch] ? 0.f : dL_dcolors[3 * (size_t)i + ch];
            float bas[16], bx[16], by[16], bz[16];
            sh_basis(D, x, y, z, bas);
            for (int k = 0; k < 16; k++) bx[k] = by[k] = bz[k] = 0.f;
            if (D > 0) {
                by[1] = -SH_C1; bz[2] = SH_C1; bx[3] = -SH_C1;
                if (D > 1) {
                    const float xx = x * x, yy = y * y, zz = z * z;
                    bx[4] = SH_C2[0] * y; by[4] = SH_C2[0] * x;
                    by[5] = SH_C2[1] * z; bz[5] = SH_C2[1] * y;
                    bx[6] = SH_C2[2] * -2.f * x; by[6] = SH_C2[2] * -2.f * y; bz[6] = SH_C2[2] * 4.f * z;
                    bx[7] = SH_C2[3] * z; bz[7] = SH_C2[3] * x;
                    bx[8] = SH_C2[4] * 2.f * x; by[8] = SH_C2[4] * -2.f * y;
                    if (D > 2) {
                        bx[9] = SH_C3[0] * 6.f * x * y; by[9] = SH_C3[0] * (3.f * xx - 3.f * yy);
                        bx[10] = SH_C3[1] * y * z; by[10] = SH_C3[1] * x * z; bz[10] = SH_C3[1] * x * y;
                        bx[11] = SH_C3[2] * -2.f * x * y; by[11] = SH_C3[2] * (4.f * zz - xx - 3.f * yy);
                        bz[11] = SH_C3[2] * 8.f * y * z;
                        bx[12] = SH_C3[3] * -6.f * x * z; by[12] = SH_C3[3] * -6.f * y * z;
                        bz[12] = SH_C3[3] * (6.f * zz - 3.f * xx - 3.f * yy);
                        bx[13] = SH_C3[4] * (4.f * zz - 3.f * xx - yy); by[13] = SH_C3[4] * -2.f * x * y;
                        bz[13] = SH_C3[4] * 8.f * x * z;
                        bx[14] = SH_C3[5] * 2.f * x * z; by[14] = SH_C3[5] * -2.f * y * z; bz[14] = SH_C3[5] * (xx - yy);
                        bx[15] = SH_C3[6] * (3.f * xx - 3.f * yy); by[15] = SH_C3[6] * -6.f * x * y;
                    }
                }
            }
            const int nb = (D + 1) * (D + 1);
            const float *sh = shs + (size_t)i * M * 3;
            float ddir[3] = {0, 0, 0};
            for (int k = 0; k < nb; k++)
                for (int ch = 0; ch < 3; ch++) {
                    dL_dsh[((size_t)i * M + k) * 3 + ch] = bas[k] * dRGB[ch];
                    ddir[0] += bx[k] * sh[3 * k + ch] * dRGB[ch];
                    ddir[1] += by[k] * sh[3 * k + ch] * dRGB[ch];
                    ddir[2] += bz[k] * sh[3 * k + ch] * dRGB[ch];
                }
            /* through the normalisation dir = o / |o| */
            const float sum2 = ox * ox + oy * oy + oz * oz;
            const float inv32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            dmean[0] += ((sum2 - ox * ox) * ddir[0] - oy * ox * ddir[1] - oz * ox * ddir[2]) * inv32;
            dmean[1] += (-ox * oy * ddir[0] + (sum2 - oy * oy) * ddir[1] - oz * oy * ddir[2]) * inv32;
            dmean[2] += (-ox * oz * ddir[0] - oy * oz * ddir[1] + (sum2 - oz * oz) * ddir[2]) * inv32;
        }
        for (int j = 0; j < 3; j++) dL_dmeans3D[3 * (size_t)i + j] = dmean[j];
        /* ---- 3D covariance -> scale, quaternion (computeCov3D bwd) -------------------------------- */
        if (!cov3D_is_precomp && dL_dscales && dL_drots) {
            const float *rq = rotations + 4 * (size_t)i;
            const float r = rq[0], x = rq[1], y = rq[2], z = rq[3];
            float R[3][3];
            R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
            R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
            R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
            float s[3];
            for (int k = 0; k < 3; k++) s[k] = st->scale_modifier * scales[3 * (size_t)i + k];
            float Mm[3][3];
            for (int k = 0; k < 3; k++)
                for (int j = 0; j < 3; j++) Mm[k][j] = s[k] * R[j][k];
            const float Gs[3][3] = {{dS[0], 0.5f * dS[1], 0.5f * dS[2]},
                                    {0.5f * dS[1], dS[3], 0.5f * dS[4]},
                                    {0.5f * dS[2], 0.5f * dS[4], dS[5]}};
            float dM[3][3]; /* dL/dM = 2 M G */
            for (int k = 0; k < 3; k++)
                for (int j = 0; j < 3; j++)
                    dM[k][j] = 2.f * (Mm[k][0] * Gs[0][j] + Mm[k][1] * Gs[1][j] + Mm[k][2] * Gs[2][j]);
            float dR[3][3]; /* dL/dR[j][k] = s_k dM[k][j] */
            for (int k = 0; k < 3; k++) {
                dL_dscales[3 * (size_t)i + k] = R[0][k] * dM[k][0] + R[1][k] * dM[k][1] + R[2][k] * dM[k][2];
                for (int j = 0; j < 3; j++) dR[j][k] = s[k] * dM[k][j];
            }
            float *dq = dL_drots + 4 * (size_t)i;
            dq[0] = 2.f * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
            dq[1] = 2.f * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - 2.f * x * dR[1][1] - r * dR[1][2] +
                           z * dR[2][0] + r * dR[2][1] - 2.f * x * dR[2][2]);
            dq[2] = 2.f * (-2.f * y * dR[0][0] + x * dR[0][1] + r * dR[0][2] + x * dR[1][0] + z * dR[1][2] -
                           r * dR[2][0] + z * dR[2][1] - 2.f * y * dR[2][2]);
            dq[3] = 2.f * (-2.f * z * dR[0][0] - r * dR[0][1] + x * dR[0][2] + r * dR[1][0] - 2.f * z * dR[1][1] +
                           y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
        }
    }
}

/* ---------------------------------------------------------------------------------------------------------
 * Parameter activations at the boundary: what upstream's GaussianModel getters apply per frame
 * before the rasterizer -- opacity = sigmoid(logit), scale = exp(log scale), rotation = q / max(|q|, 1e-12)
 * (scene/gaussian_model.py get_opacity / get_scaling / get_rotation = torch.sigmoid / torch.exp /
 * torch.nn.functional.normalize).  torch's exp is a platform library call, so the canonical float32 order is
 * fixed HERE and the HIP preprocess evaluates exactly these operations when it is handed raw parameters:
 * exp(x) = 2^n * p(r), n = rint(x * log2e), r = x - n * ln2 (Cody-Waite split), p = the Cephes expf polynomial,
 * every multiply-add an explicit fmaf.  ~1 ulp from the true exponential, i.e. as close as two exp libraries are
 * to each other.  flags: 1 = opacity logits, 2 = log scales, 4 = un-normalised rotations; untouched inputs copy.
 * --------------------------------------------------------------------------------------------------------- */
float gso_expf(float x) {
    if (x > 88.72283905206835f) return INFINITY;
    if (x < -103.972084045410f) return 0.0f;
    const float n = rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float y = __builtin_fmaf(p, r * r, r) + 1.0f;
    return ldexpf(y, (int)n);
}

void gso_activate_params(int P, int flags, const float *opacities, const float *scales, const float *rotations,
                         float *opacities_out, float *scales_out, float *rotations_out) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        if (opacities && opacities_out)
            opacities_out[i] = (flags & 1) ? 1.0f / (1.0f + gso_expf(-opacities[i])) : opacities[i];
        if (scales && scales_out)
            for (int k = 0; k < 3; k++)
                scales_out[3 * (size_t)i + k] = (flags & 2) ? gso_expf(scales[3 * (size_t)i + k]) : scales[3 * (size_t)i + k];
        if (rotations && rotations_out) {
            const float *q = rotations + 4 * (size_t)i;
            float d = 1.0f;
            if (flags & 4) {
                const float n2 = fmaf(q[3], q[3], fmaf(q[2], q[2], fmaf(q[1], q[1], q[0] * q[0])));
                d = fmaxf(sqrtf(n2), 1e-12f);
            }
            for (int k = 0; k < 4; k++) rotations_out[4 * (size_t)i + k] = (flags & 4) ? q[k] / d : q[k];
        }
    }
}
