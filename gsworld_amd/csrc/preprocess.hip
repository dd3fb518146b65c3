// preprocess.hip -- per-Gaussian projection stage (upstream forward.cu preprocessCUDA + auxiliary.h helpers;
// SURVEY.md 8a rows A3/A4/A9).  One thread per Gaussian, 256-thread workgroups.
//
// Arithmetic contract: compiled with -ffp-contract=off; every fused multiply-add below is an explicit
// __builtin_fmaf, division and sqrtf are the correctly rounded forms (hipcc default), so the depth keys, radii
// and tile rects are bit-identical to the canonical order fixed by oracle/gs_oracle.c.
//
// HBM traffic per Gaussian: 12 B xyz always; +28 B scale/quat, +4 B opacity once the near cull passed;
// +192 B of SH and 88 B of state written only for Gaussians that survive to a non-empty tile rect.
#include "gsr_internal.h"

#ifndef GSR_TIGHT_RECT
#define GSR_TIGHT_RECT 1
#endif

namespace {

struct PreprocessArgs {
    int P, D, M;
    int W, H, gx, gy;
    float tanfovx, tanfovy, fx, fy, scale_modifier, near_plane;
    int antialiasing;
    int param_space;  // GSR_RAW_* flags: activations evaluated here instead of three torch passes per frame
    int infer;        // GsrSettings.forward_only: nothing a backward would read is written; rec2.w = packed tile rect
    const float *means3D, *shs, *shs_rest, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
    const float *view, *proj, *campos;
    // optional rigid transform of labelled Gaussians (GsrInputs.part_*): same arithmetic as transform.hip
    const float *part_labels, *part_transforms;
    const int32_t *part_lut;
    const uint8_t *part_rescale;
    int part_lut_size, part_count;
    // optional block bounds (GsrInputs.cull_blocks) and the original numbering of a permuted model (GsrInputs.orig_index):
    // the state stays in the numbering of the ARRAYS (a block's records are neighbours in memory, and so are the records a
    // tile's list gathers); only `radii` -- the caller's array -- is written by original number, and the depth sort
    // breaks ties by it (depthsort.hip)
    const float *cull_blocks;
    const int32_t *orig_index;
    int32_t *radii;
    float4 *splat;
    float *cov3D;
    uint32_t *clamped;
    uint32_t *tiles_touched;
    uint2 *rects;
    uint32_t *block_counts;
    uint2 *block_recs;     // [P] (index, depth bits) of the block's visible Gaussians, compacted to the head of the block's
                           // own 256 slots (depthsort.hip gathers them: 8 B per VISIBLE Gaussian instead of a 4-byte key
                           // written and re-read for all N)

    // bin-then-sort path: per-tile instance totals and the visible count are accumulated here
    int num_tiles;
    uint32_t *tile_accum;
    GsrHeader *hdr;
};

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

constexpr float kC0 = 0.28209479177387814f;
constexpr float kC1 = 0.4886025119029199f;
constexpr float kC2_0 = 1.0925484305920792f, kC2_1 = -1.0925484305920792f, kC2_2 = 0.31539156525252005f,
                kC2_3 = -1.0925484305920792f, kC2_4 = 0.5462742152960396f;
constexpr float kC3_0 = -0.5900435899266435f, kC3_1 = 2.890611442640554f, kC3_2 = -0.4570457994644658f,
                kC3_3 = 0.3731763325901154f, kC3_4 = -0.4570457994644658f, kC3_5 = 1.445305721320277f,
                kC3_6 = -0.5900435899266435f;

// real SH basis of a unit direction (coefficient signs of forward.cu computeColorFromSH)
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float *b) {
    b[0] = kC0;
    if (deg > 0) {
        b[1] = -(kC1 * y);
        b[2] = kC1 * z;
        b[3] = -(kC1 * x);
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = kC2_0 * xy;
            b[5] = kC2_1 * yz;
            b[6] = kC2_2 * (fma_(2.f, zz, -xx) - yy);
            b[7] = kC2_3 * xz;
            b[8] = kC2_4 * (xx - yy);
            if (deg > 2) {
                b[9] = (kC3_0 * y) * fma_(3.f, xx, -yy);
                b[10] = (kC3_1 * xy) * z;
                b[11] = (kC3_2 * y) * (fma_(4.f, zz, -xx) - yy);
                b[12] = (kC3_3 * z) * fma_(-3.f, yy, fma_(-3.f, xx, 2.f * zz));
                b[13] = (kC3_4 * x) * (fma_(4.f, zz, -xx) - yy);
                b[14] = (kC3_5 * z) * (xx - yy);
                b[15] = (kC3_6 * x) * fma_(-3.f, yy, xx);
            }
        }
    }
}

// Position of Gaussian i after the optional rigid part transform (GsrInputs.part_*); xf = its pose row or nullptr.
__device__ __forceinline__ void prep_position(const PreprocessArgs &a, int i, float &px_, float &py_, float &pz_,
                                              const float *&xf_, int &part_) {
    float px = a.means3D[3 * (size_t)i], py = a.means3D[3 * (size_t)i + 1], pz = a.means3D[3 * (size_t)i + 2];
    // moving part?  (label -> part through the LUT; the reference compares labels after .long(): truncation)
    const float *xf = nullptr;
    int part = -1;
    if (a.part_labels != nullptr) {
        const int label = (int)a.part_labels[i];
        part = (label >= 0 && label < a.part_lut_size) ? a.part_lut[label] : -1;
        if (part >= 0 && part < a.part_count) {
            xf = a.part_transforms + (size_t)part * 17;
            // xyz' = R (s xyz) + t, in the operation order of transform.hip (plain multiplies and adds)
            const float s = xf[12];
            px *= s; py *= s; pz *= s;
            const float rx = xf[0] * px + xf[1] * py + xf[2] * pz + xf[9];
            const float ry = xf[3] * px + xf[4] * py + xf[5] * pz + xf[10];
            const float rz = xf[6] * px + xf[7] * py + xf[8] * pz + xf[11];
            px = rx; py = ry; pz = rz;
        }
    }
    px_ = px; py_ = py; pz_ = pz; xf_ = xf; part_ = part;
}

struct GeomOut {
    bool visible;
    float4 pos;      // xyz (after the part transform) + radius as float, handed to the colour phase
    uint32_t tiles;  // tiles touched
    uint32_t key;    // depth bits of a visible Gaussian, 0 otherwise
    uint2 rect;
    int radius;
};

// The exact per-Gaussian geometry: cull, projection, covariances, conic, radius, rect.
// Writes rec0 / rec1 / cov3D / rects of a visible Gaussian; the caller writes radii, tiles_touched and the block-local sort record.
__device__ __forceinline__ GeomOut prep_geometry(const PreprocessArgs &a, int i) {
    bool visible = false;
    float4 mypos = make_float4(0.f, 0.f, 0.f, 0.f);
    uint2 my_rect = make_uint2(0u, 0u);
    float px, py, pz;
    const float *xf;
    int part;
    prep_position(a, i, px, py, pz, xf, part);
    const float *m = a.view;
    // transformPoint4x3: M[r][c] = m[c*4+r]
    const float vx = fma_(m[8], pz, fma_(m[4], py, m[0] * px)) + m[12];
    const float vy = fma_(m[9], pz, fma_(m[5], py, m[1] * px)) + m[13];
    const float vz = fma_(m[10], pz, fma_(m[6], py, m[2] * px)) + m[14];
    int radius = 0;
    uint32_t touched = 0;
    if (vz > a.near_plane) {  // in_frustum with GSWorld's near plane
        const float *q = a.proj;
        const float hx = fma_(q[8], pz, fma_(q[4], py, q[0] * px)) + q[12];
        const float hy = fma_(q[9], pz, fma_(q[5], py, q[1] * px)) + q[13];
        const float hw = fma_(q[11], pz, fma_(q[7], py, q[3] * px)) + q[15];
        const float p_w = 1.0f / (hw + 0.0000001f);
        const float ndc_x = hx * p_w, ndc_y = hy * p_w;

        // ---- 3D covariance: Sigma = R diag((mod*s)^2) R^T ------------------------------------------------
        // (requested with the covariance inputs: one round trip, not two.  Requesting all of them together with the
        //  POSITION in blocks that passed the frustum test -- nearly every Gaussian of such a block is in front of the
        //  camera -- was measured in round 4: no gain, 6 468 against 6 481 frames/s one at a time)
        const float opacity_raw = a.opacities[i];
        float c0, c1, c2, c3, c4, c5;
        if (a.cov3D_precomp) {
            const float *c = a.cov3D_precomp + 6 * (size_t)i;
            c0 = c[0]; c1 = c[1]; c2 = c[2]; c3 = c[3]; c4 = c[4]; c5 = c[5];
        } else {
            float4 rq = *reinterpret_cast<const float4 *>(a.rotations + 4 * (size_t)i);
            float sc0 = a.scales[3 * (size_t)i], sc1 = a.scales[3 * (size_t)i + 1], sc2 = a.scales[3 * (size_t)i + 2];
            if (xf != nullptr) {
                // rot' = standardize(q_R (x) rot / |rot|) * |rot|  (gs_utils.py:242-249; transform.hip)
                const float norm = sqrtf(rq.x * rq.x + rq.y * rq.y + rq.z * rq.z + rq.w * rq.w);
                const float bw = rq.x / norm, bx = rq.y / norm, by = rq.z / norm, bz = rq.w / norm;
                const float aw = xf[13], ax = xf[14], ay = xf[15], az = xf[16];
                float ow = aw * bw - ax * bx - ay * by - az * bz;
                float ox = aw * bx + ax * bw + ay * bz - az * by;
                float oy = aw * by - ax * bz + ay * bw + az * bx;
                float oz = aw * bz + ax * by - ay * bx + az * bw;
                if (ow < 0.f) { ow = -ow; ox = -ox; oy = -oy; oz = -oz; }
                rq = make_float4(ow * norm, ox * norm, oy * norm, oz * norm);
                if (a.part_rescale != nullptr && a.part_rescale[part]) {
                    // the reference's rewrite of a tracked actor's log-scales: inverse_sigmoid(exp(s) * scale)
                    const float s = xf[12];
                    const float x0 = expf(sc0) * s, x1 = expf(sc1) * s, x2 = expf(sc2) * s;
                    sc0 = logf(x0 / (1.0f - x0));
                    sc1 = logf(x1 / (1.0f - x1));
                    sc2 = logf(x2 / (1.0f - x2));
                }
            }
            if (a.param_space & GSR_RAW_ROTATIONS) {  // F.normalize: q / max(|q|, 1e-12)
                const float n2 = fma_(rq.w, rq.w, fma_(rq.z, rq.z, fma_(rq.y, rq.y, rq.x * rq.x)));
                const float d = fmaxf(sqrtf(n2), 1e-12f);
                rq = make_float4(rq.x / d, rq.y / d, rq.z / d, rq.w / d);
            }
            if (a.param_space & GSR_RAW_SCALES) {
                sc0 = exp_canonical(sc0);
                sc1 = exp_canonical(sc1);
                sc2 = exp_canonical(sc2);
            }
            const float r = rq.x, x = rq.y, y = rq.z, z = rq.w;
            const float s0 = a.scale_modifier * sc0;
            const float s1 = a.scale_modifier * sc1;
            const float s2 = a.scale_modifier * sc2;
            const float R00 = fma_(-2.f, fma_(z, z, y * y), 1.f);
            const float R01 = 2.f * fma_(-r, z, x * y);
            const float R02 = 2.f * fma_(r, y, x * z);
            const float R10 = 2.f * fma_(r, z, x * y);
            const float R11 = fma_(-2.f, fma_(z, z, x * x), 1.f);
            const float R12 = 2.f * fma_(-r, x, y * z);
            const float R20 = 2.f * fma_(-r, y, x * z);
            const float R21 = 2.f * fma_(r, x, y * z);
            const float R22 = fma_(-2.f, fma_(y, y, x * x), 1.f);
            // M[k][j] = s_k * R[j][k]
            const float M00 = s0 * R00, M01 = s0 * R10, M02 = s0 * R20;
            const float M10 = s1 * R01, M11 = s1 * R11, M12 = s1 * R21;
            const float M20 = s2 * R02, M21 = s2 * R12, M22 = s2 * R22;
            c0 = fma_(M20, M20, fma_(M10, M10, M00 * M00));
            c1 = fma_(M20, M21, fma_(M10, M11, M00 * M01));
            c2 = fma_(M20, M22, fma_(M10, M12, M00 * M02));
            c3 = fma_(M21, M21, fma_(M11, M11, M01 * M01));
            c4 = fma_(M21, M22, fma_(M11, M12, M01 * M02));
            c5 = fma_(M22, M22, fma_(M12, M12, M02 * M02));
        }

        // ---- EWA 2D covariance: (J W) Sigma (J W)^T ---------------------------------------------------------
        const float limx = 1.3f * a.tanfovx, limy = 1.3f * a.tanfovy;
        const float txtz = vx / vz, tytz = vy / vz;
        const float tx = fminf(limx, fmaxf(-limx, txtz)) * vz;
        const float ty = fminf(limy, fmaxf(-limy, tytz)) * vz;
        const float J00 = a.fx / vz, J02 = -(a.fx * tx) / (vz * vz);
        const float J11 = a.fy / vz, J12 = -(a.fy * ty) / (vz * vz);
        // A = J W with W[i][j] = m[j*4+i]
        const float A00 = fma_(J02, m[2], J00 * m[0]);
        const float A01 = fma_(J02, m[6], J00 * m[4]);
        const float A02 = fma_(J02, m[10], J00 * m[8]);
        const float A10 = fma_(J12, m[2], J11 * m[1]);
        const float A11 = fma_(J12, m[6], J11 * m[5]);
        const float A12 = fma_(J12, m[10], J11 * m[9]);
        // B = A Sigma
        const float B00 = fma_(A02, c2, fma_(A01, c1, A00 * c0));
        const float B01 = fma_(A02, c4, fma_(A01, c3, A00 * c1));
        const float B02 = fma_(A02, c5, fma_(A01, c4, A00 * c2));
        const float B10 = fma_(A12, c2, fma_(A11, c1, A10 * c0));
        const float B11 = fma_(A12, c4, fma_(A11, c3, A10 * c1));
        const float B12 = fma_(A12, c5, fma_(A11, c4, A10 * c2));
        float cxx = fma_(B02, A02, fma_(B01, A01, B00 * A00));
        const float cxy = fma_(B02, A12, fma_(B01, A11, B00 * A10));
        float cyy = fma_(B12, A12, fma_(B11, A11, B10 * A10));

        const float det_cov = fma_(-cxy, cxy, cxx * cyy);
        cxx += 0.3f;
        cyy += 0.3f;
        const float det = fma_(-cxy, cxy, cxx * cyy);
        float h_scale = 1.0f;
        if (a.antialiasing) h_scale = sqrtf(fmaxf(0.000025f, det_cov / det));
        if (det != 0.0f) {
            const float det_inv = 1.f / det;
            const float conic_x = cyy * det_inv, conic_y = -cxy * det_inv, conic_z = cxx * det_inv;
            const float mid = 0.5f * (cxx + cyy);
            const float root = sqrtf(fmaxf(0.1f, fma_(mid, mid, -det)));
            const float lambda1 = mid + root, lambda2 = mid - root;
            const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
            // ndc2Pix is binary64 upstream (double literals)
            const float pix_x = (float)((((double)ndc_x + 1.0) * (double)a.W - 1.0) * 0.5);
            const float pix_y = (float)((((double)ndc_y + 1.0) * (double)a.H - 1.0) * 0.5);
            const int ir = (int)my_radius;
            const float fr = (float)ir;
            int rminx = (int)((pix_x - fr) / (float)GSR_TILE);
            int rminy = (int)((pix_y - fr) / (float)GSR_TILE);
            int rmaxx = (int)((pix_x + fr + (float)(GSR_TILE - 1)) / (float)GSR_TILE);
            int rmaxy = (int)((pix_y + fr + (float)(GSR_TILE - 1)) / (float)GSR_TILE);
            rminx = min(a.gx, max(0, rminx));
            rminy = min(a.gy, max(0, rminy));
            rmaxx = min(a.gx, max(0, rmaxx));
            rmaxy = min(a.gy, max(0, rmaxy));
            int area = (rmaxx - rminx) * (rmaxy - rminy);
            if (area != 0) radius = ir;  // (what the caller sees as radii: the reference's, whatever the lists keep)
            float opacity = 0.f, third = vz;
            if (area != 0) {
                float opacity_in = opacity_raw;
                if (a.param_space & GSR_RAW_OPACITY) opacity_in = sigmoid_canonical(opacity_in);
                opacity = opacity_in * h_scale;
                if (a.infer) {
                    // inference frames: the compositor never reads the depth word of the record; it carries what its
                    // per-quadrant cull would otherwise recompute for every candidate of every quadrant (render.hip
                    // quadrant_may_hit): tau = -ln(255 opacity), lowered by ~2 ulp (the cull may only err towards
                    // keeping), with the "conic is comfortably positive definite" flag in its last mantissa bit
                    const float tau = -0.6931471805599453f * __builtin_amdgcn_logf(255.0f * opacity);
                    const float low = tau - fma_(2e-7f, fabsf(tau), 1e-30f);
                    const float ac = conic_x * conic_z;
                    const bool safe = conic_x > 0.0f && conic_z > 0.0f && ac > 1e-20f && conic_y * conic_y <= 0.999f * ac;
                    third = __uint_as_float((__float_as_uint(low) & ~1u) | (safe ? 1u : 0u));
#if GSR_TIGHT_RECT
                    // ... and the tile rect shrinks to the tiles the splat can COLOUR.  The reference bins by the
                    // square of ceil(3 sqrt(lambda_max)) around the centre; a pixel is composited only where
                    // alpha >= 1/255, i.e. inside the ellipse  A dx^2 + 2 B dx dy + C dy^2 <= -2 tau, whose bounding box
                    // has the half-widths sqrt(-2 tau C / det), sqrt(-2 tau A / det).  For the flat, tilted splats of a
                    // scanned table that box is a fraction of the square, and every instance outside it is one the
                    // compositor would fetch, test and skip.  Only for a comfortably positive definite conic (the
                    // computed power is then <= 0 everywhere: stream_conic_is_safe), with 1e-4 relative + a quarter
                    // pixel to spare (the compositor's own float error on the power is ~4e-6 of its terms); never
                    // beyond the reference's rect, which cuts opaque splats at 3 sigma.  Inference frames only: no
                    // backward reads these lists, the image is the same bit for bit (tests, tools/fuzz_forward_only.py).
                    if (safe) {
                        const float k = -2.0f * low;  // (<= 0: opacity below 1/255, alpha never passes)
                        const float detc = fma_(-conic_y, conic_y, ac);
                        // (hardware reciprocal / square root, ~1 ulp each: the margins are a hundred times that)
                        const float rdet = __builtin_amdgcn_rcpf(detc);
                        const float kd = k * rdet;
                        // The margin follows the conic's conditioning.  The compositor's power is a sum of terms that
                        // cancel when B^2 approaches A C: its float error is ~4e-6 of sum |terms| (the slack of
                        // quadrant_may_hit), and on the ellipse's boundary sum |terms| <= ~2 k A C / det.  A pixel just
                        // outside the exact ellipse can therefore still compute alpha >= 1/255 when the exact power is
                        // within 8e-6 k A C / det of tau: the ellipse grows by the factor sqrt(1 + 1.6e-5 A C / det) at
                        // most.  2e-5 A C / det (2.5 x that, up to 2 % at the `safe` limit B^2 = 0.999 A C) + 1e-4
                        // relative + a quarter pixel.  (Round 3 used 1e-4 + 0.25 px whatever the conditioning: short by
                        // up to 0.4 % of the half-width for needle-shaped splats near the limit -- ADVICE round 3.)
                        const float rel = fma_(2e-5f, ac * rdet, 1.0001f);
                        const float hx = fma_(__builtin_amdgcn_sqrtf(kd * conic_z), rel, 0.25f);
                        const float hy = fma_(__builtin_amdgcn_sqrtf(kd * conic_x), rel, 0.25f);
                        if (!(k > 0.0f)) {
                            area = 0;
                        } else if (hx < 1e6f && hy < 1e6f) {  // (false for NaN / inf: the reference's rect stays)
                            rminx = max(rminx, (int)floorf((pix_x - hx) * (1.0f / GSR_TILE)));
                            rminy = max(rminy, (int)floorf((pix_y - hy) * (1.0f / GSR_TILE)));
                            rmaxx = min(rmaxx, (int)floorf((pix_x + hx) * (1.0f / GSR_TILE)) + 1);
                            rmaxy = min(rmaxy, (int)floorf((pix_y + hy) * (1.0f / GSR_TILE)) + 1);
                            area = max(rmaxx - rminx, 0) * max(rmaxy - rminy, 0);
                        }
                    }
#endif
                }
            }
            if (area != 0) {
                float4 *rec = a.splat + 3 * (size_t)i;
                rec[0] = make_float4(pix_x, pix_y, third, 1.0f / vz);
                rec[1] = make_float4(conic_x, conic_y, conic_z, opacity);
                if (!a.infer) {  // (the 3D covariance is kept for the backward only)
                    float2 *cv = reinterpret_cast<float2 *>(a.cov3D + 6 * (size_t)i);
                    cv[0] = make_float2(c0, c1);
                    cv[1] = make_float2(c2, c3);
                    cv[2] = make_float2(c4, c5);
                }
                my_rect = make_uint2((uint32_t)rminx | ((uint32_t)rminy << 16),
                                     (uint32_t)rmaxx | ((uint32_t)rmaxy << 16));
                a.rects[i] = my_rect;
                touched = (uint32_t)area;
                visible = true;
                // fourth word of the colour record: the radius (unused downstream), or -- inference frames -- the tile
                // rect as four bytes, which the compositor tests its tile against (super-tile binning, render.hip)
                mypos = make_float4(px, py, pz,
                                    a.infer ? __uint_as_float((uint32_t)rminx | ((uint32_t)rminy << 8) |
                                                              ((uint32_t)rmaxx << 16) | ((uint32_t)rmaxy << 24))
                                            : fr);
            }
        }
    }
    GeomOut o;
    o.visible = visible;
    o.pos = mypos;
    o.tiles = touched;
    o.key = visible ? __float_as_uint(vz) : 0u;  // (vz > near_plane > 0: never the 0 pattern)
    o.rect = my_rect;
    o.radius = radius;
    return o;
}

// SH -> RGB (or the precomputed colour) of visible Gaussian g at position pp.xyz; writes rec2 and the clamp flags.
template <bool FAST_SH16>
__device__ __forceinline__ void prep_colour(const PreprocessArgs &a, const int g, const float4 pp) {
    float cr, cg, cb;
    uint32_t clamp_bits = 0;
    if (a.colors_precomp) {
        cr = a.colors_precomp[3 * (size_t)g];
        cg = a.colors_precomp[3 * (size_t)g + 1];
        cb = a.colors_precomp[3 * (size_t)g + 2];
    } else {
        float dx = pp.x - a.campos[0], dy = pp.y - a.campos[1], dz = pp.z - a.campos[2];
        const float len = sqrtf(fma_(dz, dz, fma_(dy, dy, dx * dx)));
        dx = dx / len; dy = dy / len; dz = dz / len;
        float b[16];
        sh_basis(a.D, dx, dy, dz, b);
        if (a.shs_rest) {
            // split storage (features_dc | features_rest): coefficient 0 from one array, 1.. from the other
            const float *dc = a.shs + 3 * (size_t)g;
            const float *rest = a.shs_rest + (size_t)g * (a.M - 1) * 3;
            const int nb = (a.D + 1) * (a.D + 1);
            cr = b[0] * dc[0]; cg = b[0] * dc[1]; cb = b[0] * dc[2];
            if (a.D == 3) {
                float f[45];
#pragma unroll
                for (int k = 0; k < 45; k++) f[k] = rest[k];  // 15 x dwordx3, all in flight together
#pragma unroll
                for (int k = 1; k < 16; k++) {
                    cr = fma_(b[k], f[3 * k - 3], cr);
                    cg = fma_(b[k], f[3 * k - 2], cg);
                    cb = fma_(b[k], f[3 * k - 1], cb);
                }
            } else {
                for (int k = 1; k < nb; k++) {
                    cr = fma_(b[k], rest[3 * k - 3], cr);
                    cg = fma_(b[k], rest[3 * k - 2], cg);
                    cb = fma_(b[k], rest[3 * k - 1], cb);
                }
            }
        } else if (FAST_SH16) {
            // D == 3, M == 16: 48 contiguous floats, 16-byte aligned -> 12 x dwordx4
            const float4 *sh4 = reinterpret_cast<const float4 *>(a.shs + (size_t)g * 48);
            float4 v[12];
#pragma unroll
            for (int k = 0; k < 12; k++) v[k] = sh4[k];
            const float *f = reinterpret_cast<const float *>(v);
            cr = b[0] * f[0]; cg = b[0] * f[1]; cb = b[0] * f[2];
#pragma unroll
            for (int k = 1; k < 16; k++) {
                cr = fma_(b[k], f[3 * k], cr);
                cg = fma_(b[k], f[3 * k + 1], cg);
                cb = fma_(b[k], f[3 * k + 2], cb);
            }
        } else {
            const float *sh = a.shs + (size_t)g * a.M * 3;
            const int nb = (a.D + 1) * (a.D + 1);
            cr = b[0] * sh[0]; cg = b[0] * sh[1]; cb = b[0] * sh[2];
            for (int k = 1; k < nb; k++) {
                cr = fma_(b[k], sh[3 * k], cr);
                cg = fma_(b[k], sh[3 * k + 1], cg);
                cb = fma_(b[k], sh[3 * k + 2], cb);
            }
        }
        cr += 0.5f; cg += 0.5f; cb += 0.5f;
        clamp_bits = (cr < 0.f ? 1u : 0u) | (cg < 0.f ? 0x100u : 0u) | (cb < 0.f ? 0x10000u : 0u);
        cr = fmaxf(cr, 0.f); cg = fmaxf(cg, 0.f); cb = fmaxf(cb, 0.f);
    }
    a.splat[3 * (size_t)g + 2] = make_float4(cr, cg, cb, pp.w);
    if (!a.infer) a.clamped[g] = clamp_bits;
}

// ---------------------------------------------------------------------------------------------------------
// View-frustum test of one block of GSR_BLOCK consecutive Gaussians (GsrInputs.cull_blocks).  true = NO Gaussian of
// the block can have a non-empty tile rect in this frame (or pass the near test), so the reference writes radii = 0 for
// every one of them and the block's workgroup has nothing to do.  Has to err towards false only.
//
// The block's record: box of the centres (lo, hi), rho >= sqrt(lambda_max(Sigma)) of every member (scale modifier 1),
// the common part label.  Lane c < 8 of the calling wave takes corner c through the part pose (affine), the view
// matrix (affine) and the projection; the members' centres lie in the convex hull of the corners all the way:
//   * view depth in [min, max] of the corners' -> all behind the near plane: done;
//   * with every corner in front (w > 0) the pixel coordinates are linear-fractional with a positive denominator on the
//     hull, so they lie between the corners' minima and maxima;
//   * radius: my_radius = ceil(3 sqrt(lambda_1)), lambda_1 = mid + sqrt(max(0.1, mid^2 - det)) <= (largest eigenvalue of
//     the dilated 2D covariance) + sqrt(0.1) = lambda_max(A Sigma A^T) + 0.3 + 0.3163 with A = J W, and
//     lambda_max(A Sigma A^T) <= |J|_2^2 |W|_2^2 lambda_max(Sigma); |W|_2^2 <= 1 + |W^T W - I|_F;
//     J J^T = [[a^2 + b^2, b d], [b d, c^2 + d^2]] with a = fx / vz, |b| <= fx limx / vz, c = fy / vz, |d| <= fy limy / vz
//     (|tx / vz| is clamped to limx = 1.3 tan(fov / 2)), so |J|_2^2 <= (max(fx^2 (1 + limx^2), fy^2 (1 + limy^2)) +
//     fx fy limx limy) / vz^2 (Gershgorin);
//   * the rect of a Gaussian is empty when pix + r < 1 or pix - r >= 16 x tiles on an axis (getRect's clamping).
// Margins: 1e-3 relative on the squared radius, 2 px on the radius, 1 px on the box -- the float error of the corner
// chain is ~1e-6 of its terms.  A corner that is not finite (NaN / infinite box or pose): not culled.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool prep_block_culled(const PreprocessArgs &a, int block) {
    const float4 b0 = *reinterpret_cast<const float4 *>(a.cull_blocks + 8 * (size_t)block);
    const float4 b1 = *reinterpret_cast<const float4 *>(a.cull_blocks + 8 * (size_t)block + 4);
    const int lane = gsr_lane();
    float px = (lane & 1) ? b0.w : b0.x, py = (lane & 2) ? b1.x : b0.y, pz = (lane & 4) ? b1.y : b0.z;
    float rho = b1.z;
    if (a.part_labels != nullptr) {
        const float lf = b1.w;
        if (!(lf == lf)) return false;  // members carry different labels: no common pose
        const int label = (int)lf;
        const int part = (label >= 0 && label < a.part_lut_size) ? a.part_lut[label] : -1;
        if (part >= 0 && part < a.part_count) {
            if (a.part_rescale != nullptr && a.part_rescale[part]) return false;  // (rewritten scales: rho does not hold)
            const float *xf = a.part_transforms + (size_t)part * 17;
            const float s = xf[12];
            px *= s; py *= s; pz *= s;
            const float rx = xf[0] * px + xf[1] * py + xf[2] * pz + xf[9];
            const float ry = xf[3] * px + xf[4] * py + xf[5] * pz + xf[10];
            const float rz = xf[6] * px + xf[7] * py + xf[8] * pz + xf[11];
            px = rx; py = ry; pz = rz;
            // R(q) = (1 - n) I + n R(q / |q|), n = |q|^2: the pose quaternion multiplies every member's, so the rotation's
            // norm grows by at most max(1, 2 n - 1)
            const float n = xf[13] * xf[13] + xf[14] * xf[14] + xf[15] * xf[15] + xf[16] * xf[16];
            rho *= fmaxf(1.0f, 2.0f * n - 1.0f) * 1.0001f;
        }
    }
    const float *m = a.view, *q = a.proj;
    const float vz = fma_(m[10], pz, fma_(m[6], py, m[2] * px)) + m[14];
    const float hx = fma_(q[8], pz, fma_(q[4], py, q[0] * px)) + q[12];
    const float hy = fma_(q[9], pz, fma_(q[5], py, q[1] * px)) + q[13];
    const float hw = fma_(q[11], pz, fma_(q[7], py, q[3] * px)) + q[15] + 0.0000001f;
    const float inv = 1.0f / hw;
    const float cx = fma_(hx * inv + 1.0f, (float)a.W, -1.0f) * 0.5f;
    const float cy = fma_(hy * inv + 1.0f, (float)a.H, -1.0f) * 0.5f;
    // (fminf / fmaxf below drop a NaN operand: a corner that is not finite -- a NaN or infinite box, a pose with one -- has
    // to be seen before the reduction; every group of eight lanes holds the same eight corners)
    const bool finite = fabsf(vz) < 1e30f && fabsf(hw) < 1e30f && fabsf(cx) < 1e30f && fabsf(cy) < 1e30f && rho == rho;
    if (__builtin_amdgcn_ballot_w64(!finite) != 0ull) return false;
    float zlo = vz, zhi = vz, wlo = hw, xlo = cx, xhi = cx, ylo = cy, yhi = cy;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        zlo = fminf(zlo, __shfl_xor(zlo, o, 64)); zhi = fmaxf(zhi, __shfl_xor(zhi, o, 64));
        wlo = fminf(wlo, __shfl_xor(wlo, o, 64));
        xlo = fminf(xlo, __shfl_xor(xlo, o, 64)); xhi = fmaxf(xhi, __shfl_xor(xhi, o, 64));
        ylo = fminf(ylo, __shfl_xor(ylo, o, 64)); yhi = fmaxf(yhi, __shfl_xor(yhi, o, 64));
    }
    const float zmargin = 1e-5f * (fabsf(zlo) + fabsf(zhi) + 1.0f);
    if (zhi < a.near_plane - zmargin) return true;  // every member fails `vz > near_plane`
    if (!(zlo > zmargin && wlo > 1e-6f)) return false;
    // |W|_2^2 <= 1 + |W^T W - I|_F for the view matrix' 3 x 3 block (W[i][j] = m[j * 4 + i])
    const float g00 = m[0] * m[0] + m[1] * m[1] + m[2] * m[2] - 1.0f, g11 = m[4] * m[4] + m[5] * m[5] + m[6] * m[6] - 1.0f;
    const float g22 = m[8] * m[8] + m[9] * m[9] + m[10] * m[10] - 1.0f;
    const float g01 = m[0] * m[4] + m[1] * m[5] + m[2] * m[6], g02 = m[0] * m[8] + m[1] * m[9] + m[2] * m[10];
    const float g12 = m[4] * m[8] + m[5] * m[9] + m[6] * m[10];
    const float wn2 = 1.0f + sqrtf(g00 * g00 + g11 * g11 + g22 * g22 + 2.0f * (g01 * g01 + g02 * g02 + g12 * g12));
    const float limx = 1.3f * a.tanfovx, limy = 1.3f * a.tanfovy;
    const float j22 = (fmaxf(a.fx * a.fx * (1.0f + limx * limx), a.fy * a.fy * (1.0f + limy * limy)) +
                       a.fx * a.fy * limx * limy) / ((zlo - zmargin) * (zlo - zmargin));
    const float sr = rho * fabsf(a.scale_modifier);
    const float rb = 3.0f * sqrtf(sr * sr * wn2 * j22 * 1.001f + 0.6163f) * 1.0001f + 2.0f;
    const float xend = (float)(GSR_TILE * a.gx) + 1.0f, yend = (float)(GSR_TILE * a.gy) + 1.0f;
    return xhi + rb < -1.0f || xlo - rb > xend || yhi + rb < -1.0f || ylo - rb > yend;
}

// grid = (blocks of 256 Gaussians, frames): blockIdx.y picks the frame's argument block (gsr_internal.h GsrBatch).
// (Measured in round 5 and not kept: the workgroups that read one block of the model for the B frames of a step as
// neighbours on one XCD, so that the model comes out of that L2 after the first frame: 57.3 -> 55.0 us for four identical
// frames, but 38.7 -> 42.3 us for the two different cameras of a closed-loop step.)
template <bool FAST_SH16, bool COUNT_TILES>
__global__ __launch_bounds__(GSR_BLOCK) void preprocess_kernel(const GsrBatch<PreprocessArgs> bt) {
    const PreprocessArgs a = bt.f[blockIdx.y];  // (by value: every field is requested at the top, not at its first use)
    const uint32_t blk = blockIdx.x;
    extern __shared__ uint32_t s_tcnt[];  // [num_tiles] when COUNT_TILES
    const int i = (int)blk * GSR_BLOCK + (int)threadIdx.x;
    bool visible = false;
    float4 mypos = make_float4(0.f, 0.f, 0.f, 0.f);  // xyz + radius, handed to the colour phase
    uint32_t my_tiles = 0, my_key = 0;
    uint2 my_rect = make_uint2(0u, 0u);
    if (COUNT_TILES) {
        for (int t = (int)threadIdx.x; t < a.num_tiles; t += GSR_BLOCK) s_tcnt[t] = 0u;
    }
    // the Gaussian's number in the caller's arrays (radii)
    const int oi = (a.orig_index != nullptr && a.radii != nullptr && i < a.P) ? a.orig_index[i] : i;
    if (a.cull_blocks != nullptr) {
        // (the first wave evaluates the block's test, ~100 instructions; the other three wait for its verdict at a
        // barrier instead of repeating it: a skipped workgroup costs ~150 wave-instructions instead of ~400, and seven
        // in ten are skipped)
        __shared__ int s_culled;
        if (threadIdx.x < GSR_WAVE) {
            const bool c = prep_block_culled(a, (int)blk);
            if (threadIdx.x == 0) s_culled = c ? 1 : 0;
        }
        __syncthreads();
        if (s_culled != 0) {
            if (i < a.P) {
                if (a.radii != nullptr) a.radii[oi] = 0;
                if (!a.infer) a.tiles_touched[i] = 0u;
            }
            if (threadIdx.x == 0) a.block_counts[blk] = 0u;
            return;
        }
    }
    if (i < a.P) {
        const GeomOut o = prep_geometry(a, i);
        visible = o.visible;
        mypos = o.pos;
        my_tiles = o.tiles;
        my_rect = o.rect;
        if (a.radii != nullptr) a.radii[oi] = o.radius;
        if (!a.infer) a.tiles_touched[i] = o.tiles;
        my_key = o.key;
    }

    // ---- phase 2: colour, on the block-compacted list of survivors ------------------------------------------
    // Typically only a fraction of the 256 lanes survive cull + rect; evaluating the SH (48 loads + ~110 VALU per
    // Gaussian) in place would run all 4 waves at that fraction of their lanes.  Dense lanes instead.
    __shared__ uint32_t s_w[4];
    __shared__ float4 s_pos[GSR_BLOCK];
    __shared__ int s_idx[GSR_BLOCK];
    uint32_t cnt;
    const uint32_t incl = gsr_block_incl_scan(visible ? 1u : 0u, s_w, cnt);
    if (visible) {
        s_pos[incl - 1u] = mypos;
        s_idx[incl - 1u] = i;
        a.block_recs[(size_t)blk * GSR_BLOCK + (incl - 1u)] = make_uint2((uint32_t)i, my_key);
    }
    if (threadIdx.x == 0) {
        a.block_counts[blk] = cnt;  // consumed by the index-ordered compaction (depth-sorted paths)
        if (COUNT_TILES && cnt != 0u) atomicAdd(&a.hdr->V, cnt);
    }
    __syncthreads();
    if (COUNT_TILES) {
        // per-workgroup tile histogram in LDS, then one global add per touched tile
        uint32_t *cnt_lds = s_tcnt;
        gsr_for_each_tile(my_tiles, my_rect, a.gx, 0u, 0u,
                          [cnt_lds](uint32_t tile, uint32_t, uint32_t) { atomicAdd(&cnt_lds[tile], 1u); });
        __syncthreads();
        for (int t = (int)threadIdx.x; t < a.num_tiles; t += GSR_BLOCK) {
            const uint32_t c = s_tcnt[t];
            if (c != 0u) atomicAdd(&a.tile_accum[(size_t)(blk % GSR_BIN_SLOTS) * a.num_tiles + t], c);
        }
    }
    if (threadIdx.x < cnt) {
        prep_colour<FAST_SH16>(a, s_idx[threadIdx.x], s_pos[threadIdx.x]);
    }
}

__global__ __launch_bounds__(GSR_BLOCK) void mark_visible_kernel(int P, const float *means3D, const float *m,
                                                                 float near_plane, uint8_t *present) {
    const int i = blockIdx.x * GSR_BLOCK + threadIdx.x;
    if (i >= P) return;
    const float px = means3D[3 * (size_t)i], py = means3D[3 * (size_t)i + 1], pz = means3D[3 * (size_t)i + 2];
    const float vz = fma_(m[10], pz, fma_(m[6], py, m[2] * px)) + m[14];
    present[i] = vz > near_plane ? 1 : 0;
}

}  // namespace

int gsr_launch_preprocess(int B, const GsrFrame *fr, bool count_tiles, bool infer, hipStream_t stream) {
    GsrBatch<PreprocessArgs> bt;
    bool fast = true;
    for (int k = 0; k < B; k++) {
        const GsrSettings &st = *fr[k].st;
        const GsrInputs &in = *fr[k].in;
        const GeomState &g = fr[k].g;
        PreprocessArgs &a = bt.f[k];
        a.infer = infer ? 1 : 0;
        a.P = in.P;
        a.D = st.sh_degree;
        a.M = st.sh_coeffs;
        a.W = st.image_width;
        a.H = st.image_height;
        a.gx = gsr_div_up(a.W, GSR_TILE);
        a.gy = gsr_div_up(a.H, GSR_TILE);
        a.tanfovx = st.tanfovx;
        a.tanfovy = st.tanfovy;
        a.fx = (float)a.W / (2.0f * st.tanfovx);
        a.fy = (float)a.H / (2.0f * st.tanfovy);
        a.scale_modifier = st.scale_modifier;
        a.near_plane = st.near_plane;
        a.antialiasing = st.antialiasing;
        a.means3D = in.means3D;
        a.shs = in.shs;
        a.shs_rest = in.shs_rest;
        a.param_space = in.param_space;
        a.colors_precomp = in.colors_precomp;
        a.opacities = in.opacities;
        a.scales = in.scales;
        a.rotations = in.rotations;
        a.cov3D_precomp = in.cov3D_precomp;
        a.view = in.viewmatrix;
        a.proj = in.projmatrix;
        a.campos = in.campos;
        a.part_labels = in.part_labels;
        a.part_lut = in.part_lut;
        a.part_lut_size = in.part_lut_size;
        a.part_transforms = in.part_transforms;
        a.part_count = in.part_count;
        a.part_rescale = in.part_rescale;
        a.cull_blocks = in.cull_blocks;
        a.orig_index = in.orig_index;
        a.radii = fr[k].out->radii;
        a.splat = g.splat;
        a.cov3D = g.cov3D;
        a.clamped = g.clamped;
        a.tiles_touched = g.tiles_touched;
        a.rects = g.rects;
        a.block_counts = g.block_counts;
        a.block_recs = g.pair[1];  // (the sort's compaction gathers from here into pair[0]; its partition pass then
                                   //  overwrites this array with the bucketed records)

        a.num_tiles = a.gx * a.gy;
        a.tile_accum = g.tile_accum;
        a.hdr = g.hdr;
        // (the 12 x dwordx4 colour path needs every frame's SH array aligned)
        fast = fast && (in.colors_precomp == nullptr) && st.sh_degree == 3 && st.sh_coeffs == 16 &&
               ((reinterpret_cast<uintptr_t>(in.shs) & 15u) == 0);
    }
    const PreprocessArgs &a0 = bt.f[0];
    const dim3 grid(GeomState::prep_blocks(a0.P), B);
    const size_t lds = count_tiles ? (size_t)a0.num_tiles * sizeof(uint32_t) : 0;
    if (count_tiles) {
        if (fast)
            hipLaunchKernelGGL((preprocess_kernel<true, true>), grid, dim3(GSR_BLOCK), lds, stream, bt);
        else
            hipLaunchKernelGGL((preprocess_kernel<false, true>), grid, dim3(GSR_BLOCK), lds, stream, bt);
    } else {
        if (fast)
            hipLaunchKernelGGL((preprocess_kernel<true, false>), grid, dim3(GSR_BLOCK), 0, stream, bt);
        else
            hipLaunchKernelGGL((preprocess_kernel<false, false>), grid, dim3(GSR_BLOCK), 0, stream, bt);
    }
    return GSR_OK;
}

extern "C" int gsr_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, float near_plane,
                                uint8_t *present, void *stream) {
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) {
        gsr_set_error("gsr_mark_visible: null pointer or negative P");
        return GSR_E_INVALID;
    }
    if (P == 0) return GSR_OK;
    hipLaunchKernelGGL(mark_visible_kernel, dim3(gsr_div_up(P, GSR_BLOCK)), dim3(GSR_BLOCK), 0,
                       (hipStream_t)stream, P, means3D, viewmatrix, near_plane, present);
    return gsr_check_launch("mark_visible", false, (hipStream_t)stream);
}
