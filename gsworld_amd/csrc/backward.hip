// backward.hip -- gradient of the rasterizer (upstream backward.cu: renderCUDA, computeCov2DCUDA, preprocessCUDA;
// SURVEY.md 8a row A10, Appendix B.8).
//
// CDNA4 mapping of the compositing backward: one 16x16 tile per 256-thread workgroup, instances replayed back
// to front from LDS.  Upstream issues 9-10 global float atomics per contributing (pixel, instance); here the 10
// partials are first summed across the 64 lanes of each wave with DPP row shifts / row broadcasts (no LDS
// traffic), then across the 4 waves of the tile with one LDS atomic each, and only then flushed with ONE global
// atomic per (tile, instance, component): 256x fewer global atomics on the hot addresses.
#include "gsr_internal.h"

namespace {

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Sum over the 64 lanes of the wave; the total is returned in every lane (via readlane 63).
// DPP controls (GFX9): row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143.
__device__ __forceinline__ float wave_sum(float v) {
#if defined(GSR_SAFE_WAVE_SUM)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
#else
    int x;
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, false);
    v += __int_as_float(x);
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, false);
    v += __int_as_float(x);
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, false);
    v += __int_as_float(x);
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, false);
    v += __int_as_float(x);  // lane 15 of every row holds its row total
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false);
    v += __int_as_float(x);  // rows 1 and 3 += lane 15 of the previous row
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false);
    v += __int_as_float(x);  // rows 2 and 3 += lane 31
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
#endif
}

#ifndef GSR_BWD_DIAG
#define GSR_BWD_DIAG 0
#endif
constexpr int kGradRec = 12;  // words of a Gaussian's record in GeomState::grad_rec (kGrad sums + 2 unused)
constexpr int kGrad = 10;  // mean2D.x, mean2D.y, conic.xx, conic.xy(half), conic.yy, opacity, r, g, b, invdepth

typedef unsigned v2u __attribute__((ext_vector_type(2)));

// Wave totals of 10 per-lane values at once, as a transpose-reduce: v_permlane32_swap exchanges the upper half of
// one register with the lower half of another, so ONE swap + ONE add folds lane l with lane l + 32 for TWO values
// (lanes 0-31 then carry the first, 32-63 the second); v_permlane16_swap does the same across rows of 16 for two
// such registers (four values, one per row); the last four levels are DPP adds inside the rows.  28 instructions
// instead of the 10 x 13 of ten separate wave_sum calls.  Lanes (row r = lane / 16, column s = lane % 16) end up with
//   s == 0: components 0, 2, 1, 3 in rows 0..3     s == 1: components 4, 6, 5, 7     s == 2: 8, 8, 9, 9
// wave_reduce10_component gives that map; the value in any other lane is not meaningful.
__device__ __forceinline__ float fold32(float a, float b) {
    const v2u r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float fold16(float a, float b) {
    const v2u r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float row_sum(float v) {  // every lane of a row of 16 gets the row total
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));  // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));  // row_mirror
    return v;
}
__device__ __forceinline__ float wave_reduce10(const float (&v)[kGrad]) {
    const float z0 = fold32(v[0], v[1]), z1 = fold32(v[2], v[3]), z2 = fold32(v[4], v[5]), z3 = fold32(v[6], v[7]),
                z4 = fold32(v[8], v[9]);
    const float w0 = row_sum(fold16(z0, z1)), w1 = row_sum(fold16(z2, z3)), w2 = row_sum(fold16(z4, z4));
    const int s = gsr_lane() & 15;
    return s == 0 ? w0 : s == 1 ? w1 : w2;
}
// component held by this lane after wave_reduce10, or -1
__device__ __forceinline__ int wave_reduce10_component() {
    const int lane = gsr_lane(), r = lane >> 4, s = lane & 15;
    if (s < 2) return 4 * s + (((r & 1) << 1) | (r >> 1));
    if (s == 2 && !(r & 1)) return 8 + (r >> 1);
    return -1;
}

__global__ __launch_bounds__(GSR_BLOCK) void render_backward_kernel(
    const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list, const float4 *__restrict__ splat,
    int W, int H, int gx, const float *__restrict__ bg, const float *__restrict__ final_T,
    const uint32_t *__restrict__ n_contrib, const float *__restrict__ dL_dpix,
    const float *__restrict__ dL_dinvdepth_pix, GsrGradWord *__restrict__ grad_rec /*(P,12): the kGrad sums + 2 unused*/) {
    __shared__ float4 s_rec0[GSR_BLOCK];
    __shared__ float4 s_rec1[GSR_BLOCK];
    __shared__ float4 s_rec2[GSR_BLOCK];
    __shared__ uint32_t s_id[GSR_BLOCK];
    __shared__ float s_grad[GSR_BLOCK * kGradRec];  // [staged instance][component]
    __shared__ uint32_t s_max[4];

    // (Measured in round 6 and not kept: the workgroups in the forward's compositing order, longest lists first -- 107.1
    //  against 107.4 us at configs[4]: the kernel is not bound by its tail.  What it is bound by, from builds that leave
    //  parts out (GSR_BWD_DIAG, profiles/round6/README.md): staging + cull + flush loop 12.6 us, the walk's evaluations
    //  42 us, the wave sums + LDS atomics 50 us, the global atomics 5 us.)
    const int tile = (int)blockIdx.x;
    const int tile_x = tile % gx, tile_y = tile / gx;
    const int lane = gsr_lane(), wave = gsr_wave();
    const int lx = ((wave & 1) << 3) | (lane & 7);
    const int ly = ((wave >> 1) << 3) | (lane >> 3);
    const int px = tile_x * GSR_TILE + lx, py = tile_y * GSR_TILE + ly;
    const bool inside = px < W && py < H;
    const float pfx = (float)px, pfy = (float)py;
    const float qxf = (float)(tile_x * GSR_TILE + ((wave & 1) << 3)), qyf = (float)(tile_y * GSR_TILE + ((wave >> 1) << 3));
    const size_t pid = (size_t)py * W + px;
    const size_t plane = (size_t)H * W;

    const uint2 range = ranges[tile];
    const float T_final = inside ? final_T[pid] : 0.f;
    const uint32_t last_contributor = inside ? n_contrib[pid] : 0u;
    float T = T_final;
    float dLp0 = 0.f, dLp1 = 0.f, dLp2 = 0.f, dLd = 0.f;
    if (inside) {
        dLp0 = dL_dpix[pid];
        dLp1 = dL_dpix[plane + pid];
        dLp2 = dL_dpix[2 * plane + pid];
        if (dL_dinvdepth_pix) dLd = dL_dinvdepth_pix[pid];
    }
    const float bg_dot = fma_(bg[2], dLp2, fma_(bg[1], dLp1, bg[0] * dLp0));
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;

    // only instances [0, max over the tile of n_contrib) were ever blended
    {
        uint32_t m = last_contributor;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
        if (lane == 0) s_max[wave] = m;
    }
    __syncthreads();
    const int n_inst = (int)max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
    const int rounds = (n_inst + GSR_BLOCK - 1) / GSR_BLOCK;

    const int my_comp = wave_reduce10_component();
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, accd = 0.f;
    float lastc0 = 0.f, lastc1 = 0.f, lastc2 = 0.f, lastd = 0.f, last_alpha = 0.f;

    for (int rd = 0; rd < rounds; rd++) {
        // stage instances n_inst-1-rd*256-k, k = 0..255 (back to front)
        __syncthreads();
        const int k = rd * GSR_BLOCK + (int)threadIdx.x;
        const int idx = n_inst - 1 - k;
        if (idx >= 0) {
            const uint32_t g = point_list[range.x + (uint32_t)idx];
            const float4 *rec = splat + 3 * (size_t)g;
            s_id[threadIdx.x] = g;
            s_rec0[threadIdx.x] = rec[0];
            s_rec1[threadIdx.x] = rec[1];
            s_rec2[threadIdx.x] = rec[2];
        }
#pragma unroll
        for (int c = 0; c < kGradRec; c++) s_grad[c * GSR_BLOCK + threadIdx.x] = 0.f;
        __syncthreads();
        const int cnt = min(GSR_BLOCK, n_inst - rd * GSR_BLOCK);
        // The forward's per-quadrant cull applies unchanged: an instance that cannot reach alpha >= 1/255 on any pixel
        // of this wave's 8 x 8 quadrant is one every lane would skip below, and ~6 of 7 staged instances are of that
        // kind.  64 of them are tested at a time (one per lane), then only the survivors are walked, in order.
        for (int j0 = 0; j0 < cnt; j0 += GSR_WAVE) {
        uint64_t todo;
        {
            const int jj = j0 + lane;
            bool may = jj < cnt;
            if (may) {
                const float4 c0 = s_rec0[jj];
                may = quadrant_may_hit(c0.x, c0.y, s_rec1[jj], qxf, qyf);
            }
            todo = __builtin_amdgcn_ballot_w64(may);
#if GSR_BWD_DIAG == 3  // (timing diagnostics only: staging, cull and flush without the walk)
            todo = todo == 0x123456789abcull ? 1ull : 0ull;
#endif
        }
        while (todo != 0ull) {
            const int j = j0 + (int)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            const uint32_t contributor = (uint32_t)(n_inst - 1 - (rd * GSR_BLOCK + j));  // 0-based position
            float g_mx = 0.f, g_my = 0.f, g_cxx = 0.f, g_cxy = 0.f, g_cyy = 0.f, g_op = 0.f, g_r = 0.f, g_g = 0.f,
                  g_b = 0.f, g_d = 0.f;
            bool hit = false;
            if (contributor < last_contributor) {
                const float4 r0 = s_rec0[j];
                const float4 r1 = s_rec1[j];
                const float dx = r0.x - pfx, dy = r0.y - pfy;
                const float q = fma_(r1.z * dy, dy, (r1.x * dx) * dx);
                const float power = fma_(-(r1.y * dx), dy, -0.5f * q);
                if (power <= 0.0f) {
                    const float G = gsr_exp_power(power);
                    const float alpha = fminf(0.99f, r1.w * G);
                    if (alpha >= 1.0f / 255.0f) {
                        hit = true;
                        const float4 r2 = s_rec2[j];
                        // 1 / (1 - alpha) once, by the hardware reciprocal (1 ulp), for both quotients below: the two
                        // correctly rounded divisions were a quarter of this path's instructions, and the gradients
                        // are compared at 2e-3 of their scale (T drifts by ~1e-7 per step, sqrt(n) of them)
                        const float inv_1ma = __builtin_amdgcn_rcpf(1.f - alpha);
                        T = T * inv_1ma;
                        const float dchannel_dcolor = alpha * T;
                        float dL_dalpha;
                        acc0 = fma_(last_alpha, lastc0, (1.f - last_alpha) * acc0);
                        lastc0 = r2.x;
                        dL_dalpha = (r2.x - acc0) * dLp0;
                        acc1 = fma_(last_alpha, lastc1, (1.f - last_alpha) * acc1);
                        lastc1 = r2.y;
                        dL_dalpha = fma_(r2.y - acc1, dLp1, dL_dalpha);
                        acc2 = fma_(last_alpha, lastc2, (1.f - last_alpha) * acc2);
                        lastc2 = r2.z;
                        dL_dalpha = fma_(r2.z - acc2, dLp2, dL_dalpha);
                        g_r = dchannel_dcolor * dLp0;
                        g_g = dchannel_dcolor * dLp1;
                        g_b = dchannel_dcolor * dLp2;
                        accd = fma_(last_alpha, lastd, (1.f - last_alpha) * accd);
                        lastd = r0.w;
                        dL_dalpha = fma_(r0.w - accd, dLd, dL_dalpha);
                        g_d = dchannel_dcolor * dLd;
                        dL_dalpha *= T;
                        last_alpha = alpha;
                        dL_dalpha = fma_(-T_final * inv_1ma, bg_dot, dL_dalpha);
                        const float dL_dG = r1.w * dL_dalpha;
                        const float gdx = G * dx, gdy = G * dy;
                        const float dG_ddelx = fma_(-gdy, r1.y, -gdx * r1.x);
                        const float dG_ddely = fma_(-gdx, r1.y, -gdy * r1.z);
                        g_mx = (dL_dG * dG_ddelx) * ddelx_dx;
                        g_my = (dL_dG * dG_ddely) * ddely_dy;
                        g_cxx = (-0.5f * gdx) * dx * dL_dG;
                        g_cxy = (-0.5f * gdx) * dy * dL_dG;
                        g_cyy = (-0.5f * gdy) * dy * dL_dG;
                        g_op = G * dL_dalpha;
                    }
                }
            }
            if (__ballot(hit) == 0ull) continue;  // nothing in this wave touches the instance
            const float gv[kGrad] = {g_mx, g_my, g_cxx, g_cxy, g_cyy, g_op, g_r, g_g, g_b, g_d};
#if GSR_BWD_DIAG == 2  // (timing diagnostics only -- wrong gradients: no wave sums)
            if (my_comp >= 0) s_grad[j * kGradRec + my_comp] = g_mx + g_op + g_r + g_d + g_cxy;
            continue;
#endif
            const float total = wave_reduce10(gv);
            if (my_comp >= 0) atomicAdd(&s_grad[j * kGradRec + my_comp], total);  // LDS: at most 4 waves per address
        }
        }
        __syncthreads();
        // flush: ONE device-scope atomic per (tile, instance, component), issued TRANSPOSED -- twelve consecutive lanes
        // own the twelve words of one instance's record, so the ten sums of a Gaussian leave as one request to one
        // 48-byte record instead of ten requests to five arrays (the atomics resolve memory-side: what they cost is
        // requests, not words; thread-per-instance flushing into separate arrays was a third of this kernel)
#pragma unroll
        for (int u = 0; u < kGradRec; u++) {
            const int e = u * GSR_BLOCK + (int)threadIdx.x;
            const int j = e / kGradRec, c = e - j * kGradRec;
            if (j < cnt && c < kGrad) {
                const float v = s_grad[e];
                // (binary64 from here on: what a Gaussian collects from its tiles no longer depends on their order)
#if GSR_BWD_DIAG == 1 || GSR_BWD_DIAG == 2  // (timing diagnostics only: the sums stay in the LDS)
                if (v == 12345.678f) grad_rec[0] = (GsrGradWord)v;
#else
                if (v != 0.f) atomicAdd(&grad_rec[(size_t)kGradRec * s_id[j] + c], (GsrGradWord)v);
#endif
            }
        }
    }
}

constexpr float kC0 = 0.28209479177387814f;
constexpr float kC1 = 0.4886025119029199f;
constexpr float kC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                          0.5462742152960396f};
constexpr float kC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                          -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

struct BwdArgs {
    int P, D, M, W, H;
    float tanfovx, tanfovy, fx, fy, scale_modifier;
    int antialiasing, cov_precomp, colors_precomp, have_invdepth;
    int param_space;           // GSR_RAW_*: the inputs are raw parameters, the gradients are returned w.r.t. them
    const float *means3D, *shs, *opacities, *scales, *rotations, *view, *proj, *campos;
    const float *shs_rest;     // split SH storage: shs = (P,1,3) dc, shs_rest = (P,M-1,3)
    const int32_t *radii;
    const float *cov3D;        // precomputed input or the forward's stored copy
    const uint32_t *clamped;
    const GsrGradWord *grad_rec;  // (P,12) the compositor's sums (render_backward_kernel)
    float *dL_dmean2D, *dL_dcolors;  // written from the record (API outputs)
    float *dL_dopacity, *dL_dmeans3D, *dL_dcov3D, *dL_dsh, *dL_dscales, *dL_drots;
    float *dL_dsh_rest;        // with shs_rest: dL_dsh is the dc part
};

// Per-Gaussian chain rule, one thread per Gaussian, in two kernels: the geometry (2D mean, EWA covariance, 3D
// covariance -> scale / quaternion, opacity) here, the SH colour below.  As one kernel it held 180 VGPRs (two waves per
// SIMD) and moved every Gaussian's 2 x 180 bytes of SH through per-lane strided accesses.
// The kernel writes EVERY element of its outputs -- zeros for the Gaussians the forward culled (round 6): gsr_backward used
// to clear the seven arrays with a memset in front of every backward, 46 MB per step at configs[4] size for a kernel that
// then rewrote nearly all of it.  (Measured with it and not kept: the record array cleared the same way -- by the
// forward for the Gaussians it makes visible, by this kernel behind its read -- instead of by its memset: preprocess
// 42 -> 52 us and this kernel 24 -> 45 us for the 25 us the two memsets took.)
__global__ __launch_bounds__(GSR_BLOCK) void geometry_backward_kernel(const BwdArgs a) {
    const int i = blockIdx.x * GSR_BLOCK + (int)threadIdx.x;
    if (i >= a.P) return;
    if (a.radii[i] <= 0) {
        a.dL_dopacity[i] = 0.f;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            a.dL_dmean2D[3 * (size_t)i + k] = 0.f;
            a.dL_dcolors[3 * (size_t)i + k] = 0.f;
            a.dL_dmeans3D[3 * (size_t)i + k] = 0.f;
            if (!a.cov_precomp) a.dL_dscales[3 * (size_t)i + k] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < 6; k++) a.dL_dcov3D[6 * (size_t)i + k] = 0.f;
        if (!a.cov_precomp) *reinterpret_cast<float4 *>(a.dL_drots + 4 * (size_t)i) = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    // (mx, my, cxx, cxy) (cyy, opacity, r, g) (b, 1/depth, -, -): the sums, rounded to binary32 once
    float4 g0, g1, g2;
    {
#if GSR_GRAD_F64
        const double2 *rec = reinterpret_cast<const double2 *>(a.grad_rec + 12 * (size_t)i);
        const double2 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3], r4 = rec[4];
        g0 = make_float4((float)r0.x, (float)r0.y, (float)r1.x, (float)r1.y);
        g1 = make_float4((float)r2.x, (float)r2.y, (float)r3.x, (float)r3.y);
        g2 = make_float4((float)r4.x, (float)r4.y, 0.f, 0.f);
#else
        const float4 *rec = reinterpret_cast<const float4 *>(a.grad_rec) + 3 * (size_t)i;
        g0 = rec[0]; g1 = rec[1]; g2 = rec[2];
#endif
    }
    const float *m = a.view;
    const float px = a.means3D[3 * (size_t)i], py = a.means3D[3 * (size_t)i + 1], pz = a.means3D[3 * (size_t)i + 2];
    const float *c6 = a.cov3D + 6 * (size_t)i;
    const float tx0 = fma_(m[8], pz, fma_(m[4], py, m[0] * px)) + m[12];
    const float ty0 = fma_(m[9], pz, fma_(m[5], py, m[1] * px)) + m[13];
    const float tz = fma_(m[10], pz, fma_(m[6], py, m[2] * px)) + m[14];
    const float limx = 1.3f * a.tanfovx, limy = 1.3f * a.tanfovy;
    const float txtz = tx0 / tz, tytz = ty0 / tz;
    const float tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
    const float ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
    const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    const float fx = a.fx, fy = a.fy;
    const float J00 = fx / tz, J02 = -(fx * tx) / (tz * tz), J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
    float A[2][3], Wm[3][3];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) Wm[r][c] = m[c * 4 + r];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        A[0][j] = fma_(J02, Wm[2][j], J00 * Wm[0][j]);
        A[1][j] = fma_(J12, Wm[2][j], J11 * Wm[1][j]);
    }
    const float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    float B[2][3];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int j = 0; j < 3; j++) B[r][j] = fma_(A[r][2], S[2][j], fma_(A[r][1], S[1][j], A[r][0] * S[0][j]));
    float ca = fma_(B[0][2], A[0][2], fma_(B[0][1], A[0][1], B[0][0] * A[0][0]));
    const float cb = fma_(B[0][2], A[1][2], fma_(B[0][1], A[1][1], B[0][0] * A[1][0]));
    float cc = fma_(B[1][2], A[1][2], fma_(B[1][1], A[1][1], B[1][0] * A[1][0]));
    const float h_var = 0.3f;
    float dL_da_aa = 0.f, dL_db_aa = 0.f, dL_dc_aa = 0.f;
    // raw parameter space (GsrInputs.param_space): the same canonical activations as the forward, then their chain rule
    const bool raw_opacity = (a.param_space & GSR_RAW_OPACITY) != 0;
    float opacity_act = 0.f;
    if (a.antialiasing || raw_opacity) opacity_act = raw_opacity ? sigmoid_canonical(a.opacities[i]) : a.opacities[i];
    const float dL_dop = g1.y;
    float dL_dop_out = raw_opacity ? dL_dop * (opacity_act * (1.0f - opacity_act)) : dL_dop;
    if (a.antialiasing) {
        const float det_cov = fma_(-cb, cb, ca * cc);
        ca += h_var;
        cc += h_var;
        const float det_plus = fma_(-cb, cb, ca * cc);
        const float ratio = det_cov / det_plus;
        const float h_scale = sqrtf(fmaxf(0.000025f, ratio));
        const float d_h = dL_dop * opacity_act;
        dL_dop_out = raw_opacity ? (dL_dop * h_scale) * (opacity_act * (1.0f - opacity_act)) : dL_dop * h_scale;
        const float d_root = ratio <= 0.000025f ? 0.f : d_h / (2.f * h_scale);
        const float inv2 = 1.f / (det_plus * det_plus);
        dL_da_aa = d_root * ((cc - h_var) * det_plus - det_cov * cc) * inv2;
        dL_dc_aa = d_root * ((ca - h_var) * det_plus - det_cov * ca) * inv2;
        dL_db_aa = d_root * (-2.f * cb * det_plus + 2.f * cb * det_cov) * inv2;
    } else {
        ca += h_var;
        cc += h_var;
    }
    a.dL_dopacity[i] = dL_dop_out;
    a.dL_dmean2D[3 * (size_t)i] = g0.x;
    a.dL_dmean2D[3 * (size_t)i + 1] = g0.y;
    a.dL_dmean2D[3 * (size_t)i + 2] = 0.f;
    a.dL_dcolors[3 * (size_t)i] = g1.z;
    a.dL_dcolors[3 * (size_t)i + 1] = g1.w;
    a.dL_dcolors[3 * (size_t)i + 2] = g2.x;
    const float Lx = g0.z, Ly = g0.w, Lz = g1.x;
    const float denom = fma_(-cb, cb, ca * cc);
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
    if (denom2inv != 0.f) {
        dL_da = denom2inv * (-cc * cc * Lx + 2.f * cb * cc * Ly + (denom - ca * cc) * Lz);
        dL_dc = denom2inv * (-ca * ca * Lz + 2.f * ca * cb * Ly + (denom - ca * cc) * Lx);
        dL_db = denom2inv * 2.f * (cb * cc * Lx - (denom + 2.f * cb * cb) * Ly + ca * cb * Lz);
    }
    dL_da += dL_da_aa;
    dL_db += dL_db_aa;
    dL_dc += dL_dc_aa;
    float dS[6];
    dS[0] = A[0][0] * A[0][0] * dL_da + A[0][0] * A[1][0] * dL_db + A[1][0] * A[1][0] * dL_dc;
    dS[3] = A[0][1] * A[0][1] * dL_da + A[0][1] * A[1][1] * dL_db + A[1][1] * A[1][1] * dL_dc;
    dS[5] = A[0][2] * A[0][2] * dL_da + A[0][2] * A[1][2] * dL_db + A[1][2] * A[1][2] * dL_dc;
    dS[1] = 2.f * A[0][0] * A[0][1] * dL_da + (A[0][0] * A[1][1] + A[0][1] * A[1][0]) * dL_db +
            2.f * A[1][0] * A[1][1] * dL_dc;
    dS[2] = 2.f * A[0][0] * A[0][2] * dL_da + (A[0][0] * A[1][2] + A[0][2] * A[1][0]) * dL_db +
            2.f * A[1][0] * A[1][2] * dL_dc;
    dS[4] = 2.f * A[0][2] * A[0][1] * dL_da + (A[0][1] * A[1][2] + A[0][2] * A[1][1]) * dL_db +
            2.f * A[1][1] * A[1][2] * dL_dc;
#pragma unroll
    for (int k = 0; k < 6; k++) a.dL_dcov3D[6 * (size_t)i + k] = dS[k];
    float dA[2][3];
#pragma unroll
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
    if (a.have_invdepth) dtz -= g2.y / (tz * tz);
    float dmean[3];
#pragma unroll
    for (int j = 0; j < 3; j++) dmean[j] = m[j * 4 + 0] * dtx + m[j * 4 + 1] * dty + m[j * 4 + 2] * dtz;
    // projected mean
    const float *q = a.proj;
    const float hx = fma_(q[8], pz, fma_(q[4], py, q[0] * px)) + q[12];
    const float hy = fma_(q[9], pz, fma_(q[5], py, q[1] * px)) + q[13];
    const float hw = fma_(q[11], pz, fma_(q[7], py, q[3] * px)) + q[15];
    const float m_w = 1.0f / (hw + 0.0000001f);
    const float mul1 = hx * m_w * m_w, mul2 = hy * m_w * m_w;
    const float g2x = g0.x, g2y = g0.y;
    dmean[0] += (q[0] * m_w - q[3] * mul1) * g2x + (q[1] * m_w - q[3] * mul2) * g2y;
    dmean[1] += (q[4] * m_w - q[7] * mul1) * g2x + (q[5] * m_w - q[7] * mul2) * g2y;
    dmean[2] += (q[8] * m_w - q[11] * mul1) * g2x + (q[9] * m_w - q[11] * mul2) * g2y;
    a.dL_dmeans3D[3 * (size_t)i] = dmean[0];  // (sh_backward_kernel adds the view-direction term of the colour)
    a.dL_dmeans3D[3 * (size_t)i + 1] = dmean[1];
    a.dL_dmeans3D[3 * (size_t)i + 2] = dmean[2];
    // 3D covariance -> scale, quaternion
    if (!a.cov_precomp) {
        float4 rq = *reinterpret_cast<const float4 *>(a.rotations + 4 * (size_t)i);
        float q_norm = 1.0f;
        if (a.param_space & GSR_RAW_ROTATIONS) {  // F.normalize, as in the forward
            const float n2 = fma_(rq.w, rq.w, fma_(rq.z, rq.z, fma_(rq.y, rq.y, rq.x * rq.x)));
            q_norm = fmaxf(sqrtf(n2), 1e-12f);
            rq = make_float4(rq.x / q_norm, rq.y / q_norm, rq.z / q_norm, rq.w / q_norm);
        }
        const float r = rq.x, x = rq.y, y = rq.z, z = rq.w;
        float R[3][3];
        R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
        R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
        R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
        float s[3], sc[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            sc[k] = a.scales[3 * (size_t)i + k];
            if (a.param_space & GSR_RAW_SCALES) sc[k] = exp_canonical(sc[k]);
            s[k] = a.scale_modifier * sc[k];
        }
        float Mm[3][3];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int j = 0; j < 3; j++) Mm[k][j] = s[k] * R[j][k];
        const float Gs[3][3] = {{dS[0], 0.5f * dS[1], 0.5f * dS[2]},
                                {0.5f * dS[1], dS[3], 0.5f * dS[4]},
                                {0.5f * dS[2], 0.5f * dS[4], dS[5]}};
        float dM[3][3], dR[3][3];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                dM[k][j] = 2.f * (Mm[k][0] * Gs[0][j] + Mm[k][1] * Gs[1][j] + Mm[k][2] * Gs[2][j]);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float dsk = R[0][k] * dM[k][0] + R[1][k] * dM[k][1] + R[2][k] * dM[k][2];
            a.dL_dscales[3 * (size_t)i + k] = (a.param_space & GSR_RAW_SCALES) ? dsk * sc[k] : dsk;  // d exp(x) = exp(x)
#pragma unroll
            for (int j = 0; j < 3; j++) dR[j][k] = s[k] * dM[k][j];
        }
        float4 dq;
        dq.x = 2.f * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
        dq.y = 2.f * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - 2.f * x * dR[1][1] - r * dR[1][2] + z * dR[2][0] +
                      r * dR[2][1] - 2.f * x * dR[2][2]);
        dq.z = 2.f * (-2.f * y * dR[0][0] + x * dR[0][1] + r * dR[0][2] + x * dR[1][0] + z * dR[1][2] - r * dR[2][0] +
                      z * dR[2][1] - 2.f * y * dR[2][2]);
        dq.w = 2.f * (-2.f * z * dR[0][0] - r * dR[0][1] + x * dR[0][2] + r * dR[1][0] - 2.f * z * dR[1][1] +
                      y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
        if (a.param_space & GSR_RAW_ROTATIONS) {  // q_hat = q / n: dL/dq = (dL/dq_hat - q_hat (q_hat . dL/dq_hat)) / n
            const float dot = fma_(rq.w, dq.w, fma_(rq.z, dq.z, fma_(rq.y, dq.y, rq.x * dq.x)));
            dq = make_float4((dq.x - rq.x * dot) / q_norm, (dq.y - rq.y * dot) / q_norm, (dq.z - rq.z * dot) / q_norm,
                             (dq.w - rq.w * dot) / q_norm);
        }
        *reinterpret_cast<float4 *>(a.dL_drots + 4 * (size_t)i) = dq;
    }
}

// SH basis of degree D at the unit direction (x, y, z) and its derivatives w.r.t. x, y, z
struct ShBasis {
    float v[16], dx[16], dy[16], dz[16];
};
__device__ __forceinline__ void sh_basis(int D, float x, float y, float z, ShBasis &b) {
#pragma unroll
    for (int k = 0; k < 16; k++) b.v[k] = b.dx[k] = b.dy[k] = b.dz[k] = 0.f;
    b.v[0] = kC0;
    if (D > 0) {
        b.v[1] = -(kC1 * y); b.v[2] = kC1 * z; b.v[3] = -(kC1 * x);
        b.dy[1] = -kC1; b.dz[2] = kC1; b.dx[3] = -kC1;
        if (D > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b.v[4] = kC2[0] * xy; b.v[5] = kC2[1] * yz; b.v[6] = kC2[2] * (fma_(2.f, zz, -xx) - yy);
            b.v[7] = kC2[3] * xz; b.v[8] = kC2[4] * (xx - yy);
            b.dx[4] = kC2[0] * y; b.dy[4] = kC2[0] * x;
            b.dy[5] = kC2[1] * z; b.dz[5] = kC2[1] * y;
            b.dx[6] = kC2[2] * -2.f * x; b.dy[6] = kC2[2] * -2.f * y; b.dz[6] = kC2[2] * 4.f * z;
            b.dx[7] = kC2[3] * z; b.dz[7] = kC2[3] * x;
            b.dx[8] = kC2[4] * 2.f * x; b.dy[8] = kC2[4] * -2.f * y;
            if (D > 2) {
                b.v[9] = (kC3[0] * y) * fma_(3.f, xx, -yy);
                b.v[10] = (kC3[1] * xy) * z;
                b.v[11] = (kC3[2] * y) * (fma_(4.f, zz, -xx) - yy);
                b.v[12] = (kC3[3] * z) * fma_(-3.f, yy, fma_(-3.f, xx, 2.f * zz));
                b.v[13] = (kC3[4] * x) * (fma_(4.f, zz, -xx) - yy);
                b.v[14] = (kC3[5] * z) * (xx - yy);
                b.v[15] = (kC3[6] * x) * fma_(-3.f, yy, xx);
                b.dx[9] = kC3[0] * 6.f * x * y; b.dy[9] = kC3[0] * (3.f * xx - 3.f * yy);
                b.dx[10] = kC3[1] * y * z; b.dy[10] = kC3[1] * x * z; b.dz[10] = kC3[1] * x * y;
                b.dx[11] = kC3[2] * -2.f * x * y; b.dy[11] = kC3[2] * (4.f * zz - xx - 3.f * yy);
                b.dz[11] = kC3[2] * 8.f * y * z;
                b.dx[12] = kC3[3] * -6.f * x * z; b.dy[12] = kC3[3] * -6.f * y * z;
                b.dz[12] = kC3[3] * (6.f * zz - 3.f * xx - 3.f * yy);
                b.dx[13] = kC3[4] * (4.f * zz - 3.f * xx - yy); b.dy[13] = kC3[4] * -2.f * x * y;
                b.dz[13] = kC3[4] * 8.f * x * z;
                b.dx[14] = kC3[5] * 2.f * x * z; b.dy[14] = kC3[5] * -2.f * y * z; b.dz[14] = kC3[5] * (xx - yy);
                b.dx[15] = kC3[6] * (3.f * xx - 3.f * yy); b.dy[15] = kC3[6] * -6.f * x * y;
            }
        }
    }
}

// SH colour -> coefficient gradients and the view-direction term of dL/dmean.  CNT = floats per Gaussian of the array
// that goes through LDS: 45 = features_rest of the split storage (the dc triple is read per thread), 48 = packed
// (P,16,3); a workgroup's 256 Gaussians are CNT x 256 CONTIGUOUS floats in, and as many out, so they move as whole
// lines (float4 per lane, consecutive lanes consecutive addresses) through an LDS image with an odd row stride, and
// each thread works on its own row.  The kernel writes EVERY coefficient gradient of its 256 Gaussians -- zeros for
// culled ones and for coefficients above the active degree -- so the caller does not clear those arrays.
// CNT = 0: any other SH layout, per-thread accesses, arrays cleared by the caller.
// (Measured in round 6 and not kept: the forward of the fused training path leaving d colour / d direction -- 9 floats per
// visible Gaussian, computed while the coefficients are in preprocess' registers -- so that this kernel need not read the
// coefficients again: 56.5 -> 32.4 us here, but preprocess 39.7 -> 61.4 us at configs[4] size (the sums alone +9 us, the
// nine stores alone +12 us, planar or as records, 73 or 104 VGPRs): the step stayed at 0.505 ms.)
template <int CNT>
__global__ __launch_bounds__(GSR_BLOCK) void sh_backward_kernel(const BwdArgs a) {
    constexpr int STRIDE = CNT | 1;                                      // odd: a wave's rows start in 32 distinct banks
    constexpr int IT = CNT ? (GSR_BLOCK * CNT + 4 * GSR_BLOCK - 1) / (4 * GSR_BLOCK) : 1;  // float4 per thread
    constexpr int DIV = CNT ? CNT : 1;
    __shared__ __attribute__((aligned(16))) float s_sh[CNT ? GSR_BLOCK * STRIDE + 4 : 4];
    const int tid = (int)threadIdx.x;
    const int base = blockIdx.x * GSR_BLOCK;
    const int i = base + tid;
    const bool vis = i < a.P && a.radii[i] > 0;
    const bool split = CNT == 45 || (CNT == 0 && a.shs_rest != nullptr);
    const float *src = nullptr;
    float *dst = nullptr;
    int n = 0;
    if (CNT) {
        src = (CNT == 45 ? a.shs_rest : a.shs) + (size_t)base * CNT;
        dst = (CNT == 45 ? a.dL_dsh_rest : a.dL_dsh) + (size_t)base * CNT;
        n = min(GSR_BLOCK, a.P - base) * CNT;  // (a multiple of 4 except possibly in the last workgroup)
        const bool any = __syncthreads_or(vis) != 0;
        if (any) {
            // every load of the thread in flight before the first LDS write (a dozen dependent round trips otherwise)
            float4 t[IT];
#pragma unroll
            for (int u = 0; u < IT; u++) {
                const int e = 4 * (u * GSR_BLOCK + tid);
                if (e + 3 < n) {
                    t[u] = *reinterpret_cast<const float4 *>(src + e);
                } else {
                    t[u].x = e < n ? src[e] : 0.f;
                    t[u].y = e + 1 < n ? src[e + 1] : 0.f;
                    t[u].z = e + 2 < n ? src[e + 2] : 0.f;
                    t[u].w = 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < IT; u++) {
                const int e = 4 * (u * GSR_BLOCK + tid);
                if (e >= n) continue;
                if (STRIDE == CNT) {  // the image is the array itself
                    *reinterpret_cast<float4 *>(s_sh + e) = t[u];
                } else {
                    const float v[4] = {t[u].x, t[u].y, t[u].z, t[u].w};
                    const int g = e / DIV, k = e - g * CNT;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int kk = k + q, over = kk >= CNT ? 1 : 0;  // (a float4 straddles at most one row boundary)
                        if (e + q < n) s_sh[(g + over) * STRIDE + kk - over * CNT] = v[q];
                    }
                }
            }
        }
        __syncthreads();
        if (!vis) {
#pragma unroll
            for (int k = 0; k < CNT; k++) s_sh[tid * STRIDE + k] = 0.f;
            if (CNT == 45 && i < a.P) {
                a.dL_dsh[3 * (size_t)i] = 0.f;
                a.dL_dsh[3 * (size_t)i + 1] = 0.f;
                a.dL_dsh[3 * (size_t)i + 2] = 0.f;
            }
        }
    } else if (!vis) {
        return;
    }
    if (vis) {
        const float px = a.means3D[3 * (size_t)i], py = a.means3D[3 * (size_t)i + 1], pz = a.means3D[3 * (size_t)i + 2];
        const float ox = px - a.campos[0], oy = py - a.campos[1], oz = pz - a.campos[2];
        const float len = sqrtf(fma_(oz, oz, fma_(oy, oy, ox * ox)));
        const float x = ox / len, y = oy / len, z = oz / len;
        const uint32_t cl = a.clamped[i];
        // (the three colour sums, as geometry_backward_kernel -- the record's only reader, launched in front of this
        //  kernel -- left them in dL_dcolors)
        const float *rec = a.dL_dcolors + 3 * (size_t)i;
        float dRGB[3];
        dRGB[0] = (cl & 0xffu) ? 0.f : rec[0];
        dRGB[1] = (cl & 0xff00u) ? 0.f : rec[1];
        dRGB[2] = (cl & 0xff0000u) ? 0.f : rec[2];
        ShBasis b;
        sh_basis(a.D, x, y, z, b);
        const int nb = (a.D + 1) * (a.D + 1);
        float ddx = 0.f, ddy = 0.f, ddz = 0.f;
        // coefficient k, channel ch: in at *in, gradient out at *out
        auto term = [&](int k, int ch, const float *in, float *out) {
            const float sv = *in * dRGB[ch];  // (read first: in the LDS image `in` and `out` are the same word)
            *out = b.v[k] * dRGB[ch];
            ddx += b.dx[k] * sv;
            ddy += b.dy[k] * sv;
            ddz += b.dz[k] * sv;
        };
        if (split) {
            const float *dc = a.shs + 3 * (size_t)i;
            float *ddc = a.dL_dsh + 3 * (size_t)i;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) term(0, ch, dc + ch, ddc + ch);
        }
        if (CNT) {
            float *row = s_sh + tid * STRIDE;
            constexpr int K0 = CNT == 45 ? 1 : 0;
#pragma unroll
            for (int k = K0; k < 16; k++) {
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    float *slot = row + 3 * (k - K0) + ch;
                    if (k < nb) term(k, ch, slot, slot);
                    else *slot = 0.f;
                }
            }
        } else if (split) {
            const float *rest = a.shs_rest + (size_t)i * (a.M - 1) * 3;
            float *drest = a.dL_dsh_rest + (size_t)i * (a.M - 1) * 3;
#pragma unroll
            for (int k = 1; k < 16; k++)
                if (k < nb) {
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) term(k, ch, rest + 3 * (k - 1) + ch, drest + 3 * (k - 1) + ch);
                }
        } else {
            const float *sh = a.shs + (size_t)i * a.M * 3;
            float *dsh = a.dL_dsh + (size_t)i * a.M * 3;
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (k < nb) {
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) term(k, ch, sh + 3 * k + ch, dsh + 3 * k + ch);
                }
        }
        const float sum2 = ox * ox + oy * oy + oz * oz;
        const float inv32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
        float *dm = a.dL_dmeans3D + 3 * (size_t)i;
        dm[0] += ((sum2 - ox * ox) * ddx - oy * ox * ddy - oz * ox * ddz) * inv32;
        dm[1] += (-ox * oy * ddx + (sum2 - oy * oy) * ddy - oz * oy * ddz) * inv32;
        dm[2] += (-ox * oz * ddx - oy * oz * ddy + (sum2 - oz * oz) * ddz) * inv32;
    }
    if (CNT) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < IT; u++) {
            const int e = 4 * (u * GSR_BLOCK + tid);
            if (e >= n) continue;
            float v[4];
            if (STRIDE == CNT) {
                const float4 t = *reinterpret_cast<const float4 *>(s_sh + e);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else {
                const int g = e / DIV, k = e - g * CNT;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int kk = k + q, over = kk >= CNT ? 1 : 0;
                    v[q] = e + q < n ? s_sh[(g + over) * STRIDE + kk - over * CNT] : 0.f;
                }
            }
            if (e + 3 < n) {
                *reinterpret_cast<float4 *>(dst + e) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (e + q < n) dst[e + q] = v[q];
            }
        }
    }
}

__global__ void wave_sum_selftest_kernel(const float *in, float *out) {
    // out[w] = sum of in[w*64 .. w*64+63] for each wave of the block
    const float v = in[threadIdx.x];
    const float s = wave_sum(v);
    if (gsr_lane() == 17) out[gsr_wave()] = s;
    // out[4 + w*10 + c] = sum over the wave's lanes of (c + 1) * in[..] * (lane % (c + 2) == 0): ten different
    // per-lane values through wave_reduce10
    float g[kGrad];
#pragma unroll
    for (int c = 0; c < kGrad; c++) g[c] = (gsr_lane() % (c + 2) == 0) ? (float)(c + 1) * v : 0.f;
    const float t = wave_reduce10(g);
    const int comp = wave_reduce10_component();
    if (comp >= 0) out[4 + gsr_wave() * kGrad + comp] = t;
}

}  // namespace

extern "C" int gsr_selftest_wave_sum(const float *in256, float *out44, void *stream) {
    hipLaunchKernelGGL(wave_sum_selftest_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, in256, out44);
    return gsr_check_launch("wave_sum_selftest", true, (hipStream_t)stream);
}

extern "C" int gsr_backward(const GsrSettings *st, const GsrInputs *in, const GsrBackwardInputs *bw,
                            const GsrGrads *gr, void *stream_) {
    if (!st || !in || !bw || !gr) {
        gsr_set_error("gsr_backward: null argument struct");
        return GSR_E_INVALID;
    }
    if (in->shs_rest && (!in->shs || !gr->dL_dsh_rest || st->sh_coeffs < 2)) {
        gsr_set_error("gsr_backward: shs_rest needs shs (the dc part), sh_coeffs >= 2 and GsrGrads.dL_dsh_rest");
        return GSR_E_INVALID;
    }
    if (in->param_space & ~(GSR_RAW_OPACITY | GSR_RAW_SCALES | GSR_RAW_ROTATIONS)) {
        gsr_set_error("gsr_backward: unknown param_space bits");
        return GSR_E_INVALID;
    }
    if (in->part_labels) {
        gsr_set_error("gsr_backward: the in-preprocess part transform (GsrInputs.part_labels) is forward-only");
        return GSR_E_INVALID;
    }
    if ((in->param_space & (GSR_RAW_SCALES | GSR_RAW_ROTATIONS)) && in->cov3D_precomp) {
        gsr_set_error("gsr_backward: raw scales / rotations cannot be combined with cov3D_precomp");
        return GSR_E_INVALID;
    }
    hipStream_t stream = (hipStream_t)stream_;
    const bool debug = st->debug != 0;
    const int P = in->P, W = st->image_width, H = st->image_height;
    const int M = st->sh_coeffs;
    if (P < 0 || W <= 0 || H <= 0) {
        gsr_set_error("gsr_backward: bad sizes");
        return GSR_E_INVALID;
    }
    if (!gr->dL_dmeans2D || !gr->dL_dcolors || !gr->dL_dopacity || !gr->dL_dmeans3D || !gr->dL_dcov3D) {
        gsr_set_error("gsr_backward: gradient buffers missing");
        return GSR_E_INVALID;
    }
    const bool colors_precomp = in->colors_precomp != nullptr;
    const bool cov_precomp = in->cov3D_precomp != nullptr;
    if ((!colors_precomp && !gr->dL_dsh) || (!cov_precomp && (!gr->dL_dscales || !gr->dL_drots))) {
        gsr_set_error("gsr_backward: dL_dsh / dL_dscales / dL_drots required for this input combination");
        return GSR_E_INVALID;
    }
    // Which SH layout goes through sh_backward_kernel's LDS image (it then writes every coefficient gradient itself).
    const bool sh_aligned = ((reinterpret_cast<uintptr_t>(in->shs) | reinterpret_cast<uintptr_t>(gr->dL_dsh) |
                              reinterpret_cast<uintptr_t>(in->shs_rest) | reinterpret_cast<uintptr_t>(gr->dL_dsh_rest)) & 15u) == 0;
    const int sh_cnt = (colors_precomp || M != 16 || !sh_aligned) ? 0 : (in->shs_rest ? 45 : 48);
    const bool work = P > 0 && bw->num_rendered > 0;
    const bool sh_self_clearing = work && sh_cnt != 0;
    // zero the gradient buffers the kernels only write for visible Gaussians; buffers that are adjacent in memory (one
    // arena sliced by the caller, as gsworld_amd/_backward.py does) are cleared by ONE memset instead of ten launches.
    // dL_dconic / dL_dinvdepths are not touched any more (the compositor's sums live in the state's record array).
    {
        const size_t n = (size_t)(P > 0 ? P : 0);
        const size_t m_dc = in->shs_rest ? 1u : (size_t)M, m_rest = in->shs_rest ? (size_t)M - 1u : 0u;
        constexpr int NR = 9;
        // (with work to do, geometry_backward_kernel writes every element of the seven per-Gaussian arrays itself)
        const bool geo = !work;
        struct Range { char *p; size_t bytes; } r[NR] = {
            {(char *)(geo ? gr->dL_dmeans2D : nullptr), 3 * n * 4}, {(char *)(geo ? gr->dL_dcolors : nullptr), 3 * n * 4},
            {(char *)(geo ? gr->dL_dopacity : nullptr), n * 4}, {(char *)(geo ? gr->dL_dmeans3D : nullptr), 3 * n * 4},
            {(char *)(geo ? gr->dL_dcov3D : nullptr), 6 * n * 4},
            {(char *)(sh_self_clearing ? nullptr : gr->dL_dsh), 3 * n * m_dc * 4},
            {(char *)(geo ? gr->dL_dscales : nullptr), 3 * n * 4}, {(char *)(geo ? gr->dL_drots : nullptr), 4 * n * 4},
            {(char *)(in->shs_rest && !sh_self_clearing ? gr->dL_dsh_rest : nullptr), 3 * n * m_rest * 4}};
        for (int i = 1; i < NR; i++)  // insertion sort by address
            for (int j = i; j > 0 && r[j].p < r[j - 1].p; j--) {
                const Range t = r[j];
                r[j] = r[j - 1];
                r[j - 1] = t;
            }
        for (int i = 0; i < NR;) {
            char *p = r[i].p;
            size_t bytes = r[i].bytes;
            int j = i + 1;
            while (j < NR && p && r[j].p == p + bytes && r[j].p) bytes += r[j++].bytes;
            if (p && bytes && hipMemsetAsync(p, 0, bytes, stream) != hipSuccess) {
                gsr_set_error("gsr_backward: hipMemsetAsync failed");
                return GSR_E_HIP;
            }
            i = j;
        }
    }
    if (P == 0 || bw->num_rendered == 0) return GSR_OK;
    if (!bw->geom || !bw->binning || !bw->image || !bw->dL_dout_color || !bw->radii) {
        gsr_set_error("gsr_backward: forward state / dL_dout_color / radii missing");
        return GSR_E_INVALID;
    }
    const GeomState g = GeomState::carve((char *)bw->geom, P, gsr_div_up(W, GSR_TILE) * gsr_div_up(H, GSR_TILE));
    const BinningState b = BinningState::carve((char *)bw->binning, bw->num_rendered);
    const ImageState img = ImageState::carve((char *)bw->image, W, H);
    const int gx = gsr_div_up(W, GSR_TILE), gy = gsr_div_up(H, GSR_TILE);
    if (hipMemsetAsync(g.grad_rec, 0, (size_t)P * kGradRec * sizeof(GsrGradWord), stream) != hipSuccess) {
        gsr_set_error("gsr_backward: hipMemsetAsync failed");
        return GSR_E_HIP;
    }
    hipLaunchKernelGGL(render_backward_kernel, dim3(gx * gy), dim3(GSR_BLOCK), 0, stream, img.ranges, b.gidx[0],
                       g.splat, W, H, gx, in->background, img.final_T, img.n_contrib, bw->dL_dout_color,
                       bw->dL_dout_invdepth, g.grad_rec);
    if (int e = gsr_check_launch("render_backward", debug, stream)) return e;
    BwdArgs a;
    a.P = P; a.D = st->sh_degree; a.M = M; a.W = W; a.H = H;
    a.tanfovx = st->tanfovx; a.tanfovy = st->tanfovy;
    a.fx = (float)W / (2.0f * st->tanfovx); a.fy = (float)H / (2.0f * st->tanfovy);
    a.scale_modifier = st->scale_modifier;
    a.antialiasing = st->antialiasing; a.cov_precomp = cov_precomp; a.colors_precomp = colors_precomp;
    a.have_invdepth = bw->dL_dout_invdepth != nullptr;
    a.param_space = in->param_space; a.shs_rest = in->shs_rest; a.dL_dsh_rest = gr->dL_dsh_rest;
    a.means3D = in->means3D; a.shs = in->shs; a.opacities = in->opacities; a.scales = in->scales;
    a.rotations = in->rotations; a.view = in->viewmatrix; a.proj = in->projmatrix; a.campos = in->campos;
    a.radii = bw->radii;
    a.cov3D = cov_precomp ? in->cov3D_precomp : g.cov3D;
    a.clamped = g.clamped;
    a.grad_rec = g.grad_rec;
    a.dL_dmean2D = gr->dL_dmeans2D; a.dL_dcolors = gr->dL_dcolors;
    a.dL_dopacity = gr->dL_dopacity; a.dL_dmeans3D = gr->dL_dmeans3D; a.dL_dcov3D = gr->dL_dcov3D;
    a.dL_dsh = gr->dL_dsh; a.dL_dscales = gr->dL_dscales; a.dL_drots = gr->dL_drots;
    const dim3 grid(gsr_div_up(P, GSR_BLOCK));
    hipLaunchKernelGGL(geometry_backward_kernel, grid, dim3(GSR_BLOCK), 0, stream, a);
    if (int e = gsr_check_launch("geometry_backward", debug, stream)) return e;
    if (colors_precomp) return GSR_OK;
    if (sh_cnt == 45) hipLaunchKernelGGL(sh_backward_kernel<45>, grid, dim3(GSR_BLOCK), 0, stream, a);
    else if (sh_cnt == 48) hipLaunchKernelGGL(sh_backward_kernel<48>, grid, dim3(GSR_BLOCK), 0, stream, a);
    else hipLaunchKernelGGL(sh_backward_kernel<0>, grid, dim3(GSR_BLOCK), 0, stream, a);
    return gsr_check_launch("sh_backward", debug, stream);
}
