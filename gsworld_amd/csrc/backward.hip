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
    const float *__restrict__ dL_dinvdepth_pix, float *__restrict__ dL_dmean2D /*(P,3)*/,
    float *__restrict__ dL_dconic /*(P,4): xx, xy, -, yy*/, float *__restrict__ dL_dopacity,
    float *__restrict__ dL_dcolors /*(P,3)*/, float *__restrict__ dL_dinvdepths /*(P)*/) {
    __shared__ float4 s_rec0[GSR_BLOCK];
    __shared__ float4 s_rec1[GSR_BLOCK];
    __shared__ float4 s_rec2[GSR_BLOCK];
    __shared__ uint32_t s_id[GSR_BLOCK];
    __shared__ float s_grad[GSR_BLOCK * kGrad];
    __shared__ uint32_t s_max[4];

    const int tile = (int)blockIdx.x;
    const int tile_x = tile % gx, tile_y = tile / gx;
    const int lane = gsr_lane(), wave = gsr_wave();
    const int lx = ((wave & 1) << 3) | (lane & 7);
    const int ly = ((wave >> 1) << 3) | (lane >> 3);
    const int px = tile_x * GSR_TILE + lx, py = tile_y * GSR_TILE + ly;
    const bool inside = px < W && py < H;
    const float pfx = (float)px, pfy = (float)py;
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
        for (int c = 0; c < kGrad; c++) s_grad[c * GSR_BLOCK + threadIdx.x] = 0.f;
        __syncthreads();
        const int cnt = min(GSR_BLOCK, n_inst - rd * GSR_BLOCK);
        for (int j = 0; j < cnt; j++) {
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
                    const float G = __builtin_amdgcn_exp2f(power * 1.4426950408889634f);
                    const float alpha = fminf(0.99f, r1.w * G);
                    if (alpha >= 1.0f / 255.0f) {
                        hit = true;
                        const float4 r2 = s_rec2[j];
                        T = T / (1.f - alpha);
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
                        dL_dalpha = fma_(-T_final / (1.f - alpha), bg_dot, dL_dalpha);
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
            const float total = wave_reduce10(gv);
            if (my_comp >= 0) atomicAdd(&s_grad[my_comp * GSR_BLOCK + j], total);  // LDS: at most 4 waves per address
        }
        __syncthreads();
        // flush: thread j owns staged instance j
        // (measured: without this flush the kernel takes 137 instead of 214 us at config 5 -- a third of it is the
        // device-scope float atomics)
        if ((int)threadIdx.x < cnt) {
            const uint32_t g = s_id[threadIdx.x];
            const float v0 = s_grad[0 * GSR_BLOCK + threadIdx.x], v1 = s_grad[1 * GSR_BLOCK + threadIdx.x];
            const float v2 = s_grad[2 * GSR_BLOCK + threadIdx.x], v3 = s_grad[3 * GSR_BLOCK + threadIdx.x];
            const float v4 = s_grad[4 * GSR_BLOCK + threadIdx.x], v5 = s_grad[5 * GSR_BLOCK + threadIdx.x];
            const float v6 = s_grad[6 * GSR_BLOCK + threadIdx.x], v7 = s_grad[7 * GSR_BLOCK + threadIdx.x];
            const float v8 = s_grad[8 * GSR_BLOCK + threadIdx.x], v9 = s_grad[9 * GSR_BLOCK + threadIdx.x];
            if (v0 != 0.f) atomicAdd(&dL_dmean2D[3 * (size_t)g], v0);
            if (v1 != 0.f) atomicAdd(&dL_dmean2D[3 * (size_t)g + 1], v1);
            if (v2 != 0.f) atomicAdd(&dL_dconic[4 * (size_t)g], v2);
            if (v3 != 0.f) atomicAdd(&dL_dconic[4 * (size_t)g + 1], v3);
            if (v4 != 0.f) atomicAdd(&dL_dconic[4 * (size_t)g + 3], v4);
            if (v5 != 0.f) atomicAdd(&dL_dopacity[g], v5);
            if (v6 != 0.f) atomicAdd(&dL_dcolors[3 * (size_t)g], v6);
            if (v7 != 0.f) atomicAdd(&dL_dcolors[3 * (size_t)g + 1], v7);
            if (v8 != 0.f) atomicAdd(&dL_dcolors[3 * (size_t)g + 2], v8);
            if (v9 != 0.f) atomicAdd(&dL_dinvdepths[g], v9);
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
    const float *dL_dmean2D, *dL_dconic, *dL_dcolors, *dL_dinvdepths;
    float *dL_dopacity, *dL_dmeans3D, *dL_dcov3D, *dL_dsh, *dL_dscales, *dL_drots;
    float *dL_dsh_rest;        // with shs_rest: dL_dsh is the dc part
};

// per-Gaussian chain rule; one thread per Gaussian
__global__ __launch_bounds__(GSR_BLOCK) void preprocess_backward_kernel(const BwdArgs a) {
    const int i = blockIdx.x * GSR_BLOCK + (int)threadIdx.x;
    if (i >= a.P || a.radii[i] <= 0) return;
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
    if (raw_opacity && !a.antialiasing) a.dL_dopacity[i] *= opacity_act * (1.0f - opacity_act);
    if (a.antialiasing) {
        const float det_cov = fma_(-cb, cb, ca * cc);
        ca += h_var;
        cc += h_var;
        const float det_plus = fma_(-cb, cb, ca * cc);
        const float ratio = det_cov / det_plus;
        const float h_scale = sqrtf(fmaxf(0.000025f, ratio));
        const float dL_dop = a.dL_dopacity[i];
        const float d_h = dL_dop * opacity_act;
        a.dL_dopacity[i] = raw_opacity ? (dL_dop * h_scale) * (opacity_act * (1.0f - opacity_act)) : dL_dop * h_scale;
        const float d_root = ratio <= 0.000025f ? 0.f : d_h / (2.f * h_scale);
        const float inv2 = 1.f / (det_plus * det_plus);
        dL_da_aa = d_root * ((cc - h_var) * det_plus - det_cov * cc) * inv2;
        dL_dc_aa = d_root * ((ca - h_var) * det_plus - det_cov * ca) * inv2;
        dL_db_aa = d_root * (-2.f * cb * det_plus + 2.f * cb * det_cov) * inv2;
    } else {
        ca += h_var;
        cc += h_var;
    }
    const float Lx = a.dL_dconic[4 * (size_t)i], Ly = a.dL_dconic[4 * (size_t)i + 1], Lz = a.dL_dconic[4 * (size_t)i + 3];
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
    if (a.have_invdepth) dtz -= a.dL_dinvdepths[i] / (tz * tz);
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
    const float g2x = a.dL_dmean2D[3 * (size_t)i], g2y = a.dL_dmean2D[3 * (size_t)i + 1];
    dmean[0] += (q[0] * m_w - q[3] * mul1) * g2x + (q[1] * m_w - q[3] * mul2) * g2y;
    dmean[1] += (q[4] * m_w - q[7] * mul1) * g2x + (q[5] * m_w - q[7] * mul2) * g2y;
    dmean[2] += (q[8] * m_w - q[11] * mul1) * g2x + (q[9] * m_w - q[11] * mul2) * g2y;
    // SH -> colour
    if (!a.colors_precomp) {
        const float ox = px - a.campos[0], oy = py - a.campos[1], oz = pz - a.campos[2];
        const float len = sqrtf(fma_(oz, oz, fma_(oy, oy, ox * ox)));
        const float x = ox / len, y = oy / len, z = oz / len;
        const uint32_t cl = a.clamped[i];
        float dRGB[3];
        dRGB[0] = (cl & 0xffu) ? 0.f : a.dL_dcolors[3 * (size_t)i];
        dRGB[1] = (cl & 0xff00u) ? 0.f : a.dL_dcolors[3 * (size_t)i + 1];
        dRGB[2] = (cl & 0xff0000u) ? 0.f : a.dL_dcolors[3 * (size_t)i + 2];
        float bas[16], bx[16], by[16], bz[16];
#pragma unroll
        for (int k = 0; k < 16; k++) bas[k] = bx[k] = by[k] = bz[k] = 0.f;
        const int D = a.D;
        bas[0] = kC0;
        if (D > 0) {
            bas[1] = -(kC1 * y); bas[2] = kC1 * z; bas[3] = -(kC1 * x);
            by[1] = -kC1; bz[2] = kC1; bx[3] = -kC1;
            if (D > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                bas[4] = kC2[0] * xy; bas[5] = kC2[1] * yz; bas[6] = kC2[2] * (fma_(2.f, zz, -xx) - yy);
                bas[7] = kC2[3] * xz; bas[8] = kC2[4] * (xx - yy);
                bx[4] = kC2[0] * y; by[4] = kC2[0] * x;
                by[5] = kC2[1] * z; bz[5] = kC2[1] * y;
                bx[6] = kC2[2] * -2.f * x; by[6] = kC2[2] * -2.f * y; bz[6] = kC2[2] * 4.f * z;
                bx[7] = kC2[3] * z; bz[7] = kC2[3] * x;
                bx[8] = kC2[4] * 2.f * x; by[8] = kC2[4] * -2.f * y;
                if (D > 2) {
                    bas[9] = (kC3[0] * y) * fma_(3.f, xx, -yy);
                    bas[10] = (kC3[1] * xy) * z;
                    bas[11] = (kC3[2] * y) * (fma_(4.f, zz, -xx) - yy);
                    bas[12] = (kC3[3] * z) * fma_(-3.f, yy, fma_(-3.f, xx, 2.f * zz));
                    bas[13] = (kC3[4] * x) * (fma_(4.f, zz, -xx) - yy);
                    bas[14] = (kC3[5] * z) * (xx - yy);
                    bas[15] = (kC3[6] * x) * fma_(-3.f, yy, xx);
                    bx[9] = kC3[0] * 6.f * x * y; by[9] = kC3[0] * (3.f * xx - 3.f * yy);
                    bx[10] = kC3[1] * y * z; by[10] = kC3[1] * x * z; bz[10] = kC3[1] * x * y;
                    bx[11] = kC3[2] * -2.f * x * y; by[11] = kC3[2] * (4.f * zz - xx - 3.f * yy);
                    bz[11] = kC3[2] * 8.f * y * z;
                    bx[12] = kC3[3] * -6.f * x * z; by[12] = kC3[3] * -6.f * y * z;
                    bz[12] = kC3[3] * (6.f * zz - 3.f * xx - 3.f * yy);
                    bx[13] = kC3[4] * (4.f * zz - 3.f * xx - yy); by[13] = kC3[4] * -2.f * x * y;
                    bz[13] = kC3[4] * 8.f * x * z;
                    bx[14] = kC3[5] * 2.f * x * z; by[14] = kC3[5] * -2.f * y * z; bz[14] = kC3[5] * (xx - yy);
                    bx[15] = kC3[6] * (3.f * xx - 3.f * yy); by[15] = kC3[6] * -6.f * x * y;
                }
            }
        }
        const int nb = (D + 1) * (D + 1);
        const float *sh = a.shs + (size_t)i * a.M * 3;
        float *dsh = a.dL_dsh + (size_t)i * a.M * 3;
        float ddx = 0.f, ddy = 0.f, ddz = 0.f;
        if (a.shs_rest) {
            // split storage (features_dc | features_rest): coefficient 0 in one pair of arrays, 1.. in the other
            const float *dc = a.shs + 3 * (size_t)i;
            float *ddc = a.dL_dsh + 3 * (size_t)i;
            const float *rest = a.shs_rest + (size_t)i * (a.M - 1) * 3;
            float *drest = a.dL_dsh_rest + (size_t)i * (a.M - 1) * 3;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                ddc[ch] = bas[0] * dRGB[ch];
                const float sv = dc[ch] * dRGB[ch];
                ddx += bx[0] * sv;
                ddy += by[0] * sv;
                ddz += bz[0] * sv;
            }
            if (D == 3 && a.M == 16) {
                // 45 floats in, 45 out, every load ahead of every store: 15 + 15 dwordx3 accesses
                const float3 *rest3 = reinterpret_cast<const float3 *>(rest);
                float3 *drest3 = reinterpret_cast<float3 *>(drest);
                float3 f[15], o[15];
#pragma unroll
                for (int k = 0; k < 15; k++) f[k] = rest3[k];
#pragma unroll
                for (int k = 1; k < 16; k++) {
                    o[k - 1] = make_float3(bas[k] * dRGB[0], bas[k] * dRGB[1], bas[k] * dRGB[2]);
                    const float s0 = f[k - 1].x * dRGB[0], s1 = f[k - 1].y * dRGB[1], s2 = f[k - 1].z * dRGB[2];
                    ddx += bx[k] * s0; ddy += by[k] * s0; ddz += bz[k] * s0;
                    ddx += bx[k] * s1; ddy += by[k] * s1; ddz += bz[k] * s1;
                    ddx += bx[k] * s2; ddy += by[k] * s2; ddz += bz[k] * s2;
                }
#pragma unroll
                for (int k = 0; k < 15; k++) drest3[k] = o[k];
            } else {
#pragma unroll
                for (int k = 1; k < 16; k++) {
                    if (k < nb) {
#pragma unroll
                        for (int ch = 0; ch < 3; ch++) {
                            drest[3 * (k - 1) + ch] = bas[k] * dRGB[ch];
                            const float sv = rest[3 * (k - 1) + ch] * dRGB[ch];
                            ddx += bx[k] * sv;
                            ddy += by[k] * sv;
                            ddz += bz[k] * sv;
                        }
                    }
                }
            }
        } else if (D == 3 && a.M == 16 && ((reinterpret_cast<uintptr_t>(a.shs) | reinterpret_cast<uintptr_t>(a.dL_dsh)) & 15u) == 0) {
            // 48 contiguous floats in, 48 out: 12 + 12 dwordx4 accesses instead of 96 dword ones
            float4 v[12], o[12];
            const float4 *sh4 = reinterpret_cast<const float4 *>(sh);
#pragma unroll
            for (int q = 0; q < 12; q++) v[q] = sh4[q];
            const float *f = reinterpret_cast<const float *>(v);
            float *of = reinterpret_cast<float *>(o);
#pragma unroll
            for (int k = 0; k < 16; k++) {
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    of[3 * k + ch] = bas[k] * dRGB[ch];
                    const float sv = f[3 * k + ch] * dRGB[ch];
                    ddx += bx[k] * sv;
                    ddy += by[k] * sv;
                    ddz += bz[k] * sv;
                }
            }
            float4 *dsh4 = reinterpret_cast<float4 *>(dsh);
#pragma unroll
            for (int q = 0; q < 12; q++) dsh4[q] = o[q];
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (k < nb) {
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        dsh[3 * k + ch] = bas[k] * dRGB[ch];
                        const float sv = sh[3 * k + ch] * dRGB[ch];
                        ddx += bx[k] * sv;
                        ddy += by[k] * sv;
                        ddz += bz[k] * sv;
                    }
                }
            }
        }
        const float sum2 = ox * ox + oy * oy + oz * oz;
        const float inv32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
        dmean[0] += ((sum2 - ox * ox) * ddx - oy * ox * ddy - oz * ox * ddz) * inv32;
        dmean[1] += (-ox * oy * ddx + (sum2 - oy * oy) * ddy - oz * oy * ddz) * inv32;
        dmean[2] += (-ox * oz * ddx - oy * oz * ddy + (sum2 - oz * oz) * ddz) * inv32;
    }
    a.dL_dmeans3D[3 * (size_t)i] = dmean[0];
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
    if (!gr->dL_dmeans2D || !gr->dL_dcolors || !gr->dL_dopacity || !gr->dL_dmeans3D || !gr->dL_dcov3D ||
        !gr->dL_dconic || !gr->dL_dinvdepths) {
        gsr_set_error("gsr_backward: gradient buffers missing");
        return GSR_E_INVALID;
    }
    const bool colors_precomp = in->colors_precomp != nullptr;
    const bool cov_precomp = in->cov3D_precomp != nullptr;
    if ((!colors_precomp && !gr->dL_dsh) || (!cov_precomp && (!gr->dL_dscales || !gr->dL_drots))) {
        gsr_set_error("gsr_backward: dL_dsh / dL_dscales / dL_drots required for this input combination");
        return GSR_E_INVALID;
    }
    // zero every gradient buffer; buffers that are adjacent in memory (one arena sliced by the caller, as
    // gsworld_amd/_backward.py does) are cleared by ONE memset instead of ten launches
    {
        const size_t n = (size_t)(P > 0 ? P : 0);
        const size_t m_dc = in->shs_rest ? 1u : (size_t)M, m_rest = in->shs_rest ? (size_t)M - 1u : 0u;
        constexpr int NR = 11;
        struct Range { char *p; size_t bytes; } r[NR] = {
            {(char *)gr->dL_dmeans2D, 3 * n * 4}, {(char *)gr->dL_dcolors, 3 * n * 4}, {(char *)gr->dL_dopacity, n * 4},
            {(char *)gr->dL_dmeans3D, 3 * n * 4}, {(char *)gr->dL_dcov3D, 6 * n * 4},
            {(char *)gr->dL_dsh, 3 * n * m_dc * 4}, {(char *)gr->dL_dscales, 3 * n * 4},
            {(char *)gr->dL_drots, 4 * n * 4}, {(char *)gr->dL_dconic, 4 * n * 4}, {(char *)gr->dL_dinvdepths, n * 4},
            {(char *)(in->shs_rest ? gr->dL_dsh_rest : nullptr), 3 * n * m_rest * 4}};
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
            while (j < NR && p && r[j].p == p + bytes) bytes += r[j++].bytes;
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
    hipLaunchKernelGGL(render_backward_kernel, dim3(gx * gy), dim3(GSR_BLOCK), 0, stream, img.ranges, b.gidx[0],
                       g.splat, W, H, gx, in->background, img.final_T, img.n_contrib, bw->dL_dout_color,
                       bw->dL_dout_invdepth, gr->dL_dmeans2D, gr->dL_dconic, gr->dL_dopacity, gr->dL_dcolors,
                       gr->dL_dinvdepths);
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
    a.dL_dmean2D = gr->dL_dmeans2D; a.dL_dconic = gr->dL_dconic; a.dL_dcolors = gr->dL_dcolors;
    a.dL_dinvdepths = gr->dL_dinvdepths;
    a.dL_dopacity = gr->dL_dopacity; a.dL_dmeans3D = gr->dL_dmeans3D; a.dL_dcov3D = gr->dL_dcov3D;
    a.dL_dsh = gr->dL_dsh; a.dL_dscales = gr->dL_dscales; a.dL_drots = gr->dL_drots;
    hipLaunchKernelGGL(preprocess_backward_kernel, dim3(gsr_div_up(P, GSR_BLOCK)), dim3(GSR_BLOCK), 0, stream, a);
    return gsr_check_launch("preprocess_backward", debug, stream);
}
