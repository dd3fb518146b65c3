// knn.hip -- simple_knn distCUDA2: mean squared distance to the 3 nearest other points, exact
// (upstream simple-knn/simple_knn.cu).
//
// Same plan as upstream -- Morton order, boxes of 1024 consecutive points with AABBs, prune boxes by a reject
// radius seeded from the +-3 Morton neighbours -- re-cut for wave64: a wave owns 64 consecutive sorted points
// (spatially close), takes the box decision for all of them at once (scan a box if ANY lane needs it; extra
// candidates are harmless because every other point is a valid candidate), and streams the box's points
// through registers 64 at a time with readlane broadcasts, so there is no divergence and no LDS traffic.
// The result is bit-identical to the O(N^2) definition: the squared distance uses one fixed fmaf chain and
// the 3-best insertion is order-independent.
#include <float.h>

#include "gsr_internal.h"

namespace {

constexpr int kBox = 1024;

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

struct KnnState {
    float *minmax_partial;  // [256][6]
    float *minmax;          // [6] min xyz, max xyz
    uint32_t *n_dev;        // [1]
    uint32_t *key[2];       // [P]
    uint32_t *idx[2];       // [P]
    uint32_t *sort_table;   // [256 * nb]
    uint32_t *sort_totals;  // [256]
    float *sorted_pts;      // [P*3] points in Morton order (SoA-free: xyz interleaved)
    float *boxes;           // [nbox][6]
    static KnnState carve(char *base, int32_t P, size_t *bytes = nullptr) {
        KnnState s;
        char *p = base;
        const size_t n = (size_t)(P > 0 ? P : 1);
        s.minmax_partial = GeomState::take<float>(p, 256 * 6);
        s.minmax = GeomState::take<float>(p, 8);
        s.n_dev = GeomState::take<uint32_t>(p, 1);
        s.key[0] = GeomState::take<uint32_t>(p, n);
        s.key[1] = GeomState::take<uint32_t>(p, n);
        s.idx[0] = GeomState::take<uint32_t>(p, n);
        s.idx[1] = GeomState::take<uint32_t>(p, n);
        s.sort_table = GeomState::take<uint32_t>(p, (size_t)GSR_DEPTH_RADIX_BINS * GeomState::sort_blocks(P));
        s.sort_totals = GeomState::take<uint32_t>(p, GSR_DEPTH_RADIX_BINS);
        s.sorted_pts = GeomState::take<float>(p, 3 * n);
        s.boxes = GeomState::take<float>(p, 6 * (size_t)gsr_div_up(n, kBox));
        if (bytes) *bytes = (size_t)(p - base);
        return s;
    }
};

__global__ __launch_bounds__(GSR_BLOCK) void knn_minmax_partial_kernel(int P, const float *__restrict__ pts,
                                                                       float *__restrict__ partial) {
    __shared__ float s_red[4][6];
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * GSR_BLOCK + (int)threadIdx.x; i < P; i += gridDim.x * GSR_BLOCK) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float v = pts[3 * (size_t)i + c];
            mn[c] = fminf(mn[c], v);
            mx[c] = fmaxf(mx[c], v);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], o, 64));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o, 64));
        }
    if (gsr_lane() == 0)
        for (int c = 0; c < 3; c++) {
            s_red[gsr_wave()][c] = mn[c];
            s_red[gsr_wave()][3 + c] = mx[c];
        }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int c = (int)threadIdx.x;
        float v = s_red[0][c];
        for (int w = 1; w < 4; w++) v = c < 3 ? fminf(v, s_red[w][c]) : fmaxf(v, s_red[w][c]);
        partial[blockIdx.x * 6 + c] = v;
    }
}

__global__ void knn_minmax_final_kernel(int nblocks, const float *__restrict__ partial, float *__restrict__ minmax,
                                        uint32_t *n_dev, uint32_t P) {
    const int c = (int)threadIdx.x;
    if (c < 6) {
        float v = partial[c];
        for (int b = 1; b < nblocks; b++) v = c < 3 ? fminf(v, partial[b * 6 + c]) : fmaxf(v, partial[b * 6 + c]);
        minmax[c] = v;
    }
    if (c == 0) *n_dev = P;
}

// 10 bits per axis, x in the lowest interleave position
__device__ __forceinline__ uint32_t spread10(uint32_t x) {
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__global__ __launch_bounds__(GSR_BLOCK) void knn_morton_kernel(int P, const float *__restrict__ pts,
                                                               const float *__restrict__ minmax,
                                                               uint32_t *__restrict__ keys, uint32_t *__restrict__ idx) {
    const int i = blockIdx.x * GSR_BLOCK + (int)threadIdx.x;
    if (i >= P) return;
    uint32_t code = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float lo = minmax[c], hi = minmax[3 + c];
        const float ext = hi - lo;
        float t = ext > 0.f ? (pts[3 * (size_t)i + c] - lo) / ext : 0.f;
        t = fminf(fmaxf(t, 0.f), 1.f);
        code |= spread10((uint32_t)(t * 1023.0f)) << c;
    }
    keys[i] = code;
    idx[i] = (uint32_t)i;
}

__global__ __launch_bounds__(GSR_BLOCK) void knn_gather_box_kernel(int P, const float *__restrict__ pts,
                                                                   const uint32_t *__restrict__ order,
                                                                   float *__restrict__ sorted_pts,
                                                                   float *__restrict__ boxes) {
    // one workgroup per box of 1024 sorted points: gather them and reduce their AABB
    __shared__ float s_red[4][6];
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int k = 0; k < kBox / GSR_BLOCK; k++) {
        const int s = blockIdx.x * kBox + k * GSR_BLOCK + (int)threadIdx.x;
        if (s < P) {
            const uint32_t g = order[s];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float v = pts[3 * (size_t)g + c];
                sorted_pts[3 * (size_t)s + c] = v;
                mn[c] = fminf(mn[c], v);
                mx[c] = fmaxf(mx[c], v);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], o, 64));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o, 64));
        }
    if (gsr_lane() == 0)
        for (int c = 0; c < 3; c++) {
            s_red[gsr_wave()][c] = mn[c];
            s_red[gsr_wave()][3 + c] = mx[c];
        }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int c = (int)threadIdx.x;
        float v = s_red[0][c];
        for (int w = 1; w < 4; w++) v = c < 3 ? fminf(v, s_red[w][c]) : fmaxf(v, s_red[w][c]);
        boxes[blockIdx.x * 6 + c] = v;
    }
}

__device__ __forceinline__ float dist2(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return fma_(dz, dz, fma_(dy, dy, dx * dx));
}

__device__ __forceinline__ void insert3(float &b0, float &b1, float &b2, float d) {
    if (b0 > d) { const float t = b0; b0 = d; d = t; }
    if (b1 > d) { const float t = b1; b1 = d; d = t; }
    if (b2 > d) { b2 = d; }
}

__device__ __forceinline__ float bcast(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__global__ __launch_bounds__(GSR_BLOCK) void knn_search_kernel(int P, const float *__restrict__ sp,
                                                               const uint32_t *__restrict__ order,
                                                               const float *__restrict__ boxes, int nbox,
                                                               float *__restrict__ out) {
    const int s = blockIdx.x * GSR_BLOCK + (int)threadIdx.x;  // sorted position
    const bool valid = s < P;
    const int sc = valid ? s : P - 1;
    const float px = sp[3 * (size_t)sc], py = sp[3 * (size_t)sc + 1], pz = sp[3 * (size_t)sc + 2];
    // reject radius: third-best among the +-3 Morton neighbours
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    for (int o = -3; o <= 3; o++) {
        const int t = sc + o;
        if (o == 0 || t < 0 || t >= P) continue;
        insert3(b0, b1, b2, dist2(px, py, pz, sp[3 * (size_t)t], sp[3 * (size_t)t + 1], sp[3 * (size_t)t + 2]));
    }
    const float reject = b2;
    b0 = b1 = b2 = FLT_MAX;
    const int lane = gsr_lane();
    for (int b = 0; b < nbox; b++) {
        // conservative lower bound with the same fmaf chain as dist2 (each |component| <= the true one)
        const float *bx = boxes + 6 * (size_t)b;
        float dx = 0.f, dy = 0.f, dz = 0.f;
        if (px < bx[0] || px > bx[3]) dx = fminf(fabsf(px - bx[0]), fabsf(px - bx[3]));
        if (py < bx[1] || py > bx[4]) dy = fminf(fabsf(py - bx[1]), fabsf(py - bx[4]));
        if (pz < bx[2] || pz > bx[5]) dz = fminf(fabsf(pz - bx[2]), fabsf(pz - bx[5]));
        const float dbox = fma_(dz, dz, fma_(dy, dy, dx * dx));
        const bool want = valid && !(dbox > reject || dbox > b2);
        if (__ballot(want) == 0ull) continue;  // wave-uniform: nobody needs this box
        const int start = b * kBox;
        const int end = min(P, start + kBox);
        for (int base = start; base < end; base += 64) {
            const int t = base + lane;
            const bool tv = t < end;
            const int tc = tv ? t : end - 1;
            const float qx = sp[3 * (size_t)tc], qy = sp[3 * (size_t)tc + 1], qz = sp[3 * (size_t)tc + 2];
            const int cnt = min(64, end - base);
            for (int j = 0; j < cnt; j++) {
                const float d = dist2(px, py, pz, bcast(qx, j), bcast(qy, j), bcast(qz, j));
                if (base + j != s) insert3(b0, b1, b2, d);
            }
        }
    }
    if (valid) out[order[s]] = (b0 + b1 + b2) / 3.0f;
}

}  // namespace

extern "C" size_t gsr_knn_workspace_bytes(int32_t P) {
    size_t bytes = 0;
    KnnState::carve(nullptr, P, &bytes);
    return bytes;
}

extern "C" int gsr_knn_dist2(int32_t P, const float *points, float *mean_dist2, void *workspace,
                             size_t workspace_bytes, void *stream_) {
    if (P < 0 || (P > 0 && (!points || !mean_dist2 || !workspace))) {
        gsr_set_error("gsr_knn_dist2: null pointer or negative P");
        return GSR_E_INVALID;
    }
    if (P == 0) return GSR_OK;
    if (workspace_bytes < gsr_knn_workspace_bytes(P)) {
        gsr_set_error("gsr_knn_dist2: workspace too small");
        return GSR_E_INVALID;
    }
    hipStream_t stream = (hipStream_t)stream_;
    const KnnState s = KnnState::carve((char *)workspace, P);
    const int nred = min(256, gsr_div_up(P, GSR_BLOCK));
    hipLaunchKernelGGL(knn_minmax_partial_kernel, dim3(nred), dim3(GSR_BLOCK), 0, stream, P, points, s.minmax_partial);
    hipLaunchKernelGGL(knn_minmax_final_kernel, dim3(1), dim3(64), 0, stream, nred, s.minmax_partial, s.minmax, s.n_dev,
                       (uint32_t)P);
    // 30-bit codes, 11 bits per pass = 3 passes: start in side 1 so that the sorted order ends in idx[0]
    const int start = gsr_radix_passes(30, GSR_DEPTH_RADIX_BITS) & 1;
    hipLaunchKernelGGL(knn_morton_kernel, dim3(gsr_div_up(P, GSR_BLOCK)), dim3(GSR_BLOCK), 0, stream, P, points,
                       s.minmax, s.key[start], s.idx[start]);
    if (int e = gsr_check_launch("knn_morton", false, stream)) return e;
    uint32_t *key[2] = {s.key[0], s.key[1]};
    uint32_t *val[2] = {s.idx[0], s.idx[1]};
    if (int e = gsr_radix_sort_u32(key, val, s.n_dev, P, 30, GSR_DEPTH_RADIX_BITS, start, s.sort_table, s.sort_totals,
                                   false, stream))
        return e;
    const int nbox = gsr_div_up(P, kBox);
    hipLaunchKernelGGL(knn_gather_box_kernel, dim3(nbox), dim3(GSR_BLOCK), 0, stream, P, points, s.idx[0],
                       s.sorted_pts, s.boxes);
    hipLaunchKernelGGL(knn_search_kernel, dim3(gsr_div_up(P, GSR_BLOCK)), dim3(GSR_BLOCK), 0, stream, P, s.sorted_pts,
                       s.idx[0], s.boxes, nbox, mean_dist2);
    return gsr_check_launch("knn_search", false, stream);
}
