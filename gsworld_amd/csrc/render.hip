// render.hip -- front-to-back alpha compositing, one 16x16-pixel tile per 256-thread workgroup
// (upstream forward.cu renderCUDA).
//
// CDNA4 mapping: the 4 waves of a workgroup own the four 8x8 quadrants of the tile (lane -> (x&7, y>>3)), so
// that a wave's 64 pixels are spatially compact and its alpha-test / early-exit decisions are coherent.
// Each round stages 256 instances (3 x 16 B each: xy|depth|1/depth, conic|opacity, rgb|radius) into LDS with one
// gather per thread; the inner loop reads them back as wave-uniform ds_read_b128 broadcasts.
//
// Arithmetic contract (matches oracle/gs_oracle.c gso_render): -ffp-contract=off, explicit fmaf; the only
// non-bit-reproducible operation is exp(): exp2(power * log2e) on the hardware transcendental unit.
#include "gsr_internal.h"

namespace {

int g_render_variant = 2;
int g_render_blocks_per_cu = 5;

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

__global__ __launch_bounds__(GSR_BLOCK) void render_kernel(const uint2 *__restrict__ ranges,
                                                           const uint32_t *__restrict__ point_list,
                                                           const float4 *__restrict__ splat, int W, int H, int gx,
                                                           const float *__restrict__ bg,
                                                           float *__restrict__ out_color,
                                                           float *__restrict__ out_invdepth,
                                                           float *__restrict__ final_T,
                                                           uint32_t *__restrict__ n_contrib) {
    __shared__ float4 s_rec0[GSR_BLOCK];
    __shared__ float4 s_rec1[GSR_BLOCK];
    __shared__ float4 s_rec2[GSR_BLOCK];

    const int tile = (int)blockIdx.x;
    const int tile_x = tile % gx, tile_y = tile / gx;
    const int lane = gsr_lane(), wave = gsr_wave();
    const int lx = ((wave & 1) << 3) | (lane & 7);
    const int ly = ((wave >> 1) << 3) | (lane >> 3);
    const int px = tile_x * GSR_TILE + lx, py = tile_y * GSR_TILE + ly;
    const bool inside = px < W && py < H;
    const float pfx = (float)px, pfy = (float)py;

    const uint2 range = ranges[tile];
    const int n_inst = (int)(range.y - range.x);
    const int rounds = (n_inst + GSR_BLOCK - 1) / GSR_BLOCK;

    bool done = !inside;
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dacc = 0.f;
    uint32_t contributor = 0, last_contributor = 0;

    int todo = n_inst;
    for (int rd = 0; rd < rounds; rd++, todo -= GSR_BLOCK) {
        // whole tile saturated? (wave-uniform value from the block-wide vote)
        if (__syncthreads_count(done ? 1 : 0) == GSR_BLOCK) break;
        const int fetch = rd * GSR_BLOCK + (int)threadIdx.x;
        if (fetch < n_inst) {
            const uint32_t g = point_list[range.x + (uint32_t)fetch];
            const float4 *rec = splat + 3 * (size_t)g;
            s_rec0[threadIdx.x] = rec[0];
            s_rec1[threadIdx.x] = rec[1];
            s_rec2[threadIdx.x] = rec[2];
        }
        __syncthreads();
        const int cnt = todo < GSR_BLOCK ? todo : GSR_BLOCK;
        for (int j = 0; j < cnt; j++) {
            if (__ballot(!done) == 0ull) break;  // this wave's 64 pixels are all finished
            if (done) continue;
            contributor++;
            const float4 r0 = s_rec0[j];
            const float4 r1 = s_rec1[j];
            const float dx = r0.x - pfx, dy = r0.y - pfy;
            const float q = fma_(r1.z * dy, dy, (r1.x * dx) * dx);
            const float power = fma_(-(r1.y * dx), dy, -0.5f * q);
            if (power > 0.0f) continue;
            const float alpha = fminf(0.99f, r1.w * __builtin_amdgcn_exp2f(power * 1.4426950408889634f));
            if (alpha < 1.0f / 255.0f) continue;
            const float test_T = T * (1.0f - alpha);
            if (test_T < 0.0001f) {
                done = true;
                continue;
            }
            const float4 r2 = s_rec2[j];
            const float w = alpha * T;
            C0 = fma_(r2.x, w, C0);
            C1 = fma_(r2.y, w, C1);
            C2 = fma_(r2.z, w, C2);
            Dacc = fma_(r0.w, w, Dacc);
            T = test_T;
            last_contributor = contributor;
        }
    }
    if (inside) {
        const size_t pid = (size_t)py * W + px;
        const size_t plane = (size_t)H * W;
        final_T[pid] = T;
        n_contrib[pid] = last_contributor;
        out_color[pid] = fma_(T, bg[0], C0);
        out_color[plane + pid] = fma_(T, bg[1], C1);
        out_color[2 * plane + pid] = fma_(T, bg[2], C2);
        out_invdepth[pid] = Dacc;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Variant 1 (default): wave-independent compositing, no LDS, no barriers.
//
// Each of the 4 waves of a tile walks the tile's instance list on its own.  A chunk of 64 instances is fetched
// with ONE instance per lane (index -> 3 x dwordx4 record gather, software-pipelined one chunk ahead) and then
// replayed j = 0..63 by pulling instance j out of lane j with v_readlane: the instance becomes wave-uniform SGPR
// operands of the per-pixel VALU work.  The inner loop is pure VALU/SALU -- no ds_read latency, no LDS bandwidth
// shared between the CU's 4 SIMDs, no workgroup barrier -- and a wave retires as soon as ITS 64 pixels are
// saturated.  The 4x redundant record gathers are L1/L2 hits; the tile -> workgroup map keeps each XCD on a
// contiguous band of tiles so that its private L2 holds that band's splat records.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float rl(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__global__ __launch_bounds__(GSR_BLOCK) void render_wave_kernel(const uint2 *__restrict__ ranges,
                                                                const uint32_t *__restrict__ point_list,
                                                                const float4 *__restrict__ splat, int W, int H, int gx,
                                                                int num_tiles, const float *__restrict__ bg,
                                                                float *__restrict__ out_color,
                                                                float *__restrict__ out_invdepth,
                                                                float *__restrict__ final_T,
                                                                uint32_t *__restrict__ n_contrib) {
    // XCD-aware, bijective remap: workgroup b runs on XCD b % 8 (observed dispatch order, speed only);
    // give XCD x the contiguous tile band [start(x), start(x) + len(x)).
    int tile;
    {
        const int b = (int)blockIdx.x, xcd = b & 7, q = num_tiles >> 3, r = num_tiles & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int tile_x = tile % gx, tile_y = tile / gx;
    const int lane = gsr_lane(), wave = gsr_wave();
    const int lx = ((wave & 1) << 3) | (lane & 7);
    const int ly = ((wave >> 1) << 3) | (lane >> 3);
    const int px = tile_x * GSR_TILE + lx, py = tile_y * GSR_TILE + ly;
    const bool inside = px < W && py < H;
    const float pfx = (float)px, pfy = (float)py;

    const uint2 range = ranges[tile];
    const int n_inst = (int)(range.y - range.x);
    const uint32_t *list = point_list + range.x;

    bool done = !inside;
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dacc = 0.f;
    uint32_t contributor = 0, last_contributor = 0;

    // software pipeline: records of chunk c are in (a0,a1,a2); the index of chunk c+1's instance is in g_next
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
    uint32_t g_next = 0;
    if (lane < n_inst) {
        const uint32_t g = list[lane];
        const float4 *rec = splat + 3 * (size_t)g;
        a0 = rec[0]; a1 = rec[1]; a2 = rec[2];
    }
    if (64 + lane < n_inst) g_next = list[64 + lane];

    for (int base = 0; base < n_inst; base += 64) {
        if (__ballot(!done) == 0ull) break;  // this wave's 64 pixels are saturated
        // issue the next chunk's gathers before touching the current chunk
        float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0, b2 = b0;
        uint32_t g_next2 = 0;
        if (base + 64 + lane < n_inst) {
            const float4 *rec = splat + 3 * (size_t)g_next;
            b0 = rec[0]; b1 = rec[1]; b2 = rec[2];
        }
        if (base + 128 + lane < n_inst) g_next2 = list[base + 128 + lane];

        const int cnt = min(64, n_inst - base);
#pragma unroll 2
        for (int j = 0; j < cnt; j++) {
            const float sx = rl(a0.x, j), sy = rl(a0.y, j);
            const float cxx = rl(a1.x, j), cxy = rl(a1.y, j), cyy = rl(a1.z, j), op = rl(a1.w, j);
            const float dx = sx - pfx, dy = sy - pfy;
            const float q = fma_(cyy * dy, dy, (cxx * dx) * dx);
            const float power = fma_(-(cxy * dx), dy, -0.5f * q);
            const float alpha = fminf(0.99f, op * __builtin_amdgcn_exp2f(power * 1.4426950408889634f));
            const bool live = !done;
            const bool hit = live && power <= 0.0f && alpha >= 1.0f / 255.0f;
            contributor += live ? 1u : 0u;
            if (__ballot(hit) == 0ull) {
                if (__ballot(live) == 0ull) break;
                continue;
            }
            // wave-uniform here: broadcast colour and 1/depth before the lanes diverge
            const float cr = rl(a2.x, j), cg = rl(a2.y, j), cb = rl(a2.z, j), invd = rl(a0.w, j);
            if (hit) {
                const float test_T = T * (1.0f - alpha);
                if (test_T < 0.0001f) {
                    done = true;
                } else {
                    const float w = alpha * T;
                    C0 = fma_(cr, w, C0);
                    C1 = fma_(cg, w, C1);
                    C2 = fma_(cb, w, C2);
                    Dacc = fma_(invd, w, Dacc);
                    T = test_T;
                    last_contributor = contributor;
                }
            }
        }
        a0 = b0; a1 = b1; a2 = b2;
        g_next = g_next2;
    }
    if (inside) {
        const size_t pid = (size_t)py * W + px;
        const size_t plane = (size_t)H * W;
        final_T[pid] = T;
        n_contrib[pid] = last_contributor;
        out_color[pid] = fma_(T, bg[0], C0);
        out_color[plane + pid] = fma_(T, bg[1], C1);
        out_color[2 * plane + pid] = fma_(T, bg[2], C2);
        out_invdepth[pid] = Dacc;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Variant 2: LDS-staged, 4 instances per step, predicated, persistent workgroups on a longest-first tile queue.
//
// What the counters of variant 0 showed (profiles/round1): half of the wave time parked on lgkmcnt / barriers,
// as many SALU (exec-mask) instructions as VALU, and an average wave lifetime of half the kernel (tile lists
// differ 3x in length while every tile is resident from t = 0).  Hence:
//  * the per-pixel test (power, alpha, thresholds) of FOUR instances is evaluated back to back on operands that
//    were all fetched from LDS by one batch of ds_reads -> one lgkmcnt wait per 4 instances and 4-way ILP for a
//    wave that is alone on its SIMD at the tail;
//  * no divergent control flow: lane state is updated with selects, branches are wave-uniform (ballot);
//    n_contrib needs no per-lane counter because a live lane has examined exactly (position + 1) instances;
//  * workgroups are persistent and pull tiles from a queue ordered by descending list length, so the chip
//    drains evenly instead of waiting for the CU that happened to receive the long tiles.
// ---------------------------------------------------------------------------------------------------------
constexpr int kBatch = 4;

__global__ __launch_bounds__(GSR_BLOCK) void render_queue_kernel(const uint2 *__restrict__ ranges,
                                                                 const uint32_t *__restrict__ point_list,
                                                                 const float4 *__restrict__ splat, int W, int H, int gx,
                                                                 int num_tiles, const uint32_t *__restrict__ tile_order,
                                                                 uint32_t *__restrict__ queue_head,
                                                                 const float *__restrict__ bg,
                                                                 float *__restrict__ out_color,
                                                                 float *__restrict__ out_invdepth,
                                                                 float *__restrict__ final_T,
                                                                 uint32_t *__restrict__ n_contrib) {
    __shared__ float4 s_rec0[GSR_BLOCK + kBatch];
    __shared__ float4 s_rec1[GSR_BLOCK + kBatch];
    __shared__ float4 s_rec2[GSR_BLOCK + kBatch];
    __shared__ uint32_t s_ticket;
    const int lane = gsr_lane(), wave = gsr_wave();
    const int lx = ((wave & 1) << 3) | (lane & 7);
    const int ly = ((wave >> 1) << 3) | (lane >> 3);
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    if (threadIdx.x < kBatch) {  // padding entries: alpha = 0 -> never valid
        s_rec0[GSR_BLOCK + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
        s_rec1[GSR_BLOCK + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
        s_rec2[GSR_BLOCK + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (;;) {
        __syncthreads();  // previous tile fully consumed (LDS records and the ticket)
        if (threadIdx.x == 0) s_ticket = atomicAdd(queue_head, 1u);
        __syncthreads();
        const uint32_t ticket = s_ticket;
        if (ticket >= (uint32_t)num_tiles) break;
        const int tile = tile_order ? (int)tile_order[ticket] : (int)ticket;
        const int tile_x = tile % gx, tile_y = tile / gx;
        const int px = tile_x * GSR_TILE + lx, py = tile_y * GSR_TILE + ly;
        const bool inside = px < W && py < H;
        const float pfx = (float)px, pfy = (float)py;
        const uint2 range = ranges[tile];
        const int n_inst = (int)(range.y - range.x);
        const int rounds = (n_inst + GSR_BLOCK - 1) / GSR_BLOCK;

        bool done = !inside;
        float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dacc = 0.f;
        uint32_t last_contributor = 0;

        // global -> register pipeline, one round (256 instances) ahead of the LDS image: the records of round rd
        // are already in (f0,f1,f2) when the round starts and the Gaussian index of round rd+1 is in g_next
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 f0 = zero4, f1 = zero4, f2 = zero4;
        uint32_t g_next = 0;
        if ((int)threadIdx.x < n_inst) {
            const uint32_t g = point_list[range.x + threadIdx.x];
            const float4 *rec = splat + 3 * (size_t)g;
            f0 = rec[0]; f1 = rec[1]; f2 = rec[2];
        }
        if (GSR_BLOCK + (int)threadIdx.x < n_inst) g_next = point_list[range.x + GSR_BLOCK + threadIdx.x];

        for (int rd = 0; rd < rounds; rd++) {
            if (__syncthreads_count(done ? 1 : 0) == GSR_BLOCK) break;
            s_rec0[threadIdx.x] = f0;  // zero records past the end of the list: alpha = 0, never valid
            s_rec1[threadIdx.x] = f1;
            s_rec2[threadIdx.x] = f2;
            __syncthreads();
            {
                const int nf = (rd + 1) * GSR_BLOCK + (int)threadIdx.x;
                f0 = zero4; f1 = zero4; f2 = zero4;
                if (nf < n_inst) {
                    const float4 *rec = splat + 3 * (size_t)g_next;
                    f0 = rec[0]; f1 = rec[1]; f2 = rec[2];
                }
                if (nf + GSR_BLOCK < n_inst) g_next = point_list[range.x + (uint32_t)(nf + GSR_BLOCK)];
            }
            const int cnt = min(GSR_BLOCK, n_inst - rd * GSR_BLOCK);
            const uint32_t pos0 = (uint32_t)(rd * GSR_BLOCK);
            for (int j = 0; j < cnt; j += kBatch) {
                if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
                float4 c0[kBatch], c1[kBatch];
#pragma unroll
                for (int k = 0; k < kBatch; k++) {  // one batch of ds_reads, one lgkmcnt wait
                    c0[k] = s_rec0[j + k];
                    c1[k] = s_rec1[j + k];
                }
                float alpha[kBatch];
                bool valid[kBatch];
                uint64_t any = 0ull;
#pragma unroll
                for (int k = 0; k < kBatch; k++) {
                    const float dx = c0[k].x - pfx, dy = c0[k].y - pfy;
                    const float q = fma_(c1[k].z * dy, dy, (c1[k].x * dx) * dx);
                    const float power = fma_(-(c1[k].y * dx), dy, -0.5f * q);
                    alpha[k] = fminf(0.99f, c1[k].w * __builtin_amdgcn_exp2f(power * 1.4426950408889634f));
                    valid[k] = power <= 0.0f && alpha[k] >= 1.0f / 255.0f;
                    any |= __builtin_amdgcn_ballot_w64(valid[k]);
                }
                if ((any & __builtin_amdgcn_ballot_w64(!done)) != 0ull) {
#pragma unroll
                    for (int k = 0; k < kBatch; k++) {
                        const bool hit = valid[k] && !done;
                        if (__builtin_amdgcn_ballot_w64(hit) == 0ull) continue;
                        const float4 r2 = s_rec2[j + k];
                        // non-hit lanes run with alpha 0: T * 1 = T (>= 1e-4 by construction) and weight 0
                        const float a_eff = hit ? alpha[k] : 0.0f;
                        const float test_T = T * (1.0f - a_eff);
                        const bool stop = test_T < 0.0001f;
                        const float w = stop ? 0.0f : a_eff * T;
                        C0 = fma_(r2.x, w, C0);
                        C1 = fma_(r2.y, w, C1);
                        C2 = fma_(r2.z, w, C2);
                        Dacc = fma_(c0[k].w, w, Dacc);
                        T = stop ? T : test_T;
                        last_contributor = (hit && !stop) ? pos0 + (uint32_t)(j + k) + 1u : last_contributor;
                        done = done || stop;
                    }
                }
            }
        }
        if (inside) {
            const size_t pid = (size_t)py * W + px;
            const size_t plane = (size_t)H * W;
            final_T[pid] = T;
            n_contrib[pid] = last_contributor;
            out_color[pid] = fma_(T, bg0, C0);
            out_color[plane + pid] = fma_(T, bg1, C1);
            out_color[2 * plane + pid] = fma_(T, bg2, C2);
            out_invdepth[pid] = Dacc;
        }
    }
}

// Longest-first tile order for the queue (radix-fallback path; the counting path orders inside tile_starts_kernel).
__global__ __launch_bounds__(GSR_BLOCK) void tile_order_kernel(const uint2 *__restrict__ ranges, int num_tiles,
                                                               uint32_t *__restrict__ order) {
    __shared__ uint32_t s_bins[64];
    __shared__ uint32_t s_red[4];
    gsr_tile_order_block(ranges, num_tiles, order, s_bins, s_red);
}

// GSWorld's frame conversion (gs_world_wrapper.py:268-270): CHW float -> HWC uint8, (x*255).clamp(0,255) then a
// truncating cast.  4 pixels (12 output bytes) per thread so that stores are three aligned dwords.
__global__ __launch_bounds__(GSR_BLOCK) void pack_rgb8_kernel(const float *__restrict__ color, int n_pix,
                                                              uint8_t *__restrict__ out) {
    const int q = blockIdx.x * GSR_BLOCK + (int)threadIdx.x;  // group of 4 pixels
    const int p0 = q * 4;
    if (p0 >= n_pix) return;
    uint8_t v[12];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int p = p0 + k;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float x = p < n_pix ? color[(size_t)c * n_pix + p] : 0.f;
            x = fminf(fmaxf(x * 255.0f, 0.0f), 255.0f);
            v[3 * k + c] = (uint8_t)x;
        }
    }
    if (p0 + 3 < n_pix) {
        uint32_t *o = reinterpret_cast<uint32_t *>(out + (size_t)p0 * 3);
        o[0] = v[0] | (v[1] << 8) | (v[2] << 16) | ((uint32_t)v[3] << 24);
        o[1] = v[4] | (v[5] << 8) | (v[6] << 16) | ((uint32_t)v[7] << 24);
        o[2] = v[8] | (v[9] << 8) | (v[10] << 16) | ((uint32_t)v[11] << 24);
    } else {
        for (int k = 0; k < 12 && p0 * 3 + k < n_pix * 3; k++) out[(size_t)p0 * 3 + k] = v[k];
    }
}

}  // namespace

extern "C" int gsr_pack_rgb8(const float *color, int32_t width, int32_t height, uint8_t *out, void *stream) {
    if (!color || !out || width <= 0 || height <= 0 || (reinterpret_cast<uintptr_t>(out) & 3u)) {
        gsr_set_error("gsr_pack_rgb8: null / unaligned pointer or empty image");
        return GSR_E_INVALID;
    }
    const int n_pix = width * height;
    hipLaunchKernelGGL(pack_rgb8_kernel, dim3(gsr_div_up(gsr_div_up(n_pix, 4), GSR_BLOCK)), dim3(GSR_BLOCK), 0,
                       (hipStream_t)stream, color, n_pix, out);
    return gsr_check_launch("pack_rgb8", false, (hipStream_t)stream);
}

int gsr_launch_render(const GsrSettings &st, const GeomState &g, const uint32_t *point_list, const ImageState &img,
                      const float *background, float *out_color, float *out_invdepth, bool order_ready,
                      hipStream_t stream) {
    const int W = st.image_width, H = st.image_height;
    const int gx = gsr_div_up(W, GSR_TILE), gy = gsr_div_up(H, GSR_TILE);
    if (g_render_variant == 2) {
        static int num_cus = 0;
        if (num_cus == 0) {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
                gsr_set_error("render: hipGetDeviceProperties failed");
                return GSR_E_HIP;
            }
            num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        }
        const int T = gx * gy;
        if (!order_ready)
            hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(GSR_BLOCK), 0, stream, img.ranges, T, img.tile_order);
        const int blocks = min(T, num_cus * g_render_blocks_per_cu);
        hipLaunchKernelGGL(render_queue_kernel, dim3(blocks), dim3(GSR_BLOCK), 0, stream, img.ranges, point_list,
                           g.splat, W, H, gx, T, img.tile_order, &g.hdr->tile_queue, background, out_color,
                           out_invdepth, img.final_T, img.n_contrib);
    } else if (g_render_variant == 0)
        hipLaunchKernelGGL(render_kernel, dim3(gx * gy), dim3(GSR_BLOCK), 0, stream, img.ranges, point_list, g.splat,
                           W, H, gx, background, out_color, out_invdepth, img.final_T, img.n_contrib);
    else
        hipLaunchKernelGGL(render_wave_kernel, dim3(gx * gy), dim3(GSR_BLOCK), 0, stream, img.ranges, point_list,
                           g.splat, W, H, gx, gx * gy, background, out_color, out_invdepth, img.final_T,
                           img.n_contrib);
    return GSR_OK;
}

// 0 = LDS-staged tile kernel, 1 = wave-independent readlane kernel, 2 = batched / queued kernel (default);
// variants 0 and 1 are kept for within-process A/B measurements.  blocks_per_cu sizes variant 2's persistent grid.
extern "C" int gsr_debug_set_render_variant(int variant, int blocks_per_cu) {
    if (variant < 0 || variant > 2 || blocks_per_cu < 0 || blocks_per_cu > 8) {
        gsr_set_error("gsr_debug_set_render_variant: variant must be 0..2, blocks_per_cu 0..8");
        return GSR_E_INVALID;
    }
    g_render_variant = variant;
    if (blocks_per_cu > 0) g_render_blocks_per_cu = blocks_per_cu;
    return GSR_OK;
}
