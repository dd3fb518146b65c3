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

int g_render_variant = 1;

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
                      const float *background, float *out_color, float *out_invdepth, hipStream_t stream) {
    const int W = st.image_width, H = st.image_height;
    const int gx = gsr_div_up(W, GSR_TILE), gy = gsr_div_up(H, GSR_TILE);
    if (g_render_variant == 0)
        hipLaunchKernelGGL(render_kernel, dim3(gx * gy), dim3(GSR_BLOCK), 0, stream, img.ranges, point_list, g.splat,
                           W, H, gx, background, out_color, out_invdepth, img.final_T, img.n_contrib);
    else
        hipLaunchKernelGGL(render_wave_kernel, dim3(gx * gy), dim3(GSR_BLOCK), 0, stream, img.ranges, point_list,
                           g.splat, W, H, gx, gx * gy, background, out_color, out_invdepth, img.final_T,
                           img.n_contrib);
    return GSR_OK;
}

// 0 = LDS-staged tile kernel (kept for A/B measurements), 1 = wave-independent kernel (default)
extern "C" int gsr_debug_set_render_variant(int variant) {
    if (variant < 0 || variant > 1) {
        gsr_set_error("gsr_debug_set_render_variant: variant must be 0 or 1");
        return GSR_E_INVALID;
    }
    g_render_variant = variant;
    return GSR_OK;
}
