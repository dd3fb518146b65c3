// render.hip -- front-to-back alpha compositing (upstream forward.cu renderCUDA).
//
// CDNA4 mapping: a wave owns one 8x8 quadrant of a 16x16 tile (lane -> (x&7, y>>3)), so that its 64 pixels are
// spatially compact and its alpha-test / early-exit decisions are coherent.  Four kernels, all bit-identical
// (tests/test_forward_gpu.py): render_kernel = upstream's structure, one tile per workgroup, the reference point of
// the bit-identity test; render_queue_kernel<CULL> = batched tile kernels; render_stream_kernel = the default, one
// quadrant per wave with per-quadrant instance culling (see the block comments below for what was measured).
//
// Arithmetic contract (matches oracle/gs_oracle.c gso_render): -ffp-contract=off, explicit fmaf; the only
// non-bit-reproducible operation is exp(): exp2(power * log2e) on the hardware transcendental unit.
#include "gsr_internal.h"

namespace {

// compositing kernel selection comes with every call (GsrSettings.render_variant / render_blocks_per_cu): the library
// keeps no mutable process state, so renderers on different threads / devices never see each other's choices
constexpr int kDefaultBlocksPerCu = 6;
struct RenderChoice {
    int variant;        // 4 = wave-decoupled stream kernel (default), 0 = tile kernel, 2 / 3 = batched tile kernels
    int blocks_per_cu;
};
inline RenderChoice render_choice(const GsrSettings &st) {
    RenderChoice c;
    c.variant = st.render_variant == 0 ? 4 : (st.render_variant == 1 ? 0 : st.render_variant);
    c.blocks_per_cu = st.render_blocks_per_cu > 0 ? st.render_blocks_per_cu : kDefaultBlocksPerCu;
    return c;
}

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
            const float alpha = fminf(0.99f, r1.w * gsr_exp_power(power));
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
//  * workgroups are persistent and take tiles dealt in descending list length, so the chip drains evenly
//    instead of waiting for the CU that happened to receive the long tiles.
// ---------------------------------------------------------------------------------------------------------
constexpr int kBatch = 4;

template <bool CULL>
__global__ __launch_bounds__(GSR_BLOCK) void render_queue_kernel(const uint2 *__restrict__ ranges,
                                                                 const uint32_t *__restrict__ point_list,
                                                                 const float4 *__restrict__ splat, int W, int H, int gx,
                                                                 int num_tiles, const uint32_t *__restrict__ tile_order,
                                                                 const float *__restrict__ bg,
                                                                 float *__restrict__ out_color,
                                                                 float *__restrict__ out_invdepth,
                                                                 float *__restrict__ final_T,
                                                                 uint32_t *__restrict__ n_contrib) {
    __shared__ float4 s_rec0[GSR_BLOCK + kBatch];
    __shared__ float4 s_rec1[GSR_BLOCK + kBatch];
    __shared__ float4 s_rec2[GSR_BLOCK + kBatch];
    const int lane = gsr_lane(), wave = gsr_wave();
    const int lx = ((wave & 1) << 3) | (lane & 7);
    const int ly = ((wave >> 1) << 3) | (lane >> 3);
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    if (threadIdx.x < kBatch) {  // padding entries: alpha = 0 -> never valid
        s_rec0[GSR_BLOCK + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
        s_rec1[GSR_BLOCK + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
        s_rec2[GSR_BLOCK + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // static longest-first deal (see render_stream_kernel: a global ticket counter costs more than it balances)
    for (uint32_t ticket = blockIdx.x; ticket < (uint32_t)num_tiles; ticket += gridDim.x) {
        __syncthreads();  // previous tile fully consumed (LDS records)
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
            // instances this wave has to look at: all of the round, or (CULL) those that can reach its quadrant.
            // The masks come out of ballots, i.e. they live in SGPRs and the walk below is scalar code.
            uint64_t m0 = ~0ull, m1 = ~0ull, m2 = ~0ull, m3 = ~0ull;
            if (CULL) {
                const float qx = (float)(tile_x * GSR_TILE + ((wave & 1) << 3));
                const float qy = (float)(tile_y * GSR_TILE + ((wave >> 1) << 3));
                uint64_t mm[4];
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const float4 a = s_rec0[s * GSR_WAVE + lane], b = s_rec1[s * GSR_WAVE + lane];
                    mm[s] = __builtin_amdgcn_ballot_w64(quadrant_may_hit(a.x, a.y, b, qx, qy));
                }
                m0 = mm[0]; m1 = mm[1]; m2 = mm[2]; m3 = mm[3];
            }
            int base = 0;
            for (int j = 0; CULL || j < cnt; j += kBatch) {  // (CULL: the masks, not j, end the walk)
                if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
                int jj[kBatch];
                if (CULL) {
                    // next kBatch surviving positions in list order; short batches are padded with the zero records
#pragma unroll
                    for (int k = 0; k < kBatch; k++) {
                        while (m0 == 0ull && base < 3 * GSR_WAVE) {
                            m0 = m1; m1 = m2; m2 = m3; m3 = 0ull;
                            base += GSR_WAVE;
                        }
                        if (m0 != 0ull) {
                            jj[k] = base + (int)__builtin_ctzll(m0);
                            m0 &= m0 - 1ull;
                        } else {
                            jj[k] = GSR_BLOCK + k;
                        }
                    }
                    if (jj[0] >= GSR_BLOCK) break;  // nothing left in this round
                } else {
#pragma unroll
                    for (int k = 0; k < kBatch; k++) jj[k] = j + k;
                }
                float4 c0[kBatch], c1[kBatch];
#pragma unroll
                for (int k = 0; k < kBatch; k++) {  // one batch of ds_reads, one lgkmcnt wait
                    c0[k] = s_rec0[jj[k]];
                    c1[k] = s_rec1[jj[k]];
                }
                float alpha[kBatch];
                bool valid[kBatch];
                uint64_t any = 0ull;
#pragma unroll
                for (int k = 0; k < kBatch; k++) {
                    const float dx = c0[k].x - pfx, dy = c0[k].y - pfy;
                    const float q = fma_(c1[k].z * dy, dy, (c1[k].x * dx) * dx);
                    const float power = fma_(-(c1[k].y * dx), dy, -0.5f * q);
                    alpha[k] = fminf(0.99f, c1[k].w * gsr_exp_power(power));
                    valid[k] = power <= 0.0f && alpha[k] >= 1.0f / 255.0f;
                    any |= __builtin_amdgcn_ballot_w64(valid[k]);
                }
                if ((any & __builtin_amdgcn_ballot_w64(!done)) != 0ull) {
#pragma unroll
                    for (int k = 0; k < kBatch; k++) {
                        const bool hit = valid[k] && !done;
                        if (__builtin_amdgcn_ballot_w64(hit) == 0ull) continue;
                        const float4 r2 = s_rec2[jj[k]];
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
                        last_contributor = (hit && !stop) ? pos0 + (uint32_t)jj[k] + 1u : last_contributor;
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

// ---------------------------------------------------------------------------------------------------------
// Variant 4 (default): wave-decoupled compositing.
//
// Counters of variant 3 on the config-2 frame (profiles/round1/pmc): the 4 waves of a tile spend half their life
// parked (workgroup barriers around every 256-instance round, lgkmcnt waits inside 4 serial blend steps, a scalar
// walk over the survivor masks that issues as many SALU as VALU instructions), and the kernel ends with the few
// longest tiles running alone.  Here the unit of work is one 8x8 QUADRANT = one wave (4 x tiles units, dealt
// statically: every unit has its own resident wave when they fit, a longest-first snake otherwise); a wave shares
// nothing with the other waves of its workgroup:
//  * per round it gathers 64 instances (one per lane: index, then the 3 x 16 B record), one round ahead;
//  * each lane runs quadrant_may_hit on its candidate; survivors are compacted (ballot + mbcnt rank) into
//    the wave's private LDS list, in list order, pair-interleaved, with their list position alongside;
//  * the replay reads the list back 4 consecutive survivors at a time -- 11 LDS reads off one base register
//    with immediate offsets -- evaluates them two per packed instruction and composites in the same operation order
//    as variants 0 / 2 / 3 (bit-identical image state).
// No workgroup barrier, no scalar bit-walk, a quadrant retires the moment its 64 pixels saturate.  LDS traffic
// (48 B per survivor per wave) stays below the VALU time.  Which tiles share a CU is decided by their cost in the
// previous frame on the same renderer state (quad_work, tile_starts_kernel).
// ---------------------------------------------------------------------------------------------------------
#ifndef GSR_STREAM_ITEMS
#define GSR_STREAM_ITEMS 1
#endif
constexpr int kStreamLanesItems = GSR_STREAM_ITEMS;                              // candidates per lane per round (A/B: 1 ~ 2 > 3;
                                                                  // 1 halves the LDS list: 13 KiB per workgroup)
constexpr int kStreamRound = kStreamLanesItems * GSR_WAVE;        // 64 candidates per round
constexpr int kStreamList = kStreamRound + kBatch;                // survivors + zero padding of the last batch

typedef float v2f __attribute__((ext_vector_type(2)));

#ifndef GSR_STREAM_MASKED_BLEND
#define GSR_STREAM_MASKED_BLEND 1   // 0: the select-based blend everywhere (A/B)
#endif
#ifndef GSR_STREAM_ALIVE_BLEND
#define GSR_STREAM_ALIVE_BLEND 1    // 0: stream_blend_masked (finished pixels carried in T's sign, round 4's first version)
#endif
#ifndef GSR_STREAM_STAMPS
#define GSR_STREAM_STAMPS 0   // 1: tuning build that leaves per-quadrant cycle stamps in the image state
#endif

// one batch of survivors, evaluated for this lane's pixel (alpha: 0 for an instance that does not touch it)
struct StreamBatch {
    float alpha[kBatch], pos[kBatch];
    bool valid[kBatch];
    float4 col[kBatch];
    float araw[kBatch];  // min(0.99, opacity * exp(power)) before the alpha >= 1/255 test (stream_blend_masked)
    v2f ao[kBatch];      // (araw, 1 - araw): one packed multiply by T gives the weight and the new T (stream_blend_alive)
};

// same operations in the same order as the scalar kernels, two survivors per packed instruction
// B^2 <= (1 - 1e-3) A C with A, C > 0 (and A C out of the subnormal range): the exact power is then at most
// -2.5e-4 (A dx^2 + C dy^2), while the rounding errors of the five float operations that compute it stay below
// 2e-7 of that sum, so the float result is <= 0 for every pixel offset (an exact 0 offset gives -0).
__device__ __forceinline__ bool stream_conic_is_safe(const float4 conic_op) {
    const float ac = conic_op.x * conic_op.z;
    return conic_op.x > 0.0f && conic_op.z > 0.0f && ac > 1e-20f && conic_op.y * conic_op.y <= 0.999f * ac;
}

// CHECK_POWER = false: every survivor of the round has a comfortably positive definite conic (stream_conic_is_safe), for
// which the computed power cannot come out positive -- the `power <= 0` half of the validity test is then dropped.
template <bool CHECK_POWER, bool TRACK>
__device__ __forceinline__ StreamBatch stream_eval(const float4 (*list)[6], int i, v2f pf2x, v2f pf2y) {
    StreamBatch b;
#pragma unroll
    for (int h = 0; h < kBatch / 2; h++) {
        const float4 *pr = list[(i >> 1) + h];
        const float4 qxy = pr[0], qac = pr[1], qbo = pr[2];
        b.col[2 * h] = pr[3];
        b.col[2 * h + 1] = pr[4];
        const float2 qpos = TRACK ? *reinterpret_cast<const float2 *>(pr + 5) : make_float2(0.f, 0.f);
        const v2f dx = v2f{qxy.x, qxy.y} - pf2x, dy = v2f{qxy.z, qxy.w} - pf2y;
        const v2f cA = {qac.x, qac.y}, cC = {qac.z, qac.w};
        const v2f cB = {qbo.x, qbo.y}, op = {qbo.z, qbo.w};
        const v2f q = __builtin_elementwise_fma(cC * dy, dy, (cA * dx) * dx);   // = -0.5 (A dx^2 + C dy^2): pre-scaled
        const v2f power = __builtin_elementwise_fma(cB * dx, dy, q);             // cB = -B
        // (gsr_exp_power of gsr_internal.h, two survivors per packed instruction: the same operations per element)
        const v2f kL = {1.4426950408889634f, 1.4426950408889634f};
        const v2f pe = power * kL;
#if GSR_EXP_ACCURATE
        constexpr float L_lo = (float)(1.4426950408889634073599246810019 - (double)1.4426950408889634f);
        const v2f e = __builtin_elementwise_fma(power, v2f{L_lo, L_lo}, __builtin_elementwise_fma(power, kL, -pe));
        const v2f x = {__builtin_amdgcn_exp2f(pe.x), __builtin_amdgcn_exp2f(pe.y)};
        const v2f a = op * __builtin_elementwise_fma(x * v2f{0.6931471805599453f, 0.6931471805599453f}, e, x);
#else
        const v2f a = op * v2f{__builtin_amdgcn_exp2f(pe.x), __builtin_amdgcn_exp2f(pe.y)};
#endif
        const float a0 = fminf(0.99f, a.x), a1 = fminf(0.99f, a.y);
        b.valid[2 * h] = (!CHECK_POWER || power.x <= 0.0f) && a0 >= 1.0f / 255.0f;
        b.valid[2 * h + 1] = (!CHECK_POWER || power.y <= 0.0f) && a1 >= 1.0f / 255.0f;
        b.alpha[2 * h] = b.valid[2 * h] ? a0 : 0.0f;
        b.alpha[2 * h + 1] = b.valid[2 * h + 1] ? a1 : 0.0f;
        b.araw[2 * h] = a0;
        b.araw[2 * h + 1] = a1;
        b.ao[2 * h] = v2f{a0, 1.0f - a0};
        b.ao[2 * h + 1] = v2f{a1, 1.0f - a1};
        b.pos[2 * h] = qpos.x;
        b.pos[2 * h + 1] = qpos.y;
    }
    return b;
}

// TRACK = false (inference frames): n_contrib is not written, so the last contributor is not tracked either
template <bool TRACK>
__device__ __forceinline__ void stream_blend(const StreamBatch &b, float &T, v2f &acc_rg, v2f &acc_bd, uint32_t &last) {
    // (1 - alpha) of two survivors per packed subtraction: the same IEEE operation per element
    float oma[kBatch];
#pragma unroll
    for (int h = 0; h < kBatch / 2; h++) {
        const v2f d = v2f{1.0f, 1.0f} - v2f{b.alpha[2 * h], b.alpha[2 * h + 1]};
        oma[2 * h] = d.x;
        oma[2 * h + 1] = d.y;
    }
#pragma unroll
    for (int k = 0; k < kBatch; k++) {
        // alpha 0: T * 1 = T (>= 1e-4 by construction) and weight 0; T < 0 (finished): stop again
        const float test_T = T * oma[k];
        const bool stop = test_T < 0.0001f;
        const float w = stop ? 0.0f : b.alpha[k] * T;
        acc_rg = __builtin_elementwise_fma(v2f{b.col[k].x, b.col[k].y}, v2f{w, w}, acc_rg);
        acc_bd = __builtin_elementwise_fma(v2f{b.col[k].z, b.col[k].w}, v2f{w, w}, acc_bd);
        if (TRACK) last = (b.valid[k] && !stop) ? __float_as_uint(b.pos[k]) : last;
        T = stop ? -fabsf(T) : test_T;
    }
}

// The same blend with the alpha >= 1/255 test as an EXEC mask instead of a select (inference frames, safe conics: the
// test is the only condition).  stream_blend zeroes the alpha of a lane the instance does not touch (v_cmp + v_cndmask)
// and then runs the whole update on it with weight 0; here v_cmpx narrows EXEC to the lanes it touches, the update runs on
// those only, and EXEC is restored: one VALU instruction less per survivor (8 instead of 9 + the shared 1 - alpha) in a
// kernel that is bound by exactly that (17.75 per survivor).  Same operations on the same operands
// in the lanes that matter: bit-identical (tests: compositing variants, forward_only frames).  Hand-placed wait states:
// gfx950 wants two between a VALU write of VCC and its use as a lane mask; v94 / v95 are scratch (w | test_T).
__device__ __forceinline__ void stream_blend_masked(const StreamBatch &b, float &T, v2f &acc_rg, v2f &acc_bd) {
    float oma[kBatch];
#pragma unroll
    for (int h = 0; h < kBatch / 2; h++) {
        const v2f d = v2f{1.0f, 1.0f} - v2f{b.araw[2 * h], b.araw[2 * h + 1]};
        oma[2 * h] = d.x;
        oma[2 * h + 1] = d.y;
    }
    const uint64_t saved = __builtin_amdgcn_read_exec();  // (the lanes this batch runs on: restored after every survivor)
#pragma unroll
    for (int k = 0; k < kBatch; k++) {
        const v2f crg = {b.col[k].x, b.col[k].y}, cbd = {b.col[k].z, b.col[k].w};
        asm volatile(
            "v_cmpx_le_f32 0x3b808081, %[a]\n\t"          // EXEC &= 1/255 <= alpha
            "v_mul_f32 v95, %[T], %[oma]\n\t"              // test_T = T (1 - alpha)
            "v_cmp_gt_f32 vcc, 0x38d1b717, v95\n\t"        // stop = test_T < 1e-4
            "v_mul_f32 v94, %[a], %[T]\n\t"                // w = alpha T
            "s_nop 0\n\t"
            "v_cndmask_b32_e64 v94, v94, 0, vcc\n\t"       // stop: weight 0
            "v_cndmask_b32_e64 %[T], v95, -|%[T]|, vcc\n\t"  // stop: T = -|T| (finished), else test_T
            "v_pk_fma_f32 %[rg], %[crg], v[94:95], %[rg] op_sel_hi:[1,0,1]\n\t"
            "v_pk_fma_f32 %[bd], %[cbd], v[94:95], %[bd] op_sel_hi:[1,0,1]\n\t"
            "s_mov_b64 exec, %[sv]"
            : [T] "+v"(T), [rg] "+v"(acc_rg), [bd] "+v"(acc_bd)
            : [a] "v"(b.araw[k]), [oma] "v"(oma[k]), [crg] "v"(crg), [cbd] "v"(cbd), [sv] "s"(saved)
            : "vcc", "v94", "v95");
    }
}

// The masked blend with the FINISHED state as a wave mask too.  stream_blend_masked still pays, per survivor, a compare and
// two selects for the saturation test (weight 0 and T = -|T| in the lanes that stop) and two multiplies for alpha T and
// T (1 - alpha).  Here
//   * `alive` (an SGPR pair) holds the lanes that still composite; every survivor starts from EXEC = alive;
//   * v_cmpx narrows EXEC to the lanes the survivor touches (alpha >= 1/255), ONE packed multiply (alpha, 1 - alpha) x T
//     gives the weight and test_T, a plain v_cmp leaves the touched lanes that stop (test_T < 1e-4; 0 in inactive
//     lanes) in VCC, and one s_andn2 each takes them out of `alive` and of EXEC: T and the four accumulators are updated
//     in the lanes that go on only.  A lane that stops keeps the T it had, as the reference's `break` does.
// 6 VALU + 3 scalar instructions per survivor (+ 1 VALU for 1 - alpha) against 8 + 1 (+ 1/2).  Scalar instructions are
// not free on this chip (tools/ubench_issue.hip, profiles/round4/ubench_issue.txt; 5 waves per SIMD: one costs ~0.7 of a
// plain VALU instruction's issue time, a packed fp32 one 1.8, v_exp_f32 5.1) -- a first version with both tests as v_cmpx
// and the stopped lanes as touched ^ going-on took five and gained half as much.  The same IEEE operations on the same
// operands in every lane that matters: bit-identical (tests: forward_only frames against default frames, compositing
// variants).  T lives in v92 (the low half of the pinned pair v[92:93]: the packed multiply wants a register pair, the
// update a single register); v94 / v95 = weight | test_T.
// A step expects EXEC = alive and leaves EXEC = NEXT (alive for the next survivor, the saved mask behind the last one).
#define GSR_ALIVE_STEP(A, AO, CRG, CBD, NEXT)                                                                        \
    "v_cmpx_le_f32 0x3b808081, " A "\n\t"                              /* EXEC = alive & 1/255 <= alpha */             \
    "v_pk_mul_f32 v[94:95], " AO ", v[92:93] op_sel_hi:[1,0]\n\t"      /* alpha T | (1 - alpha) T */                   \
    "v_cmp_gt_f32 vcc, 0x38d1b717, v95\n\t"                            /* VCC = touched & test_T < 1e-4 */             \
    "s_andn2_b64 %[alive], %[alive], vcc\n\t"                                                                         \
    "s_andn2_b64 exec, exec, vcc\n\t"                                                                                 \
    "v_mov_b32 v92, v95\n\t"                                                                                          \
    "v_pk_fma_f32 %[rg], " CRG ", v[94:95], %[rg] op_sel_hi:[1,0,1]\n\t"                                              \
    "v_pk_fma_f32 %[bd], " CBD ", v[94:95], %[bd] op_sel_hi:[1,0,1]\n\t"                                              \
    "s_mov_b64 exec, " NEXT "\n\t"
__device__ __forceinline__ void stream_blend_alive(const StreamBatch &b, v2f &Tp, v2f &acc_rg, v2f &acc_bd,
                                                   uint64_t &alive, int &limit, const uint64_t saved) {
    static_assert(kBatch == 4, "the block below takes four survivors");
    const v2f crg0 = {b.col[0].x, b.col[0].y}, cbd0 = {b.col[0].z, b.col[0].w};
    const v2f crg1 = {b.col[1].x, b.col[1].y}, cbd1 = {b.col[1].z, b.col[1].w};
    const v2f crg2 = {b.col[2].x, b.col[2].y}, cbd2 = {b.col[2].z, b.col[2].w};
    const v2f crg3 = {b.col[3].x, b.col[3].y}, cbd3 = {b.col[3].z, b.col[3].w};
    asm volatile("s_mov_b64 exec, %[alive]\n\t"
                 GSR_ALIVE_STEP("%[a0]", "%[ao0]", "%[crg0]", "%[cbd0]", "%[alive]")
                 GSR_ALIVE_STEP("%[a1]", "%[ao1]", "%[crg1]", "%[cbd1]", "%[alive]")
                 GSR_ALIVE_STEP("%[a2]", "%[ao2]", "%[crg2]", "%[cbd2]", "%[alive]")
                 GSR_ALIVE_STEP("%[a3]", "%[ao3]", "%[crg3]", "%[cbd3]", "%[sv]")
                 "s_cmp_eq_u64 %[alive], 0\n\t"
                 "s_cmov_b32 %[lim], 0"
                 : "+{v[92:93]}"(Tp), [rg] "+v"(acc_rg), [bd] "+v"(acc_bd), [alive] "+s"(alive), [lim] "+s"(limit)
                 : [a0] "v"(b.araw[0]), [ao0] "v"(b.ao[0]), [crg0] "v"(crg0), [cbd0] "v"(cbd0),
                   [a1] "v"(b.araw[1]), [ao1] "v"(b.ao[1]), [crg1] "v"(crg1), [cbd1] "v"(cbd1),
                   [a2] "v"(b.araw[2]), [ao2] "v"(b.ao[2]), [crg2] "v"(crg2), [cbd2] "v"(cbd2),
                   [a3] "v"(b.araw[3]), [ao3] "v"(b.ao[3]), [crg3] "v"(crg3), [cbd3] "v"(cbd3),
                   [sv] "s"(saved)
                 : "vcc", "scc", "v94", "v95");  // (s_andn2 / s_cmp write SCC)
}

// survivor number `rank` of the round -> its half of pair rank / 2 (layout: see s_list in render_stream_kernel)
__device__ __forceinline__ void stream_list_put(float4 (*list)[6], int rank, float4 geo, float4 conic_op, float4 rgbd,
                                                float pos) {
    float4 *pr = list[rank >> 1];
    float *f = reinterpret_cast<float *>(pr) + (rank & 1);
    f[0] = geo.x;
    f[2] = geo.y;
    // the quadratic form is stored pre-multiplied: -A/2, -C/2, -B.  Scaling by a power of two commutes with every
    // rounding below, so power = fma((-B) dx, dy, fma((-C/2) dy, dy, ((-A/2) dx) dx)) has the bits of
    // fma(-(B dx), dy, -0.5 fma(C dy, dy, (A dx) dx)) -- one packed multiply less per survivor pair
    f[4] = -0.5f * conic_op.x;
    f[6] = -0.5f * conic_op.z;
    f[8] = -conic_op.y;
    f[10] = conic_op.w;
    pr[3 + (rank & 1)] = rgbd;
    f[20] = pos;
}

// ---------------------------------------------------------------------------------------------------------
// Cooperative quadrants (inference frames).  A wave is one in-order instruction stream: cull a round of 64 candidates
// (one exposed gather latency, ~1 500 cycles when the wave is alone on its SIMD), replay the survivors, next round.
// The kernel lasts as long as its longest such chain, and under a camera that sees most of the scene the longest ones
// are mostly CULL: a quadrant on the robot's base walks a 4 900-entry super-tile list in 77 rounds for 448 survivors
// (profiles/round4/stream_stamps_dense_view.txt: 197 k cycles of which 115 k are the cull; the compositor lasts 96 us
// for 36 us of mean SIMD time).  The quadrants that were costliest in the previous frame on this state
// (ss_quad_order_1024: the first few of every XCD's cost order, if clearly above the average) therefore get a WORKGROUP
// behind the main grid instead of a wave in it: waves 1-3 cull the rounds r = w - 1 (mod 3) into their own LDS lists --
// three gathers in flight instead of one, and none of them waits for a replay -- and wave 0 replays the lists in round
// order, exactly the batches the quadrant's own wave would have replayed: same operations in the same order on the same
// operands, the image does not change by a bit.  Hand-off through two counters per culling wave in LDS (rounds
// published / rounds consumed): LDS operations of a CU execute in issue order, so a list written before the counter
// is complete when the counter is seen; workgroup-scope fences keep the compiler from moving accesses across them.
// ---------------------------------------------------------------------------------------------------------
constexpr int kCoopCullers = GSR_BLOCK / GSR_WAVE - 1;  // 3
#ifndef GSR_COOP_SPIN_LIMIT
#define GSR_COOP_SPIN_LIMIT (1u << 20)                  // polls of 64+ cycles each: ~50 ms
#endif
constexpr uint32_t kCoopSpinLimit = GSR_COOP_SPIN_LIMIT;  // (the test library libgsr_hip_coopspin.so is built with 1)

struct CoopFlags {
    uint32_t ready[4], done[4], n_surv[4], unsafe_lo[4], unsafe_hi[4], stop;
};

__device__ __forceinline__ uint32_t coop_load(volatile uint32_t *p) { return *p; }

// -> true (replaying wave only): a hand-off timed out and the quadrant was written TRUNCATED -- the caller reports it in the
// frame header (GsrHeader::coop_timeouts; gsr_frame_stats), the way a binning overflow is reported.
template <bool SUPER>
__device__ __forceinline__ bool render_coop_quadrant(float4 (*s_list)[kStreamList / 2][6], CoopFlags *fl, const uint32_t q,
                                                     const uint2 *__restrict__ ranges,
                                                     const uint32_t *__restrict__ point_list,
                                                     const float4 *__restrict__ splat, int W, int H, int gx,
                                                     const float *__restrict__ bg, float *__restrict__ out_color,
                                                     float *__restrict__ out_invdepth, uint8_t *__restrict__ rgb8,
                                                     uint32_t *__restrict__ quad_work, float *__restrict__ final_T) {
    static_assert(SUPER, "cooperative quadrants are an inference-frame path");
#if GSR_STREAM_STAMPS
    const uint64_t stamp0 = __builtin_readcyclecounter();
#endif
    const int lane = gsr_lane(), wave = gsr_wave();
    const int tile = (int)(q >> 2), quad = (int)(q & 3u);
    const int qx0 = (tile % gx) * GSR_TILE + ((quad & 1) << 3);
    const int qy0 = (tile / gx) * GSR_TILE + ((quad >> 1) << 3);
    const uint32_t tx = (uint32_t)(tile % gx), ty = (uint32_t)(tile / gx);
    constexpr int kSX = GSR_SUPER_SX, kSY = GSR_SUPER_SY;
    const uint2 range = ranges[(int)((ty >> kSY) * (uint32_t)((gx + (1 << kSX) - 1) >> kSX) + (tx >> kSX))];
    const int n_inst = (int)(range.y - range.x);
    const uint32_t *src = point_list + range.x;
    const int rounds = (n_inst + GSR_WAVE - 1) / GSR_WAVE;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    volatile uint32_t *ready = fl->ready, *done = fl->done, *stop = &fl->stop;
    if (wave != 0) {
        // ---- a culling wave: rounds c, c + 3, c + 6 ...
        const int c = wave - 1;
        float4(*list)[6] = s_list[wave];
        const float qxf = (float)qx0, qyf = (float)qy0;
        float4 f0 = zero4, f1 = zero4, f2 = zero4;
        uint32_t g_next = 0;
        {
            const int p = c * GSR_WAVE + lane;
            if (p < n_inst) {
                const float4 *rec = splat + 3 * (size_t)src[p];
                f0 = rec[0]; f1 = rec[1]; f2 = rec[2];
            }
            if (p + kCoopCullers * GSR_WAVE < n_inst) g_next = src[p + kCoopCullers * GSR_WAVE];
        }
        uint32_t produced = 0;
        for (int rd = c; rd < rounds; rd += kCoopCullers) {
            const int p = rd * GSR_WAVE + lane;
            bool keep = p < n_inst;
            {  // getRect of upstream: rect_min <= tile < rect_max on both axes
                const uint32_t rb = __float_as_uint(f2.w);
                keep = keep && tx - (rb & 255u) < ((rb >> 16) & 255u) - (rb & 255u) &&
                       ty - ((rb >> 8) & 255u) < (rb >> 24) - ((rb >> 8) & 255u);
            }
            keep = keep && quadrant_may_hit<true>(f0.x, f0.y, f1, qxf, qyf, 7.0f, f0.z);
            const uint64_t unsafe = __builtin_amdgcn_ballot_w64(keep && (__float_as_uint(f0.z) & 1u) == 0u);
            const uint64_t mask = __builtin_amdgcn_ballot_w64(keep);
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            const int n_surv = (int)__builtin_popcountll(mask);
            // my list is free once the replay has consumed what I published last
            // (the replaying wave bounds ITS waits -- kCoopSpinLimit -- and raises `stop` when it leaves, whatever the reason)
            while (coop_load(&done[wave]) != produced && coop_load(stop) == 0u) __builtin_amdgcn_s_sleep(1);
            if (coop_load(stop) != 0u) break;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (keep)
                stream_list_put(list, rank, f0, f1, make_float4(f2.x, f2.y, f2.z, f0.w), __uint_as_float((uint32_t)p + 1u));
            if (lane < kBatch) stream_list_put(list, n_surv + lane, zero4, zero4, zero4, 0.0f);
            if (lane == 0) {
                fl->n_surv[wave] = (uint32_t)n_surv;
                fl->unsafe_lo[wave] = (uint32_t)unsafe;
                fl->unsafe_hi[wave] = (uint32_t)(unsafe >> 32);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            produced++;
            if (lane == 0) ready[wave] = produced;
            // my next round's records (its indices were requested a round ago)
            const int pn = (rd + kCoopCullers) * GSR_WAVE + lane;
            f0 = zero4; f1 = zero4; f2 = zero4;
            if (pn < n_inst) {
                const float4 *rec = splat + 3 * (size_t)g_next;
                f0 = rec[0]; f1 = rec[1]; f2 = rec[2];
            }
            if (pn + kCoopCullers * GSR_WAVE < n_inst) g_next = src[pn + kCoopCullers * GSR_WAVE];
        }
        return false;
    }
    // ---- the replaying wave
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pfx = (float)px, pfy = (float)py;
    const v2f pf2x = {pfx, pfx}, pf2y = {pfy, pfy};
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    float T = inside ? 1.0f : -1.0f;
    v2f acc_rg = {0.f, 0.f}, acc_bd = {0.f, 0.f};
    uint32_t work = 0;
    bool timed_out = false;
    for (int rd = 0; rd < rounds; rd++) {
        if (__builtin_amdgcn_ballot_w64(T > 0.0f) == 0ull) break;
        const int w = 1 + rd % kCoopCullers;
        const uint32_t want = (uint32_t)(rd / kCoopCullers) + 1u;  // (culling wave w has published rounds w - 1, w + 2, ... )
        // (a hand-off that never comes -- none can, by the counters' construction -- must not hang the GPU)
        uint32_t spin;  // (a vector register: the kernel has no scalar one to spare)
        asm volatile("v_mov_b32 %0, 0" : "=v"(spin));
        while (coop_load(&ready[w]) != want && spin < kCoopSpinLimit) {
            __builtin_amdgcn_s_sleep(1);
            spin++;
        }
        if (spin >= kCoopSpinLimit) {
            timed_out = true;
            break;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // (plain loads behind the acquire fence, one wait for the three: as volatile ones they were three LDS round trips in a
        //  row per round -- dense view alone -2 %)
        const int n_surv = (int)__builtin_amdgcn_readfirstlane((int)fl->n_surv[w]);
        const uint64_t unsafe = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)fl->unsafe_hi[w]) << 32) |
                                (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)fl->unsafe_lo[w]);
        const float4(*list)[6] = s_list[w];
        work += 8u;
#if GSR_STREAM_MASKED_BLEND && GSR_STREAM_ALIVE_BLEND
        if (unsafe == 0ull) {
            uint64_t alive = __builtin_amdgcn_ballot_w64(T > 0.0f);
            v2f Tp = {T, 0.0f};
            int i = 0;
            int limit = n_surv;
            const uint64_t full = __builtin_amdgcn_read_exec();
            if (n_surv > 0) do {
                const StreamBatch b = stream_eval<false, false>(list, i, pf2x, pf2y);
                stream_blend_alive(b, Tp, acc_rg, acc_bd, alive, limit, full);
                work += (uint32_t)kBatch;
                i += kBatch;
            } while (i < limit);
            asm volatile("v_cndmask_b32_e64 %0, -|%1|, %1, %2" : "=v"(T) : "v"(Tp.x), "s"(alive));
        } else
#endif
        if (unsafe == 0ull) {
            int i = 0;
            if (n_surv > 0) do {
                const StreamBatch b = stream_eval<false, false>(list, i, pf2x, pf2y);
#if GSR_STREAM_MASKED_BLEND
                stream_blend_masked(b, T, acc_rg, acc_bd);
#else
                uint32_t last = 0;
                stream_blend<false>(b, T, acc_rg, acc_bd, last);
#endif
                work += (uint32_t)kBatch;
                i += kBatch;
            } while (i < n_surv && __builtin_amdgcn_ballot_w64(T > 0.0f) != 0ull);
        } else {
            int i = 0;
            uint32_t last = 0;
            if (n_surv > 0) do {
                const StreamBatch b = stream_eval<true, false>(list, i, pf2x, pf2y);
                stream_blend<false>(b, T, acc_rg, acc_bd, last);
                work += (uint32_t)kBatch;
                i += kBatch;
            } while (i < n_surv && __builtin_amdgcn_ballot_w64(T > 0.0f) != 0ull);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) done[w] = want;
    }
    if (lane == 0) *stop = 1u;
    if (quad_work != nullptr && lane == 0) quad_work[q] = work;
#if GSR_STREAM_STAMPS
    if (lane == 0) {  // (tuning build: the replaying wave's life, as the quadrant's own wave reports it in the main grid)
        const uint64_t stamp1 = __builtin_readcyclecounter();
        const uint32_t hw = (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0x0fffffffu;
        const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20);
        reinterpret_cast<uint4 *>(final_T)[q] = make_uint4((uint32_t)stamp0, (uint32_t)stamp1, hw | (xcc << 28), work);
    }
#endif
    if (inside) {
        const size_t pid = (size_t)py * W + px;
        const size_t plane = (size_t)H * W;
        T = fabsf(T);
        const float r = fma_(T, bg0, acc_rg.x), g = fma_(T, bg1, acc_rg.y), b = fma_(T, bg2, acc_bd.x);
        if (out_color != nullptr) {
            out_color[pid] = r;
            out_color[plane + pid] = g;
            out_color[2 * plane + pid] = b;
            out_invdepth[pid] = acc_bd.y;
        }
        if (rgb8) {
            uint8_t *o = rgb8 + 3 * pid;
            o[0] = (uint8_t)fminf(fmaxf(r * 255.0f, 0.0f), 255.0f);
            o[1] = (uint8_t)fminf(fmaxf(g * 255.0f, 0.0f), 255.0f);
            o[2] = (uint8_t)fminf(fmaxf(b * 255.0f, 0.0f), 255.0f);
        }
    }
    return timed_out;
}

// SUPER (GsrSettings.forward_only): `ranges` / `point_list` are the lists of SUPER-TILES (2 x 1 tiles: 0.58 of the
// instances to place and fetch at config 2; gsr_internal.h GSR_SUPER_SX / SY).  A candidate then passes the reference's own tile test first -- its tile
// rect (four bytes in the spare word of the colour record, preprocess.hip) must contain this wave's tile -- so the
// wave composites exactly the depth-ordered list of its 16 x 16 tile and the image is bit-identical.  final_T /
// n_contrib (read by the backward only) are not written.
template <bool SUPER>
__device__ __forceinline__ void render_stream_body(const uint2 *__restrict__ ranges,
                                                                  const uint32_t *__restrict__ point_list,
                                                                  const float4 *__restrict__ splat, int W, int H, int gx,
                                                                  int num_tiles, const uint32_t *__restrict__ tile_order,
                                                                  const float *__restrict__ bg,
                                                                  float *__restrict__ out_color,
                                                                  float *__restrict__ out_invdepth,
                                                                  float *__restrict__ final_T,
                                                                  uint32_t *__restrict__ n_contrib,
                                                                  uint8_t *__restrict__ rgb8 /* optional */,
                                                                  uint32_t *__restrict__ quad_work /* optional */,
                                                                  int num_cus, int main_blocks,
                                                                  const uint32_t *__restrict__ split_flag,
                                                                  const uint32_t *__restrict__ split_list,
                                                                  const uint32_t *__restrict__ split_count,
                                                                  uint32_t *__restrict__ quad_work_b,
                                                                  const uint32_t *__restrict__ quad_order,
                                                                  const uint32_t *__restrict__ coop_list,
                                                                  int coop_blocks, GsrHeader *__restrict__ hdr,
                                                                  const uint32_t wgx /* workgroup of this frame */,
                                                                  const uint32_t *__restrict__ tile_dirty = nullptr) {
    // survivors are stored in PAIRS, component-interleaved, so that the replay reads register pairs it can feed to
    // the packed fp32 pipe (v_pk_mul/fma_f32: two survivors per instruction for the alpha evaluation, two
    // accumulators per instruction for the blend).  One pair = 6 x 16 B:
    //   [0] x0 x1 y0 y1   [1] A0 A1 C0 C1   [2] B0 B1 o0 o1   [3] r0 g0 b0 d0   [4] r1 g1 b1 d1   [5] pos0 pos1 - -
    __shared__ float4 s_list[GSR_BLOCK / GSR_WAVE][kStreamList / 2][6];
    const int lane = gsr_lane(), wave = gsr_wave();
    // Tile reuse (round 6; inference frames whose only output is the uint8 frame, under the block cache of preprocess.hip): a
    // tile's pixels depend on the splat records in its list, their order, the camera and the background.  When camera,
    // background, settings and output buffer are the previous frame's on this state (td_reuse) and no Gaussian whose records
    // this frame's preprocess recomputed touches the tile -- now, or in the previous frame -- the list holds the same records
    // in the same order and the buffer already holds the pixels: the tile is not composited.  preprocess marks the tiles of
    // every recomputed Gaussian (old and new rect) with the frame's token; every other tile is skipped here.
    const uint32_t *dirty = nullptr;
    uint32_t td_token = 0u;
    if constexpr (SUPER) {
        if (tile_dirty != nullptr && hdr->td_reuse != 0u) {
            dirty = tile_dirty;
            td_token = hdr->td_token;
        }
    }
    if constexpr (SUPER) {
        // the FIRST coop_blocks workgroups each take ONE of last frame's costliest quadrants (render_coop_quadrant).  First,
        // not last: the main grid fills the chip to a few workgroups short of what is resident at once, and a workgroup the
        // dispatcher cannot place waits for another to retire -- behind the grid a few of the longest chains of the frame
        // started 35 us late (sensor view 6.7 -> 6.1 k frames/s); in front, what may wait is the main grid's tail, which the
        // deal fills with the cheapest quadrants
        if (coop_list != nullptr && (int)wgx < coop_blocks) {
            __shared__ CoopFlags s_coop;
            if (threadIdx.x < sizeof(CoopFlags) / sizeof(uint32_t)) reinterpret_cast<uint32_t *>(&s_coop)[threadIdx.x] = 0u;
            __syncthreads();
            const uint32_t q = coop_list[wgx];
            if (q >= 4u * (uint32_t)num_tiles) return;  // (no quadrant for this workgroup: 0xFFFFFFFF)
            if (dirty != nullptr && dirty[q >> 2] != td_token) {  // (its tile keeps the previous frame's pixels)
                if (threadIdx.x == 0) hdr->td_skipped = 1u;
                return;
            }
            const bool timed_out = render_coop_quadrant<true>(s_list, &s_coop, q, ranges, point_list, splat, W, H, gx, bg,
                                                              out_color, out_invdepth, rgb8, quad_work, final_T);
            // (cannot happen by the counters' construction; if it ever does the quadrant is truncated, and the frame says so)
            if (timed_out && lane == 0) gsr_note_coop_timeout(hdr);
            return;
        }
    }
    float4(*list)[6] = s_list[wave];
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t num_tickets = 4u * (uint32_t)num_tiles;
    // static LPT schedule: units sorted longest-first are dealt round-robin over the resident waves.  Measured
    // alternatives: one device-wide ticket counter (same-address device-scope atomics resolve memory-side, ~10 ns
    // apiece back to back: cost more than the compositing itself) and 8 per-XCD counters pulled one unit ahead with
    // 2-4 workgroups per CU resident (98-111 us against 77 us for everything resident: a wave is a latency chain, so
    // the more of them are in flight the better, and with 4 x tiles <= resident waves nothing is left to balance).
    // Giving each XCD a contiguous part of the image (row chunks, or a 4 x 2 split) instead of every 8th tile did not
    // pay either (80-82 us): what the L2s gain in locality the XCDs lose in balance.
    // (snake order: pass 0 deals the longest units to waves 0..S-1, pass 1 deals the next ones to waves S-1..0, so a
    // wave that started with a long unit continues with a short one)
    // Split quadrants.  The kernel lasts as long as its longest quadrant (a wave is one in-order instruction stream),
    // and the costs spread 219 (mean) ... 369 survivors at config 2.  The quadrants that were costliest in the previous
    // frame on this state (tile_starts_kernel, workgroup 2) are therefore cut in two 8 x 4 halves: the quadrant's own
    // wave keeps rows 0-3 (its lanes 0-31), a wave of the EXTRA workgroups behind the main grid takes rows 4-7, each
    // with its own, tighter cull rectangle.  Pixels are independent, so the image state does not change by a bit.
    const uint32_t bx = wgx - (uint32_t)coop_blocks;  // (coop_blocks is a multiple of 8: workgroup bx on XCD bx mod 8)
    const bool extra = (int)bx >= main_blocks;
    uint32_t extra_q = 0;
    if (extra) {
        const uint32_t e = (bx - (uint32_t)main_blocks) * (uint32_t)(GSR_BLOCK / GSR_WAVE) + (uint32_t)wave;
        if (split_count == nullptr || e >= *split_count) return;
        extra_q = split_list[e];
    }
    const uint32_t ticket_stride = (uint32_t)main_blocks * (uint32_t)(GSR_BLOCK / GSR_WAVE);
    const uint32_t wave_global = bx * (uint32_t)(GSR_BLOCK / GSR_WAVE) + (uint32_t)wave;
    for (uint32_t pass = 0, ticket = extra ? 0u : wave_global; ticket < num_tickets;
         pass++, ticket = extra ? num_tickets
                                : pass * ticket_stride + ((pass & 1u) ? ticket_stride - 1u - wave_global : wave_global)) {
        uint32_t unit = ticket >> 2;
        if (!extra && tile_order != nullptr && quad_order == nullptr && main_blocks >= num_tiles) {
            // Every tile resident at once: workgroups b, b + #CUs, b + 2 #CUs ... share a CU (observed placement on
            // MI355X: s_getreg HW_ID of every workgroup), so position p of the cost-sorted order goes to workgroup
            // p in even groups of #CUs and to the mirrored workgroup in odd ones: every CU gets one tile of each cost
            // class, and the CU that got the costliest of one class gets the cheapest of the next.
            const uint32_t group = unit / (uint32_t)num_cus, idx = unit - group * (uint32_t)num_cus;
            const uint32_t size = min((uint32_t)num_cus, (uint32_t)num_tiles - group * (uint32_t)num_cus);
            unit = group * (uint32_t)num_cus + ((group & 1u) ? size - 1u - idx : idx);
        }
        int tile = extra ? (int)(extra_q >> 2) : (tile_order ? (int)tile_order[unit] : (int)unit);
        int quad = extra ? (int)(extra_q & 3u) : (int)(ticket & 3u);
        if (!extra && quad_order != nullptr && main_blocks >= num_tiles) {
            // Everything resident: workgroup `unit` (= wgx) takes the four quadrants the deal of
            // gsr_quad_order_block assigned to it -- four of (nearly) equal cost in the previous frame, from tiles of
            // this workgroup's XCD.  The four waves of a workgroup go to the four SIMDs of its CU, so every SIMD of the CU
            // carries the same load; dealing tiles (four quadrants of unequal cost) left the SIMD loads a sum of five
            // random quadrant costs each -- +-11 %, and the most loaded of 1024 SIMDs sets the kernel time.
            const uint32_t q = quad_order[4u * unit + (ticket & 3u)];
            if (SUPER && (q & 0x80000000u) != 0u) continue;  // (a cooperative workgroup has it: the deal marks the entry)
            tile = (int)(q >> 2);
            quad = (int)(q & 3u);
        }
        if (dirty != nullptr && dirty[tile] != td_token) {  // (tile reuse: the previous frame's pixels stand)
            if (lane == 0) hdr->td_skipped = 1u;
            continue;
        }
        // half: 0 = whole quadrant, 1 = its rows 0-3 (the other half runs elsewhere), 2 = its rows 4-7
        const int half = extra ? 2 : ((split_flag != nullptr && split_flag[4 * tile + quad] != 0u) ? 1 : 0);
        const int qx0 = (tile % gx) * GSR_TILE + ((quad & 1) << 3);
        const int qy0 = (tile / gx) * GSR_TILE + ((quad >> 1) << 3) + (half == 2 ? 4 : 0);
        const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
        const bool inside = px < W && py < H && (half == 0 || lane < 32);
        const float pfx = (float)px, pfy = (float)py;
        const float qxf = (float)qx0, qyf = (float)qy0;
        const float yext = half == 0 ? 7.0f : 3.0f;
        const uint32_t tx = (uint32_t)(tile % gx), ty = (uint32_t)(tile / gx);
        constexpr int kSX = GSR_SUPER_SX, kSY = GSR_SUPER_SY;
        const uint2 range = ranges[SUPER ? (int)((ty >> kSY) * (uint32_t)((gx + (1 << kSX) - 1) >> kSX) + (tx >> kSX)) : tile];
        const int n_inst = (int)(range.y - range.x);
        const uint32_t *src = point_list + range.x;

#if GSR_STREAM_STAMPS
        const uint64_t stamp0 = __builtin_readcyclecounter();
#endif
        const v2f pf2x = {pfx, pfx}, pf2y = {pfy, pfy};
        // A pixel that is finished (saturated, or outside the image) carries its transmittance NEGATED: T * (1 - a) is
        // then negative, so the saturation test below fires for it again by itself and no separate done flag has to
        // sit on the dependent chain (per survivor: multiply -> compare -> select).
        float T = inside ? 1.0f : -1.0f;
        v2f acc_rg = {0.f, 0.f}, acc_bd = {0.f, 0.f};  // (red, green), (blue, inverse depth)
        uint32_t last_contributor = 0;

        // global -> register pipeline: records of round rd in (f0,f1,f2), Gaussian indices of round rd+1 in g_next
        float4 f0[kStreamLanesItems], f1[kStreamLanesItems], f2[kStreamLanesItems];
        uint32_t g_next[kStreamLanesItems];
#pragma unroll
        for (int s = 0; s < kStreamLanesItems; s++) {
            const int p = s * GSR_WAVE + lane;
            f0[s] = zero4; f1[s] = zero4; f2[s] = zero4;
            g_next[s] = 0;
            if (p < n_inst) {
                const float4 *rec = splat + 3 * (size_t)src[p];
                f0[s] = rec[0]; f1[s] = rec[1]; f2[s] = rec[2];
            }
            if (p + kStreamRound < n_inst) g_next[s] = src[p + kStreamRound];
        }
        const int rounds = (n_inst + kStreamRound - 1) / kStreamRound;
        uint32_t work = 0;  // what this unit cost, in survivor evaluations (+ 5 per candidate round): next frame's order key
        for (int rd = 0; rd < rounds; rd++) {
            if (__builtin_amdgcn_ballot_w64(T > 0.0f) == 0ull) break;
            // ---- cull + compact this round's candidates into the private list
            int n_surv = 0;
            uint64_t unsafe = 0ull;  // survivors whose conic is not comfortably positive definite
#pragma unroll
            for (int s = 0; s < kStreamLanesItems; s++) {
                const int p = rd * kStreamRound + s * GSR_WAVE + lane;
                bool keep = p < n_inst;
                if (SUPER) {  // getRect of upstream: rect_min <= tile < rect_max on both axes
                    const uint32_t rb = __float_as_uint(f2[s].w);
                    keep = keep && tx - (rb & 255u) < ((rb >> 16) & 255u) - (rb & 255u) &&
                           ty - ((rb >> 8) & 255u) < (rb >> 24) - ((rb >> 8) & 255u);
                }
                if (SUPER) {  // tau and the safe-conic flag come with the record (preprocess.hip)
                    keep = keep && quadrant_may_hit<true>(f0[s].x, f0[s].y, f1[s], qxf, qyf, yext, f0[s].z);
                    unsafe |= __builtin_amdgcn_ballot_w64(keep && (__float_as_uint(f0[s].z) & 1u) == 0u);
                } else {
                    keep = keep && quadrant_may_hit(f0[s].x, f0[s].y, f1[s], qxf, qyf, yext);
                    unsafe |= __builtin_amdgcn_ballot_w64(keep && !stream_conic_is_safe(f1[s]));
                }
                const uint64_t mask = __builtin_amdgcn_ballot_w64(keep);
                const int rank = n_surv + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                if (keep)
                    stream_list_put(list, rank, f0[s], f1[s], make_float4(f2[s].x, f2[s].y, f2[s].z, f0[s].w),
                                    __uint_as_float((uint32_t)p + 1u));
                n_surv += (int)__builtin_popcountll(mask);
            }
            work += 8u;  // cull + compaction of a round of 64 candidates, in units of one replayed survivor
            if (lane < kBatch)  // alpha = 0 padding behind the last survivor
                stream_list_put(list, n_surv + lane, zero4, zero4, zero4, 0.0f);
            // ---- next round's gathers go out before the replay so that they fly under it
#pragma unroll
            for (int s = 0; s < kStreamLanesItems; s++) {
                const int p = (rd + 1) * kStreamRound + s * GSR_WAVE + lane;
                f0[s] = zero4; f1[s] = zero4; f2[s] = zero4;
                if (p < n_inst) {
                    const float4 *rec = splat + 3 * (size_t)g_next[s];
                    f0[s] = rec[0]; f1[s] = rec[1]; f2[s] = rec[2];
                }
                if (p + kStreamRound < n_inst) g_next[s] = src[p + kStreamRound];
            }
            __builtin_amdgcn_wave_barrier();  // (scheduling fence: list writes above, list reads below; same wave)
            // ---- replay the survivors, 4 at a time
            // Cycle counters per wave (config-2 frame, 5 waves per SIMD): waiting for the gathers 1 %, cull + compaction
            // 13 %, this loop 80 % at ~270 clk per survivor -- and a wave alone on its SIMD is hardly faster, so the loop
            // is written for the fewest instructions on the shortest chain: no branch inside a batch (per-instance
            // "does any lane hit" skips cost more in compare -> SGPR -> branch round trips than the VALU they saved:
            // after the cull 77 % of the evaluated pixel-instances hit anyway), survivor pairs through the packed
            // pipe, the finished state carried in T's sign.  (Measured alternatives: predicates as 0/1 floats instead
            // of wave masks: +6 us, costs an occupancy step; batches of 2 / 8.)
            // (Evaluating batch i + 1 ahead of the blend of batch i -- one basic block, ping-pong registers -- was
            // measured: +17 % replay time; the loop is bound by instruction count, not by exposed latency.)
#if GSR_STREAM_MASKED_BLEND && GSR_STREAM_ALIVE_BLEND
            if (SUPER && unsafe == 0ull) {
                // (the finished state moves from T's sign into a wave mask for the round and back: the rounds with an
                // unsafe conic, and the exit test at the top of the round, read the sign)
                uint64_t alive = __builtin_amdgcn_ballot_w64(T > 0.0f);
                v2f Tp = {T, 0.0f};
                int i = 0;
                // (one scalar compare per batch decides the loop: the blend's last instructions drop `limit` to 0 once no
                // lane is alive -- `i < n_surv && alive != 0` went through eight scalar instructions per batch)
                int limit = n_surv;
                const uint64_t full = __builtin_amdgcn_read_exec();
                if (n_surv > 0) do {
                    const StreamBatch b = stream_eval<false, false>(list, i, pf2x, pf2y);
                    stream_blend_alive(b, Tp, acc_rg, acc_bd, alive, limit, full);
                    work += (uint32_t)kBatch;
                    i += kBatch;
                } while (i < limit);
                asm volatile("v_cndmask_b32_e64 %0, -|%1|, %1, %2" : "=v"(T) : "v"(Tp.x), "s"(alive));
            } else
#endif
            if (unsafe == 0ull) {
                // (one exit test, at the bottom: with a second exit at the top the accumulators were copied through
                // six v_mov per iteration)
                int i = 0;
                if (n_surv > 0) do {
                    const StreamBatch b = stream_eval<false, !SUPER>(list, i, pf2x, pf2y);
#if GSR_STREAM_MASKED_BLEND
                    if (SUPER)
                        stream_blend_masked(b, T, acc_rg, acc_bd);
                    else
#endif
                    stream_blend<!SUPER>(b, T, acc_rg, acc_bd, last_contributor);
                    work += (uint32_t)kBatch;  // survivors actually replayed (a saturated quadrant stops early)
                    i += kBatch;
                } while (i < n_surv && __builtin_amdgcn_ballot_w64(T > 0.0f) != 0ull);
            } else {
                int i = 0;
                if (n_surv > 0) do {
                    const StreamBatch b = stream_eval<true, !SUPER>(list, i, pf2x, pf2y);
                    stream_blend<!SUPER>(b, T, acc_rg, acc_bd, last_contributor);
                    work += (uint32_t)kBatch;
                    i += kBatch;
                } while (i < n_surv && __builtin_amdgcn_ballot_w64(T > 0.0f) != 0ull);
            }
            __builtin_amdgcn_wave_barrier();  // the next round overwrites the list
        }
        if (quad_work != nullptr && lane == 0) (half == 2 ? quad_work_b : quad_work)[4 * tile + quad] = work;
#if GSR_STREAM_STAMPS
        // tuning build (tools/stream_stamps.py, tools/stamps_report.py): per quadrant, shader-clock stamps of its start
        // and end (the clock is per CU), where it ran (HW_ID | XCC_ID << 28) and what it cost -- left in final_T, which
        // inference frames do not write
        if (SUPER && lane == 0) {
            const uint64_t stamp1 = __builtin_readcyclecounter();
            const uint32_t hw = (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0x0fffffffu;
            const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20);
            reinterpret_cast<uint4 *>(final_T)[4 * tile + quad] =
                make_uint4((uint32_t)stamp0, (uint32_t)stamp1, hw | (xcc << 28), work);
        }
#endif
        if (inside) {
            const size_t pid = (size_t)py * W + px;
            const size_t plane = (size_t)H * W;
            T = fabsf(T);
            if (!SUPER) {
                final_T[pid] = T;
                n_contrib[pid] = last_contributor;
            }
            const float r = fma_(T, bg0, acc_rg.x), g = fma_(T, bg1, acc_rg.y), b = fma_(T, bg2, acc_bd.x);
            if (out_color != nullptr) {  // (NULL: an inference frame whose caller keeps the uint8 frame only, include/gsr.h)
                out_color[pid] = r;
                out_color[plane + pid] = g;
                out_color[2 * plane + pid] = b;
                out_invdepth[pid] = acc_bd.y;
            }
            if (rgb8) {  // GSWorld's frame conversion, same arithmetic as pack_rgb8_kernel
                uint8_t *o = rgb8 + 3 * pid;
                o[0] = (uint8_t)fminf(fmaxf(r * 255.0f, 0.0f), 255.0f);
                o[1] = (uint8_t)fminf(fmaxf(g * 255.0f, 0.0f), 255.0f);
                o[2] = (uint8_t)fminf(fmaxf(b * 255.0f, 0.0f), 255.0f);
            }
        }
    }
}

// grid = (workgroups of one frame, rounded up to a multiple of 8 when frames share a launch, frames): blockIdx.y picks
// the frame's argument block (gsr_internal.h GsrBatch).  Workgroups are dispatched x-fastest and round-robin over the
// XCDs, so with a row length that is a multiple of 8 workgroup b of EVERY frame runs on XCD b mod 8 -- what the quadrant
// deal (gsr_quad_order_block) assumes when it keeps a tile's four quadrants on one L2.  With more workgroups than the
// chip holds the dispatcher hands a CU its next workgroup when one retires: frames later in the launch fill the tail the
// costliest quadrants of the earlier ones leave.
struct RenderStreamArgs {
    const uint2 *ranges;
    const uint32_t *point_list;
    const float4 *splat;
    int W, H, gx, num_tiles;
    const uint32_t *tile_order;
    const float *bg;
    float *out_color, *out_invdepth, *final_T;
    uint32_t *n_contrib;
    uint8_t *rgb8;
    uint32_t *quad_work;
    int num_cus, main_blocks, total_blocks;
    const uint32_t *split_flag, *split_list, *split_count;
    uint32_t *quad_work_b;
    const uint32_t *quad_order;
    const uint32_t *coop_list;  // cooperative quadrants (null: none): the quadrant of each of the first coop_blocks workgroups
    int coop_blocks;
    GsrHeader *hdr;             // (a cooperative quadrant whose hand-off timed out is counted there)
    int frames;                 // frames of the launch (the interleaved grid: render_stream_kernel)
    const uint32_t *tile_dirty; // tile reuse: tiles marked with hdr->td_token are composited, the others keep the previous
                                // frame's pixels when hdr->td_reuse says so (nullptr: every tile, every frame)
};
template <bool SUPER>
__global__ __launch_bounds__(GSR_BLOCK) __attribute__((amdgpu_waves_per_eu(5))) void render_stream_kernel(
    const GsrBatch<RenderStreamArgs> bt) {
    // (grid = (row, frames): the frames' rows one behind the other.  A one-row grid is the A/B variant that interleaves the
    //  frames eight workgroups at a time, so that every frame's costliest quadrants start at once -- measured slower, see
    //  GSR_RENDER_INTERLEAVE)
    const uint32_t frames = gridDim.y == 1u ? (uint32_t)bt.f[0].frames : 1u;
    const uint32_t grp = blockIdx.x >> 3;
    const uint32_t frame = gridDim.y == 1u ? grp % frames : blockIdx.y;
    const uint32_t wgx = gridDim.y == 1u ? (grp / frames) * 8u + (blockIdx.x & 7u) : blockIdx.x;
    const RenderStreamArgs &a = bt.f[frame];
    if ((int)wgx >= a.total_blocks) return;  // (padding of the row to a multiple of 8)
    render_stream_body<SUPER>(a.ranges, a.point_list, a.splat, a.W, a.H, a.gx, a.num_tiles, a.tile_order, a.bg,
                              a.out_color, a.out_invdepth, a.final_T, a.n_contrib, a.rgb8, a.quad_work, a.num_cus,
                              a.main_blocks, a.split_flag, a.split_list, a.split_count, a.quad_work_b, a.quad_order,
                              a.coop_list, a.coop_blocks, a.hdr, wgx, a.tile_dirty);
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

static int render_num_cus() {
    static int num_cus = 0;
    if (num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
            prop.multiProcessorCount > 0)
            num_cus = prop.multiProcessorCount;
        else
            num_cus = 256;
    }
    return num_cus;
}

// (csrc/Makefile compiles this file with the hand-scheduled blend -- inline asm that pins v[92:95] under
//  amdgpu_waves_per_eu(5) -- and, should a compiler ever refuse it, again with -DGSR_STREAM_ALIVE_BLEND=0: the plain
//  masked blend, bit-identical frames; gsr_version() tells which one the library holds)
const char *gsr_render_build_flags() {
#if GSR_STREAM_MASKED_BLEND && GSR_STREAM_ALIVE_BLEND
    return "alive_blend=1";
#else
    return "alive_blend=0";
#endif
}

int gsr_render_cus_per_xcd() { return render_num_cus() / GSR_XCDS > 0 ? render_num_cus() / GSR_XCDS : 1; }

// The longest-first tile order only matters when workgroups take more than one tile: with every tile resident at
// once (1200 tiles on 256 CUs x 6) the deal is the identity and the binning stage need not build the order.
// The default compositor always takes an order: with every unit resident at once (tiles <= CUs x workgroups per CU) it
// decides which tiles share a CU; beyond that it is the longest-first queue order.
// Workgroups appended to the compositing grid for the second halves of split quadrants: whatever the chip can still
// hold next to the main grid (all tiles resident: tiles < CUs x workgroups per CU), none otherwise.  The split list is
// built by the counting placements (tile_starts_kernel); GsrSettings.render_split = 1 turns the splitting off.
int gsr_render_split_blocks(const GsrSettings &st, int num_tiles) {
    const RenderChoice c = render_choice(st);
    if (c.variant != 4 || st.render_split != 1) return 0;  // (off by default: measured slower)
    const int resident = render_num_cus() * c.blocks_per_cu;
    return num_tiles < resident ? resident - num_tiles : 0;
}

// true when the default compositor keeps every tile resident at once AND takes its quadrants in the order of their
// previous cost (quad_order, built by tile_starts_kernel instead of the tile-level order)
bool gsr_render_uses_quad_order(const GsrSettings &st, int num_tiles) {
    const RenderChoice c = render_choice(st);
    return c.variant == 4 && num_tiles <= 2048 && num_tiles <= render_num_cus() * c.blocks_per_cu;
}

bool gsr_render_wants_tile_order(const GsrSettings &st, int num_tiles) {
    const RenderChoice c = render_choice(st);
    return c.variant == 4 || (c.variant >= 2 && num_tiles > render_num_cus() * c.blocks_per_cu);
}

// Workgroups in front of the compositor's main grid that take one of last frame's costliest quadrants each
// (render_coop_quadrant; the quadrants are picked by the deal of ss_quad_order_1024, depthsort.hip).  GsrSettings.
// render_split: 0 = inference frames on the default path, 2 = the same, 3 = never (1 = round 3's split halves).  Measured
// per frames per launch (profiles/round5/ab_coop_by_frames_per_launch.txt): dense view +12.5 / +7.8 / +6.2 / +4.9 / +2.4 %
// at 1 / 2 / 3 / 4 / 8 frames per launch from one stream, the same within the noise with three streams in flight (other
// launches fill the tail the long quadrants leave); sensor view and closed loop (1 / 2 / 4 environments) within +-0.6 %
// everywhere.  GSR_COOP_MAX_FRAMES (= all) is what is left of a first version that kept to launches of one or two frames.
int gsr_render_coop_blocks(const GsrSettings &st, int num_tiles, int frames) {
    const RenderChoice c = render_choice(st);
    if (c.variant != 4 || st.render_split == 1 || st.render_split == 3) return 0;
    if (st.render_split != 2 && frames > GSR_COOP_MAX_FRAMES) return 0;
    if (!gsr_render_uses_quad_order(st, num_tiles)) return 0;
    const int spare = render_num_cus() * 5 - num_tiles;  // (five workgroups per CU are resident: amdgpu_waves_per_eu(5))
    const int blocks = (spare < GSR_COOP_MAX_BLOCKS ? spare : GSR_COOP_MAX_BLOCKS) / GSR_XCDS * GSR_XCDS;
    return blocks > 0 ? blocks : 0;
}

int gsr_launch_render(int B, const GsrFrame *fr, bool order_ready, bool split_ready, bool super_tiles, int coop_blocks,
                      hipStream_t stream) {
    const GsrSettings &st = *fr[0].st;
    const int W = st.image_width, H = st.image_height;
    const int gx = gsr_div_up(W, GSR_TILE), gy = gsr_div_up(H, GSR_TILE);
    // the default kernel writes the uint8 frame itself; the A/B variants get a separate conversion pass
    const RenderChoice rc = render_choice(st);
    if (B > 1 && rc.variant != 4) {
        gsr_set_error("gsr_launch_render: only the default compositor takes several frames per launch");
        return GSR_E_INVALID;
    }
    const GeomState &g = fr[0].g;
    const ImageState &img = fr[0].img;
    const uint32_t *point_list = fr[0].b.gidx[0];
    const float *background = fr[0].in->background;
    float *out_color = fr[0].out->out_color, *out_invdepth = fr[0].out->out_invdepth;
    uint8_t *out_rgb8 = fr[0].out->out_rgb8;
    const bool pack_after = out_rgb8 != nullptr && rc.variant != 4;
    if (rc.variant >= 2) {
        const int T = gx * gy;
        const bool ordered = gsr_render_wants_tile_order(st, T);
        const uint32_t *order = ordered ? img.tile_order : nullptr;
        if (ordered && !order_ready)
            for (int k = 0; k < B; k++)
                hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(GSR_BLOCK), 0, stream, fr[k].img.ranges, T,
                                   fr[k].img.tile_order);
        const int blocks = min(T, render_num_cus() * rc.blocks_per_cu);
        if (rc.variant == 4) {
            // (the split list is built by tile_starts_kernel: counting placements, grids up to 2048 tiles)
            const int coop = super_tiles && split_ready ? coop_blocks : 0;
            const int extra = coop > 0 ? coop : (split_ready && T <= 2048 ? gsr_render_split_blocks(st, T) : 0);
            const bool use_qorder = split_ready && gsr_render_uses_quad_order(st, T);
            GsrBatch<RenderStreamArgs> bt{};  // (entries beyond B stay zero: nothing uninitialised travels in the kernarg)
            for (int k = 0; k < B; k++) {
                const ImageState &im = fr[k].img;
                RenderStreamArgs &a = bt.f[k];
                a.ranges = im.ranges;
                a.point_list = fr[k].b.gidx[0];
                a.splat = fr[k].g.splat;
                a.W = W; a.H = H; a.gx = gx; a.num_tiles = T;
                a.tile_order = ordered ? im.tile_order : (const uint32_t *)nullptr;
                a.bg = fr[k].in->background;
                a.out_color = fr[k].out->out_color;
                a.out_invdepth = fr[k].out->out_invdepth;
                a.final_T = im.final_T;
                a.n_contrib = im.n_contrib;
                a.rgb8 = fr[k].out->out_rgb8;
                a.quad_work = im.quad_work;
                a.num_cus = render_num_cus();
                a.main_blocks = blocks;
                a.total_blocks = blocks + extra;
                a.split_flag = extra > 0 && coop == 0 ? im.split_flag : (const uint32_t *)nullptr;
                a.coop_list = coop > 0 ? im.split_list : (const uint32_t *)nullptr;  // (the split list, reused)
                a.coop_blocks = coop;
                a.hdr = fr[k].g.hdr;
                a.frames = B;
                a.split_list = im.split_list;
                a.split_count = im.split_count;
                a.quad_work_b = im.quad_work_b;
                a.quad_order = use_qorder ? im.quad_order : (const uint32_t *)nullptr;
                a.tile_dirty = (fr[k].pc && fr[k].td) ? im.tile_dirty : (const uint32_t *)nullptr;
            }
            const int row = B > 1 ? (blocks + extra + GSR_XCDS - 1) / GSR_XCDS * GSR_XCDS : blocks + extra;
#ifndef GSR_RENDER_INTERLEAVE
#define GSR_RENDER_INTERLEAVE 0  // MEASURED AND NOT KEPT (1 = on): the two cameras of a closed-loop step 80.7 -> 93.1 us,
                                 // four environments 12.6 -> 11.2 k frames/s, the headline 14.7 -> 14.5 k -- a frame's
                                 // workgroups next to each other in the grid share its lists in the L2s
#endif
            const dim3 grid = (GSR_RENDER_INTERLEAVE && B > 1) ? dim3(row * B, 1) : dim3(row, B > 1 ? B : 1);
            if (B == 1) bt.f[0].frames = 1;
            if (super_tiles)
                hipLaunchKernelGGL(render_stream_kernel<true>, grid, dim3(GSR_BLOCK), 0, stream, bt);
            else
                hipLaunchKernelGGL(render_stream_kernel<false>, grid, dim3(GSR_BLOCK), 0, stream, bt);
        }
        else if (rc.variant == 3)
            hipLaunchKernelGGL(render_queue_kernel<true>, dim3(blocks), dim3(GSR_BLOCK), 0, stream, img.ranges,
                               point_list, g.splat, W, H, gx, T, order, background, out_color, out_invdepth,
                               img.final_T, img.n_contrib);
        else
            hipLaunchKernelGGL(render_queue_kernel<false>, dim3(blocks), dim3(GSR_BLOCK), 0, stream, img.ranges,
                               point_list, g.splat, W, H, gx, T, order, background, out_color, out_invdepth,
                               img.final_T, img.n_contrib);
    } else {
        hipLaunchKernelGGL(render_kernel, dim3(gx * gy), dim3(GSR_BLOCK), 0, stream, img.ranges, point_list, g.splat,
                           W, H, gx, background, out_color, out_invdepth, img.final_T, img.n_contrib);
    }
    if (pack_after) {
        const int n_pix = W * H;
        hipLaunchKernelGGL(pack_rgb8_kernel, dim3(gsr_div_up(gsr_div_up(n_pix, 4), GSR_BLOCK)), dim3(GSR_BLOCK), 0,
                           stream, out_color, n_pix, out_rgb8);
    }
    return GSR_OK;
}
