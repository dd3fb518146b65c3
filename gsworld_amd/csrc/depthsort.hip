// depthsort.hip -- depth order of the visible Gaussians in THREE launches (upstream: the depth half of
// cub::DeviceRadixSort::SortPairs over the 64-bit (tile | depth) keys, rasterizer_impl.cu).
//
// What is sorted: V (depth bits, Gaussian index) records, ascending depth bits, ties by ascending index -- the order the
// reference's stable key sort leaves inside every tile.  At config 2 V = 175 k: 1.4 MB, a problem of LATENCY, not
// bandwidth.  The 3-pass LSD radix sort of round 1 (sort.hip: compaction + 3 x (histogram, row scan, scatter) = 10
// launches of ~86 workgroups each) spent 95 us on it, 1 % of the HBM roofline.  Here a sample sort (48 us; 25 + 16
// + 12 when it samples, 19 + 17 + 12 while the camera rests):
//
//   ss_compact   index-ordered compaction of the visible (key, index) records (per-workgroup offsets by redundant
//                sums of the per-block counts preprocess left), AND, in the same pass: every workgroup sorts the same
//                <= 4096 sampled keys in LDS, cuts them into B - 1 splitters, classifies its records by binary search
//                and leaves its bucket histogram
//   ss_partition every workgroup sums the histogram rows before its own (no scan kernel), then moves ITS segment into
//                the buckets, stably (wave64 match-any ranks, as the radix scatter)
//   ss_buckets   one workgroup per bucket: stable LSD radix sort in LDS on the bits that actually differ inside the
//                bucket (typically 2 passes of 8 bits), result = the bare Gaussian indices in `order`
//
// Stability (index order on equal keys) is kept end to end: compaction in index order, stable partition, stable LSD
// passes.  Samples are uniform over the VISIBLE Gaussians: sample s is the first visible key of the preprocess block
// (256 Gaussians) that holds visible Gaussian s V / S -- uniform over the blocks would starve the dense part of an
// index-coherent model.  Splitters carry the top 24 key bits only, so records with equal depth never straddle a bucket
// boundary by accident of the sample order.  Bucket count B = 256..2048 follows V (read on the device) so that a
// bucket averages <= 512 records.  ss_buckets leaves the exact quantiles of the frame in the state; the next frame on
// that state only validates them against its samples and skips the sample sort (any splitters give the same order);
// under a bit-identical view matrix (a fixed sensor camera) over a scene that has stood still for two frames (ss_trust)
// it takes them without drawing samples at all, and a bucket above its share makes the following frames sample again.
// Compaction workgroups share the blocks by COST (records + kBlockCost per block), not by count: see ss_compact_kernel.
// A bucket that does not fit the LDS (> kBucketCap records: a sampled table's unlucky bucket, a scene that jumped under
// kept splitters, depth ties) is cut once more by the same workgroup and sorted in LDS pieces; only a piece of equal
// keys goes through a bitonic network over the (key << 32 | index) composites in global memory (ss_buckets_kernel).
#include "gsr_internal.h"
#include <type_traits>

namespace {

#ifdef GSR_SS_TIMING
#define SS_STAMP(buf, slot) do { __syncthreads(); if (blockIdx.x == (buf##_wg) && threadIdx.x == 0) (buf)[slot] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SS_STAMP(buf, slot) do { } while (0)
#endif

constexpr int kT = GSR_BLOCK;        // 256 threads, 4 waves
#ifndef GSR_SS_TRUST_MIN
#define GSR_SS_TRUST_MIN 2     // A/B: 0 = a fixed camera takes the kept splitters whatever the last frames looked like
#endif
#ifndef GSR_SS_IGNORE_BAD
#define GSR_SS_IGNORE_BAD 0
#endif
#ifndef GSR_SS_SPB
#define GSR_SS_SPB 4  // (2 until round 6: frames of 1 024 buckets -- 262 k to 524 k visible Gaussians -- drew 2 048 samples;
#endif                //  with 4 096 the training step's ss_buckets is 4 us shorter, the closed loop the same)
constexpr int kSamplesPerBucket = GSR_SS_SPB;
constexpr int kMinSamples = 2048;
constexpr int kMaxSamples = 4096;
constexpr uint32_t kSplitMagic = 0x53504c54u;  // 'SPLT'
// placement cost of one Gaussian = its instances + this many (every tile row streams and tests every rank of its share)
constexpr uint32_t kRankCost = 40u;
// compaction cost of one preprocess block (256 keys fetched and tested), in records
constexpr uint32_t kBlockCost = 8u;
#ifndef GSR_SS_KEY_MASK
#define GSR_SS_KEY_MASK 0xFFFFFFFFu
#endif
// key bits the splitters see.  Equal keys always classify alike, whatever the mask; a coarser key only makes the buckets
// coarser: with the top 24 bits (rounds 1-2) a flat table seen from straight above had ~200 distinct splitter values for
// 600 k Gaussians, every frame ended "unbalanced" and the compaction kept drawing new samples (dense view: 36 -> 29 us
// with all 32 bits; config 2 unchanged)
constexpr uint32_t kKeyMask = GSR_SS_KEY_MASK;
constexpr uint32_t kNoKey = 0xFFFFFFFFu;  // (never a depth key: those are positive floats)
// Records per bucket sorted in LDS: 4 arrays of this many words + cursors = 36 KiB, four workgroups per CU.  3584 (60 KiB,
// two per CU) was the value while a bucket beyond it went through a global-memory bitonic network; now that such a bucket
// is cut once more and sorted in LDS pieces, a smaller cap with twice the residency wins wherever the sort samples
// (moving camera 8.4 -> 8.9 k frames/s, dense view +3 %, train step -0.01 ms; static headline unchanged; 1536: the same).
#ifndef GSR_SS_BUCKET_CAP
#define GSR_SS_BUCKET_CAP 2048
#endif
constexpr int kBucketCap = GSR_SS_BUCKET_CAP;

__device__ __forceinline__ int ss_num_buckets(uint32_t V, int bmax, uint32_t per) {
    int B = 256;
    while (B < bmax && (uint32_t)B * per < V) B <<= 1;
    return B;
}
__device__ __forceinline__ int ss_log2(int B) { return 31 - __builtin_clz((unsigned)B); }
// minimum / maximum over the 64 lanes, in every lane (DPP row shifts and row broadcasts: six adds' worth of VALU instead of
// six ds_bpermute round trips per value)
__device__ __forceinline__ uint32_t ss_wave_min(uint32_t v) {
    int x = (int)v;
#define SS_STEP(ctrl, rmask) x = (int)min((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(x, x, ctrl, rmask, 0xf, false))
    SS_STEP(0x111, 0xf); SS_STEP(0x112, 0xf); SS_STEP(0x114, 0xf); SS_STEP(0x118, 0xf); SS_STEP(0x142, 0xa); SS_STEP(0x143, 0xc);
#undef SS_STEP
    return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}
__device__ __forceinline__ uint32_t ss_wave_max(uint32_t v) {
    int x = (int)v;
#define SS_STEP(ctrl, rmask) x = (int)max((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(x, x, ctrl, rmask, 0xf, false))
    SS_STEP(0x111, 0xf); SS_STEP(0x112, 0xf); SS_STEP(0x114, 0xf); SS_STEP(0x118, 0xf); SS_STEP(0x142, 0xa); SS_STEP(0x143, 0xc);
#undef SS_STEP
    return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}
// a bucket above this many records makes its frame "unbalanced" (ss_bad): 1.25 shares under the view the table was built
// under, 2 under another one
__device__ __forceinline__ uint32_t ss_share_limit(uint32_t V, int B, bool moved) {
    const uint32_t share = V / (uint32_t)B;
    return (moved ? 2u * share : share + share / 4u) + 64u;
}
#ifndef GSR_SS_DIAG
#define GSR_SS_DIAG 0
#endif
#ifndef GSR_SS_PROBE_EVERY
#define GSR_SS_PROBE_EVERY 0  // MEASURED AND NOT KEPT (4 = after a failed check, check again only every fourth frame):
                              // closed loop +0.4 % (8.74 against 8.70 k frames/s), but the training step 0.503 -> 0.52 ms --
                              // its views change every step and a kept table that passes the check (exact quantiles of a
                              // nearby view) balances 1 024 buckets far better than one drawn from two samples per bucket:
                              // ss_buckets 23.7 -> 38.5 us
#endif
#ifndef GSR_SS_NEAR
#define GSR_SS_NEAR 1  // (A/B: 0 = only a bit-identical view matrix takes the kept table unchecked, as until round 6)
#endif
#ifndef GSR_SS_NEAR_TOL
#define GSR_SS_NEAR_TOL 0.015625f
#endif
// "a little": every element of the view matrix within this of the table's -- 1 / 64: under a degree, 1.6 cm of a metric scene
constexpr float kNearTol = GSR_SS_NEAR_TOL;

// tile rect -> rect in super-tile units (min rounds down, the exclusive max rounds up); s = 0: unchanged
__device__ __forceinline__ uint2 ss_super_rect(uint2 rc, int s) {
    if (s == 0) return rc;
    constexpr uint32_t sx = GSR_SUPER_SX, sy = GSR_SUPER_SY, rx = (1u << sx) - 1u, ry = (1u << sy) - 1u;
    return make_uint2(((rc.x & 0xffffu) >> sx) | (((rc.x >> 16) >> sy) << 16),
                      (((rc.y & 0xffffu) + rx) >> sx) | ((((rc.y >> 16) + ry) >> sy) << 16));
}

// number of splitters <= tkey among split[0 .. B-2]  (split is ascending; B is a power of two)
__device__ __forceinline__ uint32_t ss_bucket(const uint32_t *split, int B, uint32_t tkey) {
    uint32_t lo = 0;
    for (int step = B >> 1; step > 0; step >>= 1) {
        const uint32_t probe = lo + (uint32_t)step;  // candidate count: splitters [0, probe) all <= tkey ?
        if (split[probe - 1u] <= tkey) lo = probe;
    }
    return lo;  // in [0, B-1]  (split[B-1] is never probed: probe - 1 <= B - 2)
}

// the same for W keys at once: W independent probe chains per step instead of W searches one after the other
// (a search is log2(B) DEPENDENT LDS round trips; nothing else in these kernels costs as much)
template <int W>
__device__ __forceinline__ void ss_bucketN(const uint32_t *split, int B, const uint32_t (&tkey)[W], uint32_t (&out)[W]) {
    uint32_t lo[W];
#pragma unroll
    for (int u = 0; u < W; u++) lo[u] = 0u;
    for (int step = B >> 1; step > 0; step >>= 1) {
        uint32_t pv[W];
#pragma unroll
        for (int u = 0; u < W; u++) pv[u] = split[lo[u] + (uint32_t)step - 1u];
#pragma unroll
        for (int u = 0; u < W; u++)
            if (pv[u] <= tkey[u]) lo[u] += (uint32_t)step;
    }
#pragma unroll
    for (int u = 0; u < W; u++) out[u] = lo[u];
}

// wave64 match-any on the low `nbits` of d among the valid lanes: returns the mask of lanes holding the same value
__device__ __forceinline__ uint64_t ss_match(uint32_t d, int nbits, bool valid) {
    uint64_t same = __builtin_amdgcn_ballot_w64(valid);
    for (int b = 0; b < nbits; b++) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = __builtin_amdgcn_ballot_w64(bit);
        same &= bit ? bal : ~bal;
    }
    return same;
}

// One stable LSD pass (8-bit digit at `shift`) over n records held in LDS.  Wave w owns the contiguous quarter
// [w q, (w+1) q): counts per wave, cursors = digit start + earlier waves, then barrier-free ranking rounds (LDS
// operations of one wave retire in order).  s_cur: 4 x 256 words, s_w: 4 words.
template <bool PAIRS>
__device__ __forceinline__ void lds_radix_pass(const uint32_t *kin, const uint32_t *vin, uint32_t *kout, uint32_t *vout,
                                               int n, int shift, uint32_t *s_cur, uint32_t *s_w) {
    const int tid = (int)threadIdx.x, wave = gsr_wave(), lane = gsr_lane();
    const int q = (((n + 3) >> 2) + 63) & ~63;
    const int rounds = q >> 6, lo = wave * q;
    const uint64_t lt = gsr_lanemask_lt();
    for (int i = tid; i < 4 * 256; i += kT) s_cur[i] = 0u;
    __syncthreads();
    for (int r = 0; r < rounds; r++) {
        const int i = lo + (r << 6) + lane;
        if (i < n) atomicAdd(&s_cur[wave * 256 + (int)((kin[i] >> shift) & 255u)], 1u);
    }
    __syncthreads();
    {
        const uint32_t c0 = s_cur[tid], c1 = s_cur[256 + tid], c2 = s_cur[512 + tid], c3 = s_cur[768 + tid];
        const uint32_t tot = c0 + c1 + c2 + c3;
        uint32_t all;
        const uint32_t excl = gsr_block_incl_scan(tot, s_w, all) - tot;
        s_cur[tid] = excl;
        s_cur[256 + tid] = excl + c0;
        s_cur[512 + tid] = excl + c0 + c1;
        s_cur[768 + tid] = excl + c0 + c1 + c2;
    }
    __syncthreads();
    uint32_t *cur = s_cur + wave * 256;
    for (int r = 0; r < rounds; r++) {
        const int i = lo + (r << 6) + lane;
        const bool valid = i < n;
        const uint32_t key = valid ? kin[i] : 0u;
        const uint32_t val = (PAIRS && valid) ? vin[i] : 0u;
        const uint32_t d = (key >> shift) & 255u;
        const uint64_t same = ss_match(d, 8, valid);
        const uint32_t rank = (uint32_t)__popcll(same & lt);
        if (valid) {
            const uint32_t pos = cur[d] + rank;
            kout[pos] = key;
            if (PAIRS) vout[pos] = val;
        }
        __builtin_amdgcn_wave_barrier();  // every lane has read its cursor before the group leader moves it
        if (valid && rank == 0u) cur[d] += (uint32_t)__popcll(same);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------
// ss_prepare: ONE workgroup of 1024 threads per frame decides everything the compaction workgroups have in common --
// V, the bucket count, which splitter table classifies this frame (the kept one taken blind, the kept one validated
// against samples, or a new one drawn from them) and which run of preprocess blocks every compaction workgroup owns.
// Rounds 2-4 had every one of the 256 compaction workgroups repeat this chain for itself (no launch in between): by
// round 5's cycle stamps 20 k (kept table) to 70 k cycles (new table) of one wave per SIMD per workgroup at 8-10 cycles
// per dependent instruction, against 10 k for the compaction itself.  Sixteen waves on one CU hide each other's LDS
// round trips and carry a quarter of the per-thread work each; the price is one launch boundary.
// ---------------------------------------------------------------------------------------------------------
constexpr int kPT = 1024;            // threads of the prepare workgroup
constexpr int kPW = kPT / GSR_WAVE;  // 16 waves

// inclusive scan over the 1024 threads; s_w16: 16 words of LDS.  Two barriers.
__device__ __forceinline__ uint32_t ss_scan1024(uint32_t v, uint32_t *s_w16, uint32_t &total) {
    const uint32_t incl = gsr_wave_incl_scan(v);
    const int lane = gsr_lane(), wave = (int)(threadIdx.x >> 6);
    if (lane == 63) s_w16[wave] = incl;
    __syncthreads();
    uint32_t add = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kPW; w++) {
        const uint32_t x = s_w16[w];
        add += w < wave ? x : 0u;
        tot += x;
    }
    total = tot;
    __syncthreads();
    return incl + add;
}

// One stable LSD pass (8-bit digit at `shift`) over n keys in LDS by the 16 waves of the prepare workgroup (the sixteen-
// wave form of lds_radix_pass: wave w owns a contiguous sixteenth).  s_cur: 16 x 256 words, s_w16: 16 words.
__device__ __forceinline__ void ss_radix_pass16(const uint32_t *kin, uint32_t *kout, int n, int shift, uint32_t *s_cur,
                                                uint32_t *s_w16) {
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = gsr_lane();
    const int q = (((n + kPW - 1) / kPW) + 63) & ~63;
    const int rounds = q >> 6, lo = wave * q;
    const uint64_t lt = gsr_lanemask_lt();
    for (int i = tid; i < kPW * 256; i += kPT) s_cur[i] = 0u;
    __syncthreads();
    for (int r = 0; r < rounds; r++) {
        const int i = lo + (r << 6) + lane;
        if (i < n) atomicAdd(&s_cur[wave * 256 + (int)((kin[i] >> shift) & 255u)], 1u);
    }
    __syncthreads();
    {
        uint32_t c[kPW], tot = 0;
#pragma unroll
        for (int w = 0; w < kPW; w++) {
            c[w] = tid < 256 ? s_cur[w * 256 + tid] : 0u;
            tot += c[w];
        }
        uint32_t all;
        uint32_t run = ss_scan1024(tot, s_w16, all) - tot;
        if (tid < 256) {
#pragma unroll
            for (int w = 0; w < kPW; w++) {
                s_cur[w * 256 + tid] = run;
                run += c[w];
            }
        }
    }
    __syncthreads();
    uint32_t *cur = s_cur + wave * 256;
    for (int r = 0; r < rounds; r++) {
        const int i = lo + (r << 6) + lane;
        const bool valid = i < n;
        const uint32_t key = valid ? kin[i] : 0u;
        const uint32_t d = (key >> shift) & 255u;
        const uint64_t same = ss_match(d, 8, valid);
        const uint32_t rank = (uint32_t)__popcll(same & lt);
        if (valid) kout[cur[d] + rank] = key;
        __builtin_amdgcn_wave_barrier();  // every lane has read its cursor before the group leader moves it
        if (valid && rank == 0u) cur[d] += (uint32_t)__popcll(same);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
}

// The prepare workgroup keeps every block count of the model in LDS as 16-bit running sums: up to kPrepBlocks preprocess
// blocks (8.4 M Gaussians; larger models take the LSD radix depth sort: api.hip).  Static LDS: 132 KB of the CU's 160
// (a launch may not ask for more than 64 KB dynamically).
constexpr int kPrepBlocks = 32 * kPT;
__host__ __device__ inline int ss_prepare_per(int nb1) { return (((nb1 + kPT - 1) / kPT) + 7) & ~7; }

// gsr_quad_order_block (gsr_internal.h: which compositing workgroup takes which quadrants, from their costs in the previous
// frame) for the 1024 threads of the prepare launch's SECOND workgroup: eight quadrants per thread instead of 32.  It
// depends on nothing of the current frame, and as the one long workgroup of a later, shorter kernel -- tile_starts in
// round 3, the partition pass in round 4 -- it set that kernel's time one frame at a time (ss_partition 12.2 -> 8.2 us
// without it); beside the prepare workgroup it is free.  (Measured on the way: as an extra workgroup of preprocess it
// raised that kernel's registers from 72 to 113 -- 27.3 instead of 20.5 us.)
// With `coop_cap` > 0 the deal also names the COOPERATIVE quadrants of the frame (render.hip render_coop_quadrant): the
// first coop_cap / 8 of every XCD's cost order whose cost was above GSR_COOP_FACTOR_X16 / 16 of the mean quadrant's get a
// workgroup in front of the compositor's main grid each (coop_list[j]: the quadrant of workgroup j, on the quadrant's own
// XCD; 0xFFFFFFFF: none), and bit 31 of their entry in quad_order tells the wave the deal gave them to in the main grid to
// leave them alone.
__device__ __forceinline__ void ss_quad_order_1024(const uint32_t *__restrict__ quad_work, int Q,
                                                   uint32_t *__restrict__ quad_order, int cus_per_xcd,
                                                   uint32_t *__restrict__ coop_list, int coop_cap,
                                                   GsrHeader *__restrict__ hdr, const uint32_t *__restrict__ tile_dirty) {
    __shared__ uint32_t s_qb[GSR_XCDS * 256];
    __shared__ uint32_t s_xbase[GSR_XCDS];
    __shared__ uint32_t s_w16[kPW], s_cs16[kPW];
    __shared__ uint32_t s_coop_n;
    constexpr int KQ = 32 * GSR_BLOCK / kPT;  // 8 quadrants per thread: Q <= 32 x 256
    const int tid = (int)threadIdx.x;
    const int T = Q >> 2;
    if (tid == 0) s_coop_n = 0u;
    const bool reuse = tile_dirty != nullptr && hdr->td_reuse != 0u;
    const uint32_t tok = hdr->td_token + 1u;
    uint32_t c[KQ], cost[KQ], qmx = 0, csum = 0;
#pragma unroll
    for (int k = 0; k < KQ; k++) {
        const int q = tid + k * kPT;
        c[k] = q < Q ? min(quad_work[q], (1u << 24) - 1u) : 0u;
        // (tile reuse, render.hip: a tile nobody marked keeps the previous frame's pixels -- its quadrants cost nothing in this
        //  frame, sort behind everything else and share workgroups that retire at once, instead of holding a residency slot
        //  each beside one quadrant that does work.  The token is still the previous frame's here: ss_compact advances it.)
        if (reuse && q < Q && tile_dirty[q >> 2] != tok) c[k] = 0u;
        cost[k] = c[k];
        qmx = max(qmx, c[k]);
        csum += c[k] >> 4;  // (in sixteenths: 8192 costs below 2^24 stay below 2^32)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        qmx = max(qmx, (uint32_t)__shfl_xor((int)qmx, o, 64));
        csum += (uint32_t)__shfl_xor((int)csum, o, 64);
    }
    if ((tid & 63) == 0) {
        s_w16[tid >> 6] = qmx;
        s_cs16[tid >> 6] = csum;
    }
    for (int i = tid; i < GSR_XCDS * 256; i += kPT) s_qb[i] = 0u;
    for (int i = tid; i < coop_cap; i += kPT) coop_list[i] = 0xFFFFFFFFu;  // (coop_cap = 0: no cooperative quadrants)
    __syncthreads();
    uint32_t total16 = 0u;
#pragma unroll
    for (int w = 0; w < kPW; w++) {
        qmx = max(qmx, s_w16[w]);
        total16 += s_cs16[w];
    }
    // cost > factor x mean  <=>  cost > factor_x16 x (sum / 16) / Q
    const uint32_t coop_thr =
        (uint32_t)min((uint64_t)0xFFFFFFFEull, ((uint64_t)total16 * GSR_COOP_FACTOR_X16) / (uint32_t)max(Q, 1));
    // bucket = 255 - floor(cost * 256 / (max + 1)): cost < 2^24, so the product fits 32 bits after the shift
    const int sh = qmx >= (1u << 16) ? 8 : 0;  // (keeps cost * 256 below 2^32 and the divisor non-zero)
    const uint32_t div = (qmx >> sh) + 1u;
    const float inv = 256.0f / (float)div;
#pragma unroll
    for (int k = 0; k < KQ; k++) {
        const int q = tid + k * kPT;
        const uint32_t xcd = (uint32_t)(q >> 2) % GSR_XCDS;
        c[k] = xcd * 256u + 255u - min(255u, (uint32_t)((float)(c[k] >> sh) * inv));  // bin = (XCD, cost class)
        if (q < Q) atomicAdd(&s_qb[c[k]], 1u);
    }
    __syncthreads();
    {  // exclusive running sum over the 2048 bins: thread t owns bins 2 t, 2 t + 1 (XCD t / 128)
        const uint32_t v0 = s_qb[2 * tid], v1 = s_qb[2 * tid + 1];
        uint32_t tot;
        uint32_t run = ss_scan1024(v0 + v1, s_w16, tot) - (v0 + v1);
        if ((tid & 127) == 0) s_xbase[tid >> 7] = run;  // first slot of the XCD's list
        s_qb[2 * tid] = run;
        s_qb[2 * tid + 1] = run + v0;
    }
    __syncthreads();
    const uint32_t cus = (uint32_t)max(cus_per_xcd, 1);
#pragma unroll
    for (int k = 0; k < KQ; k++) {
        const int q = tid + k * kPT;
        if (q < Q) {
            const uint32_t xcd = c[k] >> 8;
            const uint32_t p = atomicAdd(&s_qb[c[k]], 1u) - s_xbase[xcd];        // position in the XCD's sorted list
            const uint32_t nwg = ((uint32_t)T - 1u - xcd) / GSR_XCDS + 1u;         // workgroups (= tiles) of this XCD
            const uint32_t slot = p >> 2, round = slot / cus, idx = slot - round * cus;
            const uint32_t size = min(cus, nwg - round * cus);
            const uint32_t cu = (round & 1u) ? size - 1u - idx : idx;
            const uint32_t b = GSR_XCDS * (round * cus + cu) + xcd;
            // (the cooperative workgroups come FIRST in the compositor's grid: workgroup j runs on XCD j mod 8, and the
            //  quadrant's list is in the L2 of XCD tile mod 8 = xcd; the wave the deal gives the quadrant to in the main grid
            //  finds it marked and leaves it alone -- in the entry it reads anyway, not behind one more round trip)
            const bool co = p < (uint32_t)(coop_cap / GSR_XCDS) && cost[k] > coop_thr;
            quad_order[4u * b + (p & 3u)] = (uint32_t)q | (co ? 0x80000000u : 0u);
            if (co) {
                coop_list[xcd + GSR_XCDS * p] = (uint32_t)q;
                atomicAdd(&s_coop_n, 1u);
            }
        }
    }
    __syncthreads();
    if (tid == 0) hdr->coop_quads = s_coop_n;  // (tests and tools: gsr_debug_sort_state)
}

__device__ __forceinline__ void ss_prepare_body(int P, int nb1, int bmax, int nbc,
                                                const uint2 *__restrict__ block_recs,
                                                const uint32_t *__restrict__ block_counts,
                                                const uint32_t *__restrict__ splitters,
                                                uint32_t *__restrict__ splitters_new, uint32_t *__restrict__ seg_off,
                                                uint32_t *__restrict__ seg_first, GsrHeader *__restrict__ hdr,
                                                uint64_t *__restrict__ dbg, const float *__restrict__ view,
                                                uint32_t sig, const int role, const bool pc_enabled) {
    // role 0: the frame's plan -- V, bucket count, which table classifies, samples, new splitters; role 1: the runs of blocks
    // of the compaction workgroups.  Two workgroups since round 6: both start from the same sums of the block counts
    // (summed twice: 23 KB from the L2), neither reads what the other writes, and the frame's sort waits for the longer of
    // the two chains instead of their sum (31-35 k cycles as one workgroup by the stamps, of which the ranges ~4 k).
    const unsigned dbg_wg = 0; (void)dbg_wg;
    SS_STAMP(dbg, 0);
#ifdef GSR_SS_TIMING
    if (role == 0 && threadIdx.x < 8) dbg[48 + threadIdx.x] = 0ull;
#endif
    __shared__ uint32_t s_key[2 * kMaxSamples];
    __shared__ __attribute__((aligned(16))) uint32_t s_cur[kPW * 256];
    __shared__ uint32_t s_split[2048], s_hist[2048];  // (bmax <= 2048)
    __shared__ uint32_t s_pex[kPT + 8];               // visible Gaussians before every thread's slice
    __shared__ __attribute__((aligned(16))) uint16_t s_cnt16[kPrepBlocks];  // running sums inside every slice
    __shared__ uint32_t s_w16[kPW];
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = gsr_lane();
    // ---- who owns what.  Thread t sums the counts of its contiguous slice of `per` preprocess blocks (a multiple of 8:
    // whole 16-byte LDS words); a scan of the 1024 slice sums gives V and every slice's position in the running sum of
    // VISIBLE Gaussians.  Compaction workgroups own consecutive runs of blocks of equal cost (below), however the visible
    // Gaussians are spread over the index range; samples are taken at equal steps of the running sum: uniform over the
    // visible Gaussians.
    const int per = ss_prepare_per(nb1);
    // What decides whether the splitters in the state are taken as they are is requested NOW, with the table itself: by
    // the time the counts are summed it has all arrived.
    const uint32_t h_magic = hdr->ss_magic, h_buckets = hdr->ss_buckets, h_bad = hdr->ss_bad, h_trust = hdr->ss_trust,
                   h_P = hdr->ss_P, h_near = hdr->ss_near, h_near_fail = hdr->ss_near_fail, h_vfail = hdr->ss_vfail;
    // (the block cache's words too: thread 0 turns them over below, in front of a barrier everybody waits at)
    const uint32_t h_pc_pending = hdr->pc_pending, h_pc_parity = hdr->pc_parity, h_pc_sig_next = hdr->pc_sig_next,
                   h_pc_hit = hdr->pc_hit;
    bool same_view = true, near_view = true;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t was = hdr->ss_view[k];
        same_view = same_view && __float_as_uint(view[k]) == was;
        near_view = near_view && fabsf(view[k] - __uint_as_float(was)) <= kNearTol;  // (false on a NaN)
    }
    uint32_t pre_sp[2];
#pragma unroll
    for (int k = 0; k < 2; k++) pre_sp[k] = tid + k * kPT < bmax - 1 ? splitters[tid + k * kPT] : 0xFFFFFFFFu;
    // the counts: coalesced loads, eight in flight per thread, into LDS as 16-bit words (zero behind the last block)
    for (int i0 = tid; i0 < per * kPT; i0 += 8 * kPT) {
        uint32_t c[8];
#pragma unroll
        for (int u = 0; u < 8; u++) c[u] = i0 + u * kPT < nb1 ? block_counts[i0 + u * kPT] : 0u;
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (i0 + u * kPT < per * kPT) s_cnt16[i0 + u * kPT] = (uint16_t)c[u];
    }
    __syncthreads();
    uint32_t mine = 0;
    {
        // the slice: per / 8 aligned 16-byte words (8 counts each), turned into their running sums in place (a slice holds
        // <= 32 blocks x 256: fits 16 bits)
        uint4 *sl = reinterpret_cast<uint4 *>(s_cnt16 + (size_t)tid * per);
        for (int w = 0; w < (per >> 3); w++) {
            const uint4 v = sl[w];
            const uint32_t x[4] = {v.x, v.y, v.z, v.w};
            uint32_t o[4];
#pragma unroll
            for (int h = 0; h < 4; h++) {
                mine += x[h] & 0xffffu;
                const uint32_t a = mine;
                mine += x[h] >> 16;
                o[h] = a | (mine << 16);
            }
            sl[w] = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
    SS_STAMP(dbg, 1);
    uint32_t V;
    const uint32_t p_incl = ss_scan1024(mine, s_w16, V), p_excl = p_incl - mine;
    (void)p_incl;
    s_pex[tid] = p_excl;
    if (tid == 0) s_pex[kPT] = V;
    if (tid == 0 && role == 0) {  // first kernel of the frame that touches the header
        // (block cache, preprocess.hip: the slot this frame's preprocess wrote its camera and poses to becomes the current one)
        if (pc_enabled && h_pc_pending == GSR_PC_MAGIC) {
            hdr->pc_parity = (h_pc_parity & 1u) ^ 1u;
            hdr->pc_sig = h_pc_sig_next;
            hdr->pc_magic = GSR_PC_MAGIC;
        } else {
            hdr->pc_magic = 0u;
        }
        hdr->pc_pending = 0u;
        hdr->pc_hit_last = pc_enabled ? h_pc_hit : 0u;
        hdr->pc_hit = 0u;
        hdr->td_skipped = 0u;  // (the tile token is advanced by ss_compact: the deal beside this workgroup still reads it)
        hdr->V = V;
        hdr->R = 0u;
        hdr->overflow = 0u;
        hdr->coop_timeout_now = 0u;
        hdr->r_capacity = 0u;
        hdr->R_raw = 0u;
        hdr->tile_queue = 0u;
    }
    if (V == 0u) {
        if (role == 1)
            for (int j = tid; j <= nbc; j += kPT) {
                seg_off[j] = 0u;
                seg_first[j] = (uint32_t)nb1;
            }
        if (tid == 0 && role == 0) {
            hdr->ss_fresh = 0u;
            hdr->ss_B = 256u;
            hdr->ss_stride = 1u;
            hdr->ss_near = 0u;
            hdr->ss_moved = 1u;
        }
        return;
    }
    __syncthreads();
    SS_STAMP(dbg, 22);
    if (role == 1) {
    // ---- the run of blocks of every compaction workgroup.  Ownership follows a COST: a block costs its visible Gaussians
    // (records to classify and move) + kBlockCost (its 256 keys have to be fetched and tested whatever they hold) -- equal
    // record shares alone hand a workgroup in an empty stretch of the model a thousand blocks to sweep.  The cost before
    // block j is (records before j) + kBlockCost j; workgroup b's run starts at the block in which that crosses b's
    // share boundary: thread b finds boundary b.
    if (tid <= nbc) {
        uint32_t j = 0u, before_j = 0u;
        if (tid == nbc) {
            j = (uint32_t)nb1;
            before_j = V;
        } else if (tid > 0) {
            const uint32_t W = V + kBlockCost * (uint32_t)nb1;
            const uint32_t t = (uint32_t)((double)W * (double)tid / (double)nbc);
            uint32_t us = 0u;  // last slice whose cost-before is <= t
#pragma unroll
            for (int st = kPT / 2; st > 0; st >>= 1)
                if (s_pex[us + (uint32_t)st] + kBlockCost * min((uint32_t)nb1, (us + (uint32_t)st) * (uint32_t)per) <= t)
                    us += (uint32_t)st;
            const uint32_t jb = min((uint32_t)nb1, us * (uint32_t)per), je = min((uint32_t)nb1, jb + (uint32_t)per);
            uint32_t pos = 0u;  // last block of the slice whose cost-before is <= t (the slice's first one qualifies)
#pragma unroll
            for (int st = 16; st > 0; st >>= 1) {
                const uint32_t k = jb + pos + (uint32_t)st;
                if (pos + (uint32_t)st < (uint32_t)per && k < je &&
                    s_pex[us] + (uint32_t)s_cnt16[k - 1u] + kBlockCost * k <= t)
                    pos += (uint32_t)st;
            }
            j = jb + pos;
            before_j = s_pex[us] + (pos > 0u ? (uint32_t)s_cnt16[j - 1u] : 0u);
        }
        seg_off[tid] = before_j;
        seg_first[tid] = j;
    }
        return;
    }
    // A fixed sensor camera (GSWorld's right_cam and the like) over a scene that stands still: the view matrix is bit
    // for bit the one the splitters in the state were built under and the last frames that classified with the kept
    // table came out as balanced as exact quantiles of an unchanged scene do (ss_trust, kept by ss_partition /
    // ss_buckets) -- then they are taken as they are: no samples, no check.  Splitters only decide the BALANCE of the
    // buckets, never the order; but balance matters: under a fixed camera a MOVING scene (an arm swinging through a
    // depth range that was empty a frame ago, where the kept buckets are wide) can put ten thousand records into one
    // bucket, far beyond the LDS, and that bucket's workgroup then sorts in global memory for a millisecond.  Such a
    // scene never earns the trust; its frames check the kept table against samples below.
    //
    // Bucket count (left in the header: every later kernel of the frame, and band_ranges, read it there).  A frame that
    // SAMPLES aims at GSR_SS_PER_BUCKET records per bucket: its table is an estimate, buckets come out at up to a few
    // times their share, and the slowest bucket workgroup sets the time of a launch that does not fill the chip.  A BLIND
    // frame's table holds exact quantiles of the same scene: every bucket gets its share to within the depth ties, so it
    // takes buckets of GSR_SS_PER_BUCKET_FULL records -- half as many workgroups with half the fixed cost per record
    // (measured, round 5: 1024 against 512 everywhere, static camera: eight frames per launch +3.1 % on the sensor view,
    // +5.7 % on the dense one, dense view alone +5.5 %, sensor view alone -0.4 %; the closed loop, whose frames sample,
    // -5.5 %).  The first blind frame after sampling ones finds a table of twice its count and takes every second entry
    // (the 2i-th of 2B quantiles is the i-th of B): `stride`.  A frame that samples again finds a table of half its count,
    // which fails `reuse` below: one drawn table per change from resting to moving.
    int B0 = ss_num_buckets(V, bmax, (uint32_t)GSR_SS_PER_BUCKET);
    // (hysteresis: a camera whose visible count hovers around a power-of-two multiple of the bucket size -- the surrogate's
    //  wrist camera sees 263 k Gaussians, 512 x 512 = 262 144 -- would flip between two bucket counts from frame to frame,
    //  and a kept table of the other count is no table: a new one drawn nearly every step.  The kept count stays while its
    //  buckets average 224 .. 576 records: an eighth either side of the nominal 256 .. 512.)
    if (GSR_SS_NEAR != 0 && h_magic == kSplitMagic && h_P == sig && h_buckets >= 256u && h_buckets <= (uint32_t)bmax &&
        (h_buckets & (h_buckets - 1u)) == 0u && V >= h_buckets * 224u && V <= h_buckets * 576u)
        B0 = (int)h_buckets;
    const int Bf = ss_num_buckets(V, bmax, (uint32_t)GSR_SS_PER_BUCKET_FULL);  // (B0 or B0 / 2)
    const bool table_ok = h_magic == kSplitMagic && h_buckets <= (uint32_t)bmax && (GSR_SS_IGNORE_BAD || h_bad == 0u) &&
                          h_trust <= 255u && h_P == sig;
    const bool blind_same = same_view && table_ok && (h_buckets == (uint32_t)Bf || h_buckets == 2u * (uint32_t)Bf) &&
                            h_trust >= (uint32_t)GSR_SS_TRUST_MIN;
    // Round 6: a camera that MOVES A LITTLE (a wrist camera riding on the arm: millimetres and a fraction of a degree per
    // step) over the table its previous frame left -- the exact quantiles of THAT frame's depth order, one step old.  Depths
    // shift smoothly under such a move, so the buckets stay near their shares; drawing samples to confirm it (two binary
    // searches per sample, a gather, a classification: 14.2 against 8.3 us on this one workgroup) is most of what the frame's
    // sort waits for.  Such a frame goes blind as well once the kept table has been earning its trust under the moving
    // camera -- the frames before it classified with it and no bucket came out above TWICE its share (ss_buckets: the
    // bound for frames whose view differs from the table's; a static camera keeps 1.25) -- with the full bucket count (a
    // bucket of twice its share still fits the LDS), and a near-blind frame that does come out unbalanced doubles the
    // trust the next one has to show (ss_near_fail): a rig whose small moves do unbalance its tables ends up sampling.
    uint32_t near_fail = h_near_fail > 6u ? 0u : h_near_fail;  // (a fresh state holds garbage)
    if (h_magic == kSplitMagic && h_near == 1u && h_bad != 0u && near_fail < 6u) near_fail++;
    const bool blind_near = GSR_SS_NEAR != 0 && !same_view && near_view && table_ok && h_buckets == (uint32_t)B0 &&
                            h_trust >= (2u << near_fail);
    bool blind = blind_same || blind_near;
    const int B = blind_same ? Bf : B0;
    const uint32_t stride = blind_same ? h_buckets / (uint32_t)Bf : 1u;
    if (tid == 0) {
        hdr->ss_B = (uint32_t)B;
        hdr->ss_stride = stride;
        hdr->ss_moved = same_view ? 0u : 1u;
        hdr->ss_near_fail = near_fail;
    }
    // (at least kMinSamples: with few buckets the splitters would otherwise be cut from two samples each, and one bucket
    // in a few hundred frames outgrows the LDS)
    const uint32_t S = (uint32_t)min(kMaxSamples, max(kSamplesPerBucket * B, kMinSamples));
    const int logS = ss_log2((int)S);  // (S is a power of two)
    // (GSR_SS_PROBE_EVERY, an A/B switch: the kept table checked against this frame's samples only when the last check passed,
    //  or every n-th frame after one that failed -- ss_vfail: frames since.  Off: see the macro.)
    const bool table_there = h_magic == kSplitMagic && h_buckets == (uint32_t)B && h_P == sig;
    const uint32_t vfail = (h_magic == kSplitMagic && h_vfail <= 255u) ? h_vfail : 0u;
    const bool attempt = table_there && (GSR_SS_PROBE_EVERY == 0 || vfail == 0u || vfail >= (uint32_t)GSR_SS_PROBE_EVERY);
    const bool need_table = blind || attempt;
    bool ascending = false;
    if (need_table) {
    // the kept table into LDS (used blind, or validated below)
    if (stride == 1u) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int i = tid + k * kPT;
            if (i < B) {
                s_split[i] = i < B - 1 ? pre_sp[k] : 0xFFFFFFFFu;
                s_hist[i] = 0u;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 2; k++)
            if (tid + k * kPT < bmax) s_hist[tid + k * kPT] = pre_sp[k];
        __syncthreads();
        uint32_t v[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int i = tid + k * kPT;
            v[k] = i < B - 1 ? s_hist[2 * i + 1] : 0xFFFFFFFFu;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int i = tid + k * kPT;
            if (i < B) {
                s_split[i] = v[k];
                s_hist[i] = 0u;
            }
        }
    }
    __syncthreads();
    // (a state buffer handed back by the allocator can carry a valid-looking header over arrays somebody else wrote in
    // between: what is taken for BALANCE must still be an ascending table, or the order breaks)
    uint32_t unsorted = 0u;
    for (int i = tid; i + 1 < B - 1; i += kPT) unsorted |= s_split[i + 1] < s_split[i] ? 1u : 0u;
    ascending = __syncthreads_or((int)unsorted) == 0;
    }
    blind = blind && ascending;
    if (tid == 0) {
        hdr->ss_blind = blind ? 1u : 0u;
        hdr->ss_near = (blind && !blind_same) ? 1u : 0u;
    }
    SS_STAMP(dbg, 23);
    if (blind) {
        if (tid == 0) {
            hdr->ss_fresh = 0u;
            hdr->ss_vfail = 0u;
        }
        return;
    }
    // ---- samples.  Sample s belongs to the preprocess block that holds visible Gaussian floor(s V / S) of the index
    // order: uniform over the VISIBLE Gaussians.  (Until round 4 the sample was the FIRST visible key of that block: the
    // same thing for a model in random order, but in a spatially sorted model -- gsworld_amd/layout.py -- the visible
    // Gaussians sit in a fifth of the blocks, several samples drew the same key, the splitters came out uneven every
    // frame and the buckets outgrew the LDS: ss_buckets 12 -> 72 us.)  Two binary searches per sample -- the thread slice
    // (s_pex), then the block inside it (the slice's running sums) -- four samples side by side per thread.  (Measured in
    // round 5 and dropped: every thread handing out the samples of its own slice in one walk -- the visible Gaussians of a
    // wrist camera sit in a handful of slices, whose threads then do all the work.)  A block's n samples are its FIRST n
    // visible records -- one or two cache lines of the block's compacted records instead of a line per sample; inside a
    // block of 256 consecutive Gaussians -- neighbours in space for a laid-out model, a random subset otherwise -- any n
    // records are as good a sample of the block's depths as any other.
    // (S = 2048 or 4096: two or four samples per thread -- sixteen waves' dependent LDS reads on one CU are what this phase
    //  costs, 7.5 k cycles with four chains per thread whatever S was; with S = 2048 the other two only repeated the last
    //  sample)
    auto draw = [&](auto swc) {
        constexpr int SW = decltype(swc)::value;
        const double s_per_rank = (double)S / (double)V;
        uint32_t tgt[SW], u[SW];
#pragma unroll
        for (int q = 0; q < SW; q++) {
            const uint32_t smp = (uint32_t)(tid + q * kPT);
            tgt[q] = (uint32_t)(((uint64_t)min(smp, S - 1u) * V) >> logS);
            u[q] = 0u;
        }
#pragma unroll
        for (int st = kPT / 2; st > 0; st >>= 1) {
#pragma unroll
            for (int q = 0; q < SW; q++)
                if (s_pex[u[q] + (uint32_t)st] <= tgt[q]) u[q] += (uint32_t)st;  // last slice that starts at or before
        }
        uint32_t pos[SW], base[SW], n[SW], lt[SW];
#pragma unroll
        for (int q = 0; q < SW; q++) {
            base[q] = min((uint32_t)nb1, u[q] * (uint32_t)per);
            n[q] = min((uint32_t)nb1, base[q] + (uint32_t)per) - base[q];
            lt[q] = tgt[q] - s_pex[u[q]];
            pos[q] = 0u;
        }
#pragma unroll
        for (int st = 16; st > 0; st >>= 1) {  // first block of the slice whose running sum exceeds lt (per <= 32)
#pragma unroll
            for (int q = 0; q < SW; q++)
                if (pos[q] + (uint32_t)st <= n[q] && (uint32_t)s_cnt16[base[q] + pos[q] + (uint32_t)st - 1u] <= lt[q])
                    pos[q] += (uint32_t)st;
        }
        uint32_t rec[SW];
#pragma unroll
        for (int q = 0; q < SW; q++) {
            const uint32_t smp = min((uint32_t)(tid + q * kPT), S - 1u);
            const uint32_t pb = min(pos[q], n[q] - 1u), blk = base[q] + pb;
            const uint32_t before_b = pb > 0u ? (uint32_t)s_cnt16[blk - 1u] : 0u;
            const uint32_t cnt_b = (uint32_t)s_cnt16[blk] - before_b;
            // how many samples before this one fall into the same block: the block's first sample is the first whose
            // rank reaches the records before the block (a last-bit error of the quotient only picks a neighbour)
            const uint32_t s_first = (uint32_t)__builtin_ceil((double)(s_pex[u[q]] + before_b) * s_per_rank);
            const uint32_t off = cnt_b > 0u ? min(smp - min(smp, s_first), cnt_b - 1u) : 0u;
            rec[q] = blk * (uint32_t)GSR_BLOCK + off;  // the record that lends its key
        }
        SS_STAMP(dbg, 24);
        // the keys: every gather of the thread in flight at once
        uint32_t k[SW];
#pragma unroll
        for (int q = 0; q < SW; q++) k[q] = (uint32_t)(tid + q * kPT) < S ? block_recs[rec[q]].y : 0u;
#pragma unroll
        for (int q = 0; q < SW; q++)
            if ((uint32_t)(tid + q * kPT) < S) s_key[tid + q * kPT] = k[q] & kKeyMask;
    };
    if (S <= 2u * (uint32_t)kPT) draw(std::integral_constant<int, 2>{});
    else draw(std::integral_constant<int, kMaxSamples / kPT>{});
    __syncthreads();
    SS_STAMP(dbg, 2);
    // ---- splitters.  A closed-loop camera hardly moves: the exact quantiles ss_buckets left in the state after the
    // previous frame usually still cut THIS frame's samples evenly.  Check that (the table must be ascending -- a fresh
    // state holds garbage -- and no bucket may draw more than four times its share of the samples) and skip the sample
    // sort when it holds.
    // (the largest of B Poisson(2) sample counts grows with B: 8 passes for 512 buckets 9 times out of 10, 12 for 2048)
    const uint32_t reuse_max = 4u * (S / (uint32_t)B) + (B > 512 ? 2u * (uint32_t)(ss_log2(B) - 9) : 0u);
    const bool checked = attempt && S >= (uint32_t)B && ascending;
    bool reuse = checked;
    if (reuse) {
        uint32_t bad = 0;
        auto classify = [&](auto swc) {
            constexpr int SW = decltype(swc)::value;
            uint32_t tk[SW], bk[SW];
#pragma unroll
            for (int q = 0; q < SW; q++) tk[q] = (uint32_t)(tid + q * kPT) < S ? s_key[tid + q * kPT] : kNoKey;
            ss_bucketN<SW>(s_split, B, tk, bk);
#pragma unroll
            for (int q = 0; q < SW; q++)
                if (tk[q] != kNoKey) atomicAdd(&s_hist[bk[q]], 1u);
        };
        if (S <= 2u * (uint32_t)kPT) classify(std::integral_constant<int, 2>{});
        else classify(std::integral_constant<int, kMaxSamples / kPT>{});
        __syncthreads();
        for (int i = tid; i < B; i += kPT)
            if (s_hist[i] > reuse_max) bad = 1u;
        reuse = __syncthreads_or((int)bad) == 0;
    }
    if (!reuse) {
        // New splitters: the (i + 1) S / B-th smallest samples.  Splitters only decide the balance of the buckets, so the
        // samples are sorted by a 16-bit code -- their offset from the smallest sample, shifted until the largest fits
        // 16 bits: 1 / 65536 of the samples' range, exact when the range is below that (a plane seen from straight above)
        // -- in TWO 8-bit LSD passes instead of four over the full keys; a splitter is its code put back on the smallest
        // sample: ascending, and equal keys classify alike whatever the table.
        uint32_t mn = 0xFFFFFFFFu, mx = 0u;
        for (int i = tid; i < (int)S; i += kPT) {
            mn = min(mn, s_key[i]);
            mx = max(mx, s_key[i]);
        }
        mn = ss_wave_min(mn);
        mx = ss_wave_max(mx);
        __shared__ uint32_t s_mm[2 * kPW];
        if (lane == 0) {
            s_mm[wave] = mn;
            s_mm[kPW + wave] = mx;
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < kPW; w++) {
            mn = min(mn, s_mm[w]);
            mx = max(mx, s_mm[kPW + w]);
        }
        const uint32_t range = mx - mn;
        // Round 6: the quantiles from a HISTOGRAM of 12-bit codes first -- 4096 bins over the samples' range (s_cur, one
        // word per bin), one LDS atomic per sample, one scan of the bins, and every bin names the splitters whose rank falls
        // into it (S / B is a power of two: rank (i + 1) S / B): five barriers instead of the two LSD passes' fourteen and
        // their ranking rounds (a wrist camera draws a new table nearly every step: ss_prepare 17.3 -> us per closed-loop
        // step).  A splitter is then the lower edge of its bin: no finer than 1 / 4096 of the range, so a
        // bin that draws more than four shares of the samples (a clump of depths beside far outliers: its records could
        // not be cut apart and would outgrow a bucket's LDS) sends the frame down the exact route below.
#ifndef GSR_SS_HIST_SPLITTERS
#define GSR_SS_HIST_SPLITTERS 1
#endif
        bool drawn = false;
        if (GSR_SS_HIST_SPLITTERS) {
            static_assert(kPW * 256 == 4 * kPT, "a thread owns four of the 4096 bins");
            const int hshift = range < 4096u ? 0 : (32 - __builtin_clz(range)) - 12;
            const int lg = ss_log2((int)S) - ss_log2(B);  // samples per splitter: S / B = 2, 4 or 8
            for (int i = tid; i < 4 * kPT; i += kPT) s_cur[i] = 0u;
            __syncthreads();
            for (int i = tid; i < (int)S; i += kPT) atomicAdd(&s_cur[(s_key[i] - mn) >> hshift], 1u);
            __syncthreads();
            const uint4 c = *reinterpret_cast<const uint4 *>(s_cur + 4 * tid);
            const uint32_t cnt[4] = {c.x, c.y, c.z, c.w};
            const uint32_t mine4 = c.x + c.y + c.z + c.w, lim = 4u << lg;
            uint32_t tot4;
            uint32_t run = ss_scan1024(mine4, s_w16, tot4) - mine4;
            const bool heavy = __syncthreads_or((c.x > lim || c.y > lim || c.z > lim || c.w > lim) ? 1 : 0) != 0;
            if (!heavy) {
                // NOT into the table a later frame's validation reads before ss_buckets rewrites it: the drawn table goes to
                // its own array, which this frame's compaction and partition passes read
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (cnt[k] != 0u) {
                        // ranks [run, run + cnt) hold the samples of this bin: splitter i has rank (i + 1) << lg
                        const uint32_t i1 = (run + (1u << lg) - 1u) >> lg, i2 = (run + cnt[k] + (1u << lg) - 1u) >> lg;
                        const uint32_t edge = ((((uint32_t)(4 * tid + k)) << hshift) + mn) & kKeyMask;
                        for (uint32_t j = max(i1, 1u); j < i2 && j < (uint32_t)B; j++) splitters_new[j - 1u] = edge;
                    }
                    run += cnt[k];
                }
                if (tid == 0) splitters_new[B - 1] = 0xFFFFFFFFu;
                drawn = true;
            }
        }
        if (!drawn) {
            const int shift = range < 65536u ? 0 : (32 - __builtin_clz(range)) - 16;
            for (int i = tid; i < (int)S; i += kPT) s_key[i] = (s_key[i] - mn) >> shift;
            __syncthreads();
            ss_radix_pass16(s_key, s_key + kMaxSamples, (int)S, 0, s_cur, s_w16);
            ss_radix_pass16(s_key + kMaxSamples, s_key, (int)S, 8, s_cur, s_w16);
            const uint32_t *sorted = s_key;
            for (int i = tid; i < B; i += kPT) {
                const uint32_t q = (uint32_t)(((uint64_t)(i + 1) * S) / (uint32_t)B);
                splitters_new[i] = (i < B - 1 && q < S) ? (((sorted[q] << shift) + mn) & kKeyMask) : 0xFFFFFFFFu;
            }
        }
    }
    if (tid == 0) {
        hdr->ss_fresh = reuse ? 0u : 1u;
        // (frames since a check of the kept table failed; 0: the last check passed, or there was nothing to check)
        hdr->ss_vfail = checked ? (reuse ? 0u : 1u) : (table_there ? min(vfail + 1u, 255u) : 0u);
    }
    SS_STAMP(dbg, 3);
}

// ---------------------------------------------------------------------------------------------------------
// ss_compact: compaction + classification.  Workgroup b owns the preprocess blocks [seg_first[b], seg_first[b + 1]) that
// ss_prepare cut out for it; records are written as (index, key): the 64-bit little-endian view is key << 32 | index, the
// composite the global-memory fallback of ss_buckets sorts.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ss_compact_body(int bmax, const uint2 *__restrict__ block_recs,
                                                const uint32_t *__restrict__ block_counts, uint2 *__restrict__ pairs,
                                                uint32_t *__restrict__ table, const uint32_t *__restrict__ splitters,
                                                const uint32_t *__restrict__ splitters_new,
                                                const uint32_t *__restrict__ seg_off,
                                                const uint32_t *__restrict__ seg_first,
                                                GsrHeader *__restrict__ hdr, uint64_t *__restrict__ dbg0, const bool td_enabled) {
    extern __shared__ uint32_t smem[];
    uint64_t *dbg = dbg0; const unsigned dbg_wg = 64; (void)dbg_wg; (void)dbg;
    SS_STAMP(dbg, 8);
    // (tile reuse: what this frame's preprocess marked carries the NEXT token; from here on it is the current one -- the deal in
    //  the prepare launch has read the old one, the compositor compares with the new)
    if (td_enabled && blockIdx.x == 0 && threadIdx.x == 0) hdr->td_token = hdr->td_token + 1u;
#ifdef GSR_SS_TIMING
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
    uint32_t *s_split = smem;                     // [bmax]
    uint32_t *s_hist = s_split + bmax;            // [bmax]
    uint32_t *s_boff = s_hist + bmax;             // [4 kT + 1] offsets of up to 1024 blocks
    __shared__ uint32_t s_w[4];
    const int tid = (int)threadIdx.x, wave = gsr_wave(), lane = gsr_lane();
    const int me = (int)blockIdx.x;
    // everything this workgroup needs of the frame's plan in one round trip
    const uint32_t V = hdr->V;
    const uint32_t *__restrict__ split_src = hdr->ss_fresh != 0u ? splitters_new : splitters;
    const int first = (int)seg_first[me], last = (int)seg_first[me + 1];
    const uint32_t before = seg_off[me];
    if (V == 0u) return;
    const int B = (int)hdr->ss_B;
    const uint32_t stride = hdr->ss_fresh != 0u ? 1u : hdr->ss_stride;  // (a drawn table is read entry by entry)
    {
        uint32_t sp[8];
#pragma unroll
        for (int k = 0; k < 8; k++)  // (stride: 2 when a blind frame takes every second entry of the kept table)
            sp[k] = tid + k * kT < B - 1 ? split_src[(uint32_t)(tid + k * kT + 1) * stride - 1u] : 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (tid + k * kT < B) {
                s_split[tid + k * kT] = sp[k];
                s_hist[tid + k * kT] = 0u;
            }
    }
    __syncthreads();
    SS_STAMP(dbg, 9);
    // ---- the walk over this workgroup's blocks [first, last), at most 1024 at a time (their offsets sit in LDS): one
    // wave per block of 256 Gaussians, no workgroup barrier inside.  A wave requests the keys of its next kWalk blocks
    // in one go (4 kWalk loads in flight per lane) and only then ranks them: a wave is a chain of HBM round trips
    // otherwise.  Blocks without a visible Gaussian are skipped on their count.
    uint32_t chunk_before = before;
    for (int c0 = first; c0 < last; c0 += 4 * kT) {
        const int nblk = min(4 * kT, last - c0);
        uint32_t tot;
        {
            uint32_t c[4], sum = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = tid * 4 + k;
                c[k] = j < nblk ? block_counts[c0 + j] : 0u;
                sum += c[k];
            }
            uint32_t run = gsr_block_incl_scan(sum, s_w, tot) - sum;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = tid * 4 + k;
                if (j <= nblk) s_boff[j] = run;
                run += c[k];
            }
            if (tid == kT - 1 && nblk == 4 * kT) s_boff[nblk] = run;
        }
        __syncthreads();
        SS_STAMP(dbg, 4);
        {
            // a wave per block: the block's records sit compacted at the head of its own 256 slots (preprocess), so a
            // block is one 8-byte load per lane for up to 64 records; kWalk blocks are requested together
            constexpr int NW = kT / GSR_WAVE, kWalk = 8;
            for (int k0 = wave; k0 < nblk; k0 += NW * kWalk) {
                uint2 rec[kWalk];
                uint32_t cnt[kWalk], pos[kWalk];
#pragma unroll
                for (int w = 0; w < kWalk; w++) {
                    const int k = k0 + w * NW;
                    pos[w] = k < nblk ? s_boff[k] : 0u;
                    cnt[w] = k < nblk ? s_boff[k + 1] - pos[w] : 0u;
                    rec[w] = (uint32_t)lane < cnt[w] ? block_recs[(size_t)(c0 + k) * GSR_BLOCK + lane] : make_uint2(0u, 0u);
                }
#pragma unroll
                for (int w = 0; w < kWalk; w++) {
                    if ((uint32_t)lane < cnt[w]) pairs[chunk_before + pos[w] + (uint32_t)lane] = rec[w];
                    // (a block with more than 64 visible Gaussians: the rest, 64 at a time)
                    for (uint32_t j = 64u + (uint32_t)lane; j < cnt[w]; j += 64u)
                        pairs[chunk_before + pos[w] + j] = block_recs[(size_t)(c0 + k0 + w * NW) * GSR_BLOCK + j];
                }
            }
        }
        __syncthreads();  // (the offsets are rewritten by the next chunk)
        chunk_before += tot;
    }
    SS_STAMP(dbg, 5);
    __syncthreads();  // this workgroup's records are written: visible to all of its threads
    // ---- classification, dense: every thread takes eight records of the segment per step (only ~12 % of the lanes of
    // the walk hold a visible Gaussian -- searching there would run one serial search per 64 Gaussians)
    {
        const uint32_t seg0 = before, seg1 = chunk_before;
        for (uint32_t i0 = seg0 + (uint32_t)tid; i0 < seg1; i0 += 8u * kT) {
            uint32_t tk[8], bk[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t i = i0 + (uint32_t)(u * kT);
                tk[u] = i < seg1 ? (pairs[i].y & kKeyMask) : 0u;
            }
            ss_bucketN<8>(s_split, B, tk, bk);
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (i0 + (uint32_t)(u * kT) < seg1) atomicAdd(&s_hist[bk[u]], 1u);
        }
    }
    SS_STAMP(dbg, 6);
    __syncthreads();
    for (int i = tid; i < B; i += kT) table[(size_t)blockIdx.x * bmax + i] = s_hist[i];
    SS_STAMP(dbg, 7);
#ifdef GSR_SS_TIMING
    if (tid == 0) {
        const unsigned long long el = __builtin_amdgcn_s_memtime() - t_start;
        const unsigned long long old = atomicMax((unsigned long long *)&dbg[10], el);
        if (el > old) { dbg[11] = blockIdx.x; dbg[12] = (unsigned long long)(last - first); dbg[13] = chunk_before - before; }
        if (blockIdx.x == 0) { dbg[14] = el; }
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------
// ss_colscan: exclusive running sum of every bucket's column of the histogram rows (in place) and the column totals.
// One workgroup of 16 waves per 64 buckets, lane = bucket, every wave a contiguous share of the rows.  (The partition
// pass used to sum the rows before its own by itself: O(workgroups x buckets) loads per workgroup -- 17 us of its time
// at config 2, 87 us at 883 k visible Gaussians.)
// ---------------------------------------------------------------------------------------------------------
constexpr int kColT = 1024;  // 16 waves: 256 rows are one batch of 16 loads per lane and pass

__device__ __forceinline__ void ss_colscan_body(int bmax, int nbc, uint32_t *__restrict__ table,
                                                           uint32_t *__restrict__ totals,
                                                           const GsrHeader *__restrict__ hdr) {
    constexpr int NWV = kColT / GSR_WAVE;
    __shared__ uint32_t s_sum[NWV][GSR_WAVE];
    const int lane = gsr_lane(), wave = (int)(threadIdx.x >> 6);
    const uint32_t V = hdr->V;
    if (V == 0u) return;
    const int B = (int)hdr->ss_B;
    const int b = (int)blockIdx.x * GSR_WAVE + lane;
    if ((int)blockIdx.x * GSR_WAVE >= B) return;
    const int q = (nbc + NWV - 1) / NWV, r0 = min(nbc, wave * q), r1 = min(nbc, r0 + q);
    constexpr int kB = 16;
    uint32_t sum = 0;
    for (int r = r0; r < r1; r += kB) {
        uint32_t v[kB];
#pragma unroll
        for (int u = 0; u < kB; u++) v[u] = r + u < r1 ? table[(size_t)(r + u) * bmax + b] : 0u;
#pragma unroll
        for (int u = 0; u < kB; u++) sum += v[u];
    }
    s_sum[wave][lane] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NWV; w++) {
        const uint32_t x = s_sum[w][lane];
        if (w < wave) run += x;
        total += x;
    }
    for (int r = r0; r < r1; r += kB) {
        uint32_t v[kB];
#pragma unroll
        for (int u = 0; u < kB; u++) v[u] = r + u < r1 ? table[(size_t)(r + u) * bmax + b] : 0u;
#pragma unroll
        for (int u = 0; u < kB; u++) {
            if (r + u < r1) table[(size_t)(r + u) * bmax + b] = run;
            run += v[u];
        }
    }
    if (wave == 0) totals[b] = total;
}

// ---------------------------------------------------------------------------------------------------------
// ss_partition: bucket starts from the histogram rows, then the stable move of this workgroup's segment.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ss_partition_body(int bmax, const uint2 *__restrict__ in,
                                                          uint2 *__restrict__ out, const uint32_t *__restrict__ table,
                                                          const uint32_t *__restrict__ totals,
                                                          const uint32_t *__restrict__ splitters,
                                                          const uint32_t *__restrict__ splitters_new,
                                                          const uint32_t *__restrict__ seg_off,
                                                          uint32_t *__restrict__ bucket_start,
                                                          GsrHeader *__restrict__ hdr, uint64_t *__restrict__ dbg0) {
    extern __shared__ uint32_t smem[];
    __shared__ uint32_t s_w[4];
    uint64_t *dbg = dbg0 + 16; const unsigned dbg_wg = 64; (void)dbg_wg; (void)dbg;
    SS_STAMP(dbg, 0);
    uint32_t *s_split = smem;            // [bmax]
    uint32_t *s_run = s_split + bmax;    // [bmax]  next free slot of every bucket for this workgroup
    uint32_t *s_cnt = s_run + bmax;      // [4][bmax]
    const int tid = (int)threadIdx.x, wave = gsr_wave(), lane = gsr_lane();
    const uint32_t V = hdr->V;
    // the table this frame's compaction classified with: the kept one, or the one it drew (ss_compact_kernel)
    const uint32_t *__restrict__ split_src = hdr->ss_fresh != 0u ? splitters_new : splitters;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // The balance flag of the last frame was read by every compaction workgroup (all of them are done: this kernel
        // follows theirs); ss_buckets sets it again.  Trust in the kept table grows by one with every frame that
        // classified WITH it and came out balanced, and is gone with the first one that did not (or that drew its own
        // splitters, which says nothing about the kept ones).
        const uint32_t was_bad = hdr->ss_bad, trust = hdr->ss_trust;
        // (a state no frame has sorted on yet holds garbage here: no kept table, no trust)
        hdr->ss_trust = (hdr->ss_magic == kSplitMagic && was_bad == 0u && hdr->ss_prev_fresh == 0u)
                            ? (trust < 255u ? trust + 1u : 255u) : 0u;
        hdr->ss_prev_fresh = hdr->ss_fresh;
        hdr->ss_bad = 0u;
    }
    if (V == 0u) return;
    const int B = (int)hdr->ss_B, nbits = ss_log2(B), PER = B / kT;  // 1, 2, 4 or 8 buckets per thread
    const uint32_t stride = hdr->ss_fresh != 0u ? 1u : hdr->ss_stride;
    const int me = (int)blockIdx.x;
    (void)lane;
    {
        // bucket starts from the column totals, my first slot per bucket from my (prefixed) histogram row: ss_colscan
        uint32_t sum = 0, T[8], M[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            T[k] = k < PER ? totals[tid * PER + k] : 0u;
            M[k] = k < PER ? table[(size_t)me * bmax + tid * PER + k] : 0u;
            sum += T[k];
        }
        uint32_t all;
        uint32_t run = gsr_block_incl_scan(sum, s_w, all) - sum;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k < PER) {
                const int d = tid * PER + k;
                s_run[d] = run + M[k];
                s_split[d] = d < B - 1 ? split_src[(uint32_t)(d + 1) * stride - 1u] : 0xFFFFFFFFu;
                if (me == 0) bucket_start[d] = run;
                run += T[k];
            }
        if (me == 0 && tid == 0) bucket_start[B] = V;
    }
    SS_STAMP(dbg, 1);
    const uint32_t s0 = seg_off[me], s1 = seg_off[me + 1];
    const uint64_t lt = gsr_lanemask_lt();
    constexpr int kPR = 8;  // rounds per wave and tile: a tile is 4 x kPR x 64 = 2048 records (most segments: one tile)
    for (uint32_t tile = s0; tile < s1; tile += (uint32_t)(4 * kPR * GSR_WAVE)) {  // (a full tile: rounds = kPR)
        for (int i = tid; i < 4 * B; i += kT) s_cnt[(i >> nbits) * bmax + (i & (B - 1))] = 0u;
        __syncthreads();  // (also: s_run / s_split of the set-up above, cursors of the previous tile)
        // the tile's records are dealt to the four waves in contiguous quarters of `rounds` x 64 -- as many rounds as the
        // tile needs, not always kPR: a compaction workgroup's segment is ~700 records at config 2, which used to be eight
        // ranking rounds on wave 0, three on wave 1 and none on the others; now three on each (order: wave, round, lane --
        // the index order, as before)
        const uint32_t left = s1 - tile;
        const int rounds = (int)min((uint32_t)kPR, (left + 4u * GSR_WAVE - 1u) / (4u * GSR_WAVE));
        const uint32_t wbase = tile + (uint32_t)(wave * rounds * GSR_WAVE);
        const uint32_t tend = min(s1, tile + (uint32_t)(4 * rounds * GSR_WAVE));
        uint32_t idx[kPR], key[kPR], dig[kPR], tk[kPR];
#pragma unroll
        for (int r = 0; r < kPR; r++) {
            const uint32_t i = wbase + (uint32_t)(r * GSR_WAVE + lane);
            const uint2 rec = (r < rounds && i < tend) ? in[i] : make_uint2(0u, 0u);
            idx[r] = rec.x;
            key[r] = rec.y;
            tk[r] = rec.y & kKeyMask;
        }
        ss_bucketN<kPR>(s_split, B, tk, dig);
#pragma unroll
        for (int r = 0; r < kPR; r++) {
            const uint32_t i = wbase + (uint32_t)(r * GSR_WAVE + lane);
            if (r < rounds && i < tend) atomicAdd(&s_cnt[wave * bmax + (int)dig[r]], 1u);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k < PER) {
                const int d = tid * PER + k;
                const uint32_t c0 = s_cnt[d], c1 = s_cnt[bmax + d], c2 = s_cnt[2 * bmax + d], c3 = s_cnt[3 * bmax + d];
                const uint32_t start = s_run[d];
                s_cnt[d] = start;
                s_cnt[bmax + d] = start + c0;
                s_cnt[2 * bmax + d] = start + c0 + c1;
                s_cnt[3 * bmax + d] = start + c0 + c1 + c2;
                s_run[d] = start + c0 + c1 + c2 + c3;
            }
        __syncthreads();
        uint32_t *cur = s_cnt + wave * bmax;
#pragma unroll
        for (int r = 0; r < kPR; r++) {
            if (r >= rounds) break;
            const uint32_t i = wbase + (uint32_t)(r * GSR_WAVE + lane);
            const bool valid = i < tend;
            const uint64_t same = ss_match(dig[r], nbits, valid);
            const uint32_t rank = (uint32_t)__popcll(same & lt);
            if (valid) out[cur[dig[r]] + rank] = make_uint2(idx[r], key[r]);
            __builtin_amdgcn_wave_barrier();
            if (valid && rank == 0u) cur[dig[r]] += (uint32_t)__popcll(same);
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
    }
    SS_STAMP(dbg, 2);
}

// ---------------------------------------------------------------------------------------------------------
// ss_buckets: one workgroup per bucket.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ss_buckets_body(int bmax, uint2 *__restrict__ recs, uint2 *__restrict__ scratch,
                                                        const uint32_t *__restrict__ bucket_start,
                                                        uint32_t *__restrict__ order, uint32_t *__restrict__ splitters,
                                                        const uint2 *__restrict__ rects, uint2 *__restrict__ rect_sorted,
                                                        uint32_t *__restrict__ tile_cum, uint32_t *__restrict__ bucket_tiles,
                                                        GsrHeader *__restrict__ hdr, uint64_t *__restrict__ dbg0,
                                                        const float *__restrict__ view, uint32_t sig, int sshift,
                                                        const int32_t *__restrict__ orig, const int bucket) {
    extern __shared__ uint32_t smem[];
    uint64_t *dbg = dbg0 + 32; const unsigned dbg_wg = 100; (void)dbg_wg; (void)dbg;
    SS_STAMP(dbg, 0);
#ifdef GSR_SS_TIMING
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    auto note_time = [&](int nrec) {  // the slowest bucket workgroup of the frame (slots 48..50) and the sum of all (51, 52)
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long el = __builtin_amdgcn_s_memtime() - t_start;
            const unsigned long long old = atomicMax((unsigned long long *)&dbg0[48], el);
            if (el > old) { dbg0[49] = (uint64_t)nrec; dbg0[50] = (uint64_t)bucket; }
            atomicAdd((unsigned long long *)&dbg0[51], el);
            atomicAdd((unsigned long long *)&dbg0[52], 1ull);
        }
    };
#else
    auto note_time = [&](int) {};
#endif
    uint32_t *s_k = smem;                      // [2][kBucketCap]
    uint32_t *s_v = s_k + 2 * kBucketCap;      // [2][kBucketCap]
    uint32_t *s_cur = s_v + 2 * kBucketCap;    // [4][256]
    __shared__ uint32_t s_w[4], s_d[4];
    const int tid = (int)threadIdx.x;
    const uint32_t V = hdr->V;
    if (V == 0u) return;
    const int B = (int)hdr->ss_B;
    if (bucket >= B) return;
    if (bucket == 0 && tid == 0) {
        hdr->ss_magic = kSplitMagic;
        hdr->ss_buckets = (uint32_t)B;
        hdr->ss_P = sig;
    }
    if (bucket == 0 && tid < 16) hdr->ss_view[tid] = __float_as_uint(view[tid]);
    const uint32_t s = bucket_start[bucket];
    const int n = (int)(bucket_start[bucket + 1] - s);
    // above what the exact quantiles of the last frame give when nothing moved (share V / B, plus depth ties): the scene
    // is changing under the camera, the next frames check the kept table against samples (ss_compact_kernel)
    // (a frame whose view is not the table's -- hdr->ss_moved, ss_prepare -- is held to twice its share instead)
    if (tid == 0 && (uint32_t)n > ss_share_limit(V, B, hdr->ss_moved != 0u)) hdr->ss_bad = 1u;
    if (n == 0) {
        if (tid == 0) bucket_tiles[bucket] = 0u;
        return;
    }
    uint2 *seg = recs + s;
    // ---- what leaves the kernel for records [abs, abs + cnt) of the depth order, `at(i)` = (index, key) of the i-th
    // of them: depth order, the tile rects in that order (what the placement streams), the running sum of tiles touched
    // inside the bucket (the placement cuts the depth order into shares of equal INSTANCE count with it), and next
    // frame's splitters -- the exact B-quantiles of this frame's depth order, each written by whoever holds its rank
    auto emit = [&](auto at, uint32_t abs, int cnt, uint32_t carry) -> uint32_t {
        // Quantile i is rank q(i) = (i + 1) V / B, B a power of two.  Until round 6 every thread walked all B - 1 of them
        // with a 64-bit division each -- 5 k of a bucket workgroup's ~25 k cycles by the stamps -- to find the one or two
        // that fall into its records; now the walk starts two short of a float estimate of the first that can (q is
        // monotone; the estimate is off by far less than one) and ends at the first beyond.
        {
            const int lg = ss_log2(B);
            const float est = (float)abs * ((float)B / (float)V);
            const int i0 = max((int)est - 2, 0);
            for (int i = i0 + tid; i < B - 1; i += kT) {
                const uint32_t q = (uint32_t)(((uint64_t)(i + 1) * V) >> lg);
                if (q >= abs + (uint32_t)cnt) break;
                if (q >= abs) splitters[i] = at((int)(q - abs)).y & kKeyMask;
            }
        }
        SS_STAMP(dbg, 5);
        // (every rect gather of the thread goes out before the first is used: kBucketCap / kT per thread and chunk; a
        //  piece of equal keys that went through the global-memory network can be longer than one chunk)
        constexpr int kPerE = kBucketCap / kT;
        for (int c0 = 0; c0 < cnt; c0 += kBucketCap) {
            uint32_t gi[kPerE];
            uint2 rc[kPerE];
#pragma unroll
            for (int e = 0; e < kPerE; e++) {
                const int i = c0 + e * kT + tid;
                gi[e] = i < cnt ? at(i).x : 0u;
#if GSR_SS_DIAG == 1  // (timing diagnostics only -- wrong rects: the gather as a coalesced read)
                rc[e] = i < cnt ? rects[abs + (uint32_t)i] : make_uint2(0u, 0u);
#else
                rc[e] = i < cnt ? rects[gi[e]] : make_uint2(0u, 0u);
#endif
            }
#pragma unroll
            for (int e = 0; e < kPerE; e++) {
                if (c0 + e * kT >= cnt) break;
                const int i = c0 + e * kT + tid;
                uint32_t t = 0;
                if (i < cnt) {
                    const uint2 r2 = ss_super_rect(rc[e], sshift);
                    order[abs + i] = gi[e];
                    rect_sorted[abs + i] = r2;
                    t = ((r2.y & 0xffffu) - (r2.x & 0xffffu)) * ((r2.y >> 16) - (r2.x >> 16)) + kRankCost;
                }
                uint32_t tot;
                const uint32_t incl = gsr_block_incl_scan(t, s_w, tot);
                if (i < cnt) tile_cum[abs + i] = carry + incl;
                carry += tot;
                if (e == 0 && c0 == 0) SS_STAMP(dbg, 6);
            }
        }
        return carry;
    };
    // ---- cnt <= kBucketCap records from global memory into the LDS, sorted there by key (stable: LSD passes over the
    // bits in which the keys differ); `by_index` first sorts them by index the same way, for records that did not
    // arrive in index order.  -> which half of s_k / s_v holds the result
    // The keys go to the LDS as OFFSETS from the smallest key of the piece (kbase): the passes then cover the bits of the
    // piece's RANGE -- 14-16 for a bucket that holds 1 / 512 of the depth order -- not the bits in which any two of its keys
    // differ (17-20 by the stamps: two keys either side of a power of two differ in every bit below it): two 8-bit passes
    // where rounds 2-4 ran three.
    uint32_t kbase = 0u;
    __shared__ uint32_t s_e[4];
    auto sort_in_lds = [&](const uint2 *from, int cnt, bool by_index) -> int {
        constexpr int kPer = kBucketCap / kT;
        uint2 r[kPer];
        uint32_t kmn = 0xFFFFFFFFu, kmx = 0u, vdiff = 0;
        const uint32_t v0 = from[0].x;
#pragma unroll
        for (int e = 0; e < kPer; e++) {
            const int i = tid + e * kT;
            r[e] = i < cnt ? from[i] : make_uint2(v0, 0xFFFFFFFFu);
            if (i < cnt) {
                kmn = min(kmn, r[e].y);
                kmx = max(kmx, r[e].y);
                vdiff |= r[e].x ^ v0;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            kmn = min(kmn, (uint32_t)__shfl_xor((int)kmn, o, 64));
            kmx = max(kmx, (uint32_t)__shfl_xor((int)kmx, o, 64));
            vdiff |= (uint32_t)__shfl_xor((int)vdiff, o, 64);
        }
        __syncthreads();  // (s_w may still be read by the scan of an earlier emit)
        if (gsr_lane() == 0) {
            s_w[gsr_wave()] = kmn;
            s_e[gsr_wave()] = kmx;
            s_d[gsr_wave()] = vdiff;
        }
        __syncthreads();
        kmn = min(min(s_w[0], s_w[1]), min(s_w[2], s_w[3]));
        kmx = max(max(s_e[0], s_e[1]), max(s_e[2], s_e[3]));
        vdiff = s_d[0] | s_d[1] | s_d[2] | s_d[3];
        kbase = kmn;
#pragma unroll
        for (int e = 0; e < kPer; e++) {
            const int i = tid + e * kT;
            if (i < cnt) {
                s_v[i] = r[e].x;
                s_k[i] = r[e].y - kmn;
            }
        }
        __syncthreads();
        int src = 0;
        if (by_index) {
            const int vbits = vdiff == 0u ? 0 : 32 - __builtin_clz(vdiff);
            for (int shift = 0; shift < vbits; shift += 8) {
                lds_radix_pass<true>(s_v + src * kBucketCap, s_k + src * kBucketCap, s_v + (src ^ 1) * kBucketCap,
                                     s_k + (src ^ 1) * kBucketCap, cnt, shift, s_cur, s_w);
                src ^= 1;
            }
        }
        const uint32_t range = kmx - kmn;
        const int bits = range == 0u ? 0 : 32 - __builtin_clz(range);
#ifdef GSR_SS_TIMING
        if (blockIdx.x == dbg_wg && tid == 0) { dbg[8] = (uint64_t)bits; dbg[9] = (uint64_t)cnt; }
#endif
        for (int shift = 0; shift < bits; shift += 8) {
            lds_radix_pass<true>(s_k + src * kBucketCap, s_v + src * kBucketCap, s_k + (src ^ 1) * kBucketCap,
                                 s_v + (src ^ 1) * kBucketCap, cnt, shift, s_cur, s_w);
            src ^= 1;
        }
        return src;
    };
    // ---- a permuted model (GsrInputs.orig_index): the records carry positions in the permuted arrays and the stable sort
    // left equal keys in that order; the reference breaks depth ties by the ORIGINAL number.  Ties are rare (~1 300 pairs
    // among the 175 k visible Gaussians of configs[1], but most buckets hold one), so instead of sorting every bucket by
    // original number first, the members of every run of equal keys are ranked among themselves: their original numbers
    // go to the free half of the LDS, every member counts the smaller ones of its run and moves to that place.
    auto fix_ties = [&](int src, int cnt) {
        if (orig == nullptr) return;
        uint32_t *kk = s_k + src * kBucketCap, *vv = s_v + src * kBucketCap, *oo = s_v + (src ^ 1) * kBucketCap;
        constexpr int kPer = kBucketCap / kT;
        // pair (i, i + 1) holds equal keys: fixed for the whole fix-up (keys do not move).  A run of EXACTLY two equal keys --
        // nearly every tie there is -- is put in order by the thread of its first member on the spot: it fetches both
        // original numbers itself and nobody else touches the two slots.  Only members of runs of three and more stage
        // their original numbers for the rounds below.  (Round 4 ran the rounds for every tie: three barriers and the
        // staging pass in nearly every bucket, 4.6 k of a bucket workgroup's ~25 k cycles by the stamps.)
        bool eq[kPer], two[kPer];
        uint32_t pa[kPer], pb[kPer];
        int longrun = 0;
#pragma unroll
        for (int e = 0; e < kPer; e++) {
            const int i = tid + e * kT;
            eq[e] = i + 1 < cnt && kk[i] == kk[i + 1];
            const bool prev = i > 0 && i < cnt && kk[i - 1] == kk[i];
            const bool next2 = eq[e] && i + 2 < cnt && kk[i + 1] == kk[i + 2];
            two[e] = eq[e] && !prev && !next2;
            // member of a run of three or more: as a pair's first element, or as the last element of such a run
            const bool in_long = (eq[e] && !two[e]) || (prev && !eq[e] && i >= 2 && kk[i - 2] == kk[i - 1]);
            pa[e] = (two[e] || in_long) ? (uint32_t)orig[vv[i]] : 0u;
            pb[e] = two[e] ? (uint32_t)orig[vv[i + 1]] : 0u;
            if (in_long) oo[i] = pa[e];
            longrun |= in_long ? 1 : 0;
        }
#pragma unroll
        for (int e = 0; e < kPer; e++) {
            const int i = tid + e * kT;
            if (two[e] && pa[e] > pb[e]) {
                const uint32_t v0 = vv[i], v1 = vv[i + 1];
                vv[i] = v1;
                vv[i + 1] = v0;
            }
        }
        if (__syncthreads_or(longrun) == 0) return;  // (no run of three or more in this bucket; the swaps are visible)
        // Odd-even transposition restricted to the pairs of the long runs: in a round the pairs that start at even (odd)
        // positions are disjoint, each is put in order by one thread, a barrier ends the round; done when an even and an odd
        // round in a row moved nothing.  A run of L equal keys takes at most L rounds; thousands of equal depths (a plane
        // facing the camera) cost a barrier each and stay exact.  (Measured against it in round 4: ranking every member
        // inside a +-6 window with straight-line code and falling back to the rounds for longer runs -- slower on both
        // views: the window is probed whether or not a run is long.  Round 6, after the stamps showed fix-ups of 10-38 k
        // cycles in the slowest bucket workgroup of a closed-loop frame: every member of a run of up to 192 finding the ends
        // of its run and counting the smaller original numbers, two barriers per bucket -- ss_buckets 21.3 -> 23.8 us: the
        // usual run is three to six members, for which a handful of rounds is the shorter chain.)
        int quiet = 0;
        for (int round = 0; quiet < 2; round++) {
            int moved = 0;
#pragma unroll
            for (int e = 0; e < kPer; e++) {
                const int i = tid + e * kT;
                if (eq[e] && !two[e] && ((i ^ round) & 1) == 0) {
                    const uint32_t o0 = oo[i], o1 = oo[i + 1];
                    if (o0 > o1) {
                        const uint32_t v0 = vv[i], v1 = vv[i + 1];
                        oo[i] = o1; oo[i + 1] = o0;
                        vv[i] = v1; vv[i + 1] = v0;
                        moved = 1;
                    }
                }
            }
            quiet = __syncthreads_or(moved) != 0 ? 0 : quiet + 1;
        }
    };
    if (n > kBucketCap) {
        // Does not fit the LDS (the kept splitters were taken unchecked and the scene had moved; a sample check that
        // missed; depth ties).  The bucket is cut once more, by this workgroup alone: sub-splitters from 1024 of its own
        // keys, an unordered scatter into the record buffer the partition pass has finished with (`scratch`, same
        // offsets), then every piece is sorted in the LDS -- by index first, since the scatter lost the index order
        // the stable sort by key relies on.  A piece that still does not fit (more than kBucketCap records between two
        // sub-splitters: ties) goes through a bitonic network over its (key << 32 | index) composites in global memory.
        constexpr int kSub = 64, kSubSamples = 1024;
        __shared__ uint32_t s_sub[kSub], s_cnt[kSub], s_off[kSub + 1], s_fill[kSub];
        uint2 *tmp = scratch + s;
        const int m = min(kSub, (n + kBucketCap / 2 - 1) / (kBucketCap / 2));
        for (int i = tid; i < kSubSamples; i += kT) s_k[i] = seg[(int)(((int64_t)i * n) / kSubSamples)].y;
        if (tid < kSub) s_cnt[tid] = 0u;
        __syncthreads();
        lds_radix_pass<false>(s_k, nullptr, s_k + kBucketCap, nullptr, kSubSamples, 0, s_cur, s_w);
        lds_radix_pass<false>(s_k + kBucketCap, nullptr, s_k, nullptr, kSubSamples, 8, s_cur, s_w);
        lds_radix_pass<false>(s_k, nullptr, s_k + kBucketCap, nullptr, kSubSamples, 16, s_cur, s_w);
        lds_radix_pass<false>(s_k + kBucketCap, nullptr, s_k, nullptr, kSubSamples, 24, s_cur, s_w);
        if (tid < m - 1) s_sub[tid] = s_k[((tid + 1) * kSubSamples) / m];
        __syncthreads();
        // piece of a key = number of sub-splitters <= key: equal keys stay together
        auto piece = [&](uint32_t key) -> int {
            int lo = 0, hi = m - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s_sub[mid] <= key) lo = mid + 1; else hi = mid;
            }
            return lo;
        };
        for (int i = tid; i < n; i += kT) atomicAdd(&s_cnt[piece(seg[i].y)], 1u);
        __syncthreads();
        if (tid == 0) {
            uint32_t run = 0;
            for (int j = 0; j < m; j++) {
                s_off[j] = run;
                s_fill[j] = run;
                run += s_cnt[j];
            }
            s_off[m] = run;
        }
        __syncthreads();
        for (int i = tid; i < n; i += kT) {
            const uint2 r = seg[i];
            tmp[atomicAdd(&s_fill[piece(r.y)], 1u)] = r;
        }
        __syncthreads();  // (this workgroup's stores to `tmp` are complete and visible to all of its waves)
        uint32_t carry = 0;
        for (int j = 0; j < m; j++) {
            const uint32_t o = s_off[j];
            const int nj = (int)(s_off[j + 1] - o);
            if (nj == 0) continue;
            if (nj <= kBucketCap) {
                const int src = sort_in_lds(tmp + o, nj, true);
                fix_ties(src, nj);
                const uint32_t *kk = s_k + src * kBucketCap, *vv = s_v + src * kBucketCap;
                carry = emit([&](int i) { return make_uint2(vv[i], kk[i] + kbase); }, s + o, nj, carry);
            } else {
                int N = 2;
                while (N < nj) N <<= 1;
                uint64_t *comp = reinterpret_cast<uint64_t *>(tmp + o);
                if (orig != nullptr) {
                    // (permuted model: order by (key, original number); the positions travel as payload in the
                    // records' first home, which the scatter above has finished with)
                    uint32_t *pos = reinterpret_cast<uint32_t *>(seg + o);
                    for (int i = tid; i < nj; i += kT) {
                        const uint2 r = tmp[o + i];
                        pos[i] = r.x;
                        tmp[o + i] = make_uint2((uint32_t)orig[r.x], r.y);
                    }
                    __syncthreads();
                    bitonic_sort_block(comp, nj, N, pos);
                    carry = emit([&](int i) { return make_uint2(pos[i], (uint32_t)(comp[i] >> 32)); }, s + o, nj, carry);
                } else {
                    bitonic_sort_block(comp, nj, N);
                    carry = emit([&](int i) { const uint64_t c = comp[i]; return make_uint2((uint32_t)c, (uint32_t)(c >> 32)); },
                                 s + o, nj, carry);
                }
            }
            __syncthreads();  // the LDS halves are free for the next piece
        }
        if (tid == 0) bucket_tiles[bucket] = carry;
        note_time(n);
        return;
    }
    SS_STAMP(dbg, 1);
    const int src = sort_in_lds(seg, n, false);
    SS_STAMP(dbg, 3);
#ifdef GSR_SS_TIMING
    const unsigned long long t_fix0 = __builtin_amdgcn_s_memtime();
#endif
    fix_ties(src, n);
#ifdef GSR_SS_TIMING
    __syncthreads();
    if (tid == 0) {
        const unsigned long long el = __builtin_amdgcn_s_memtime() - t_fix0;
        const unsigned long long old = atomicMax((unsigned long long *)&dbg0[53], el);
        if (el > old) dbg0[54] = (uint64_t)n;
    }
#endif
    const uint32_t *kk = s_k + src * kBucketCap, *vv = s_v + src * kBucketCap;
    const uint32_t carry = emit([&](int i) { return make_uint2(vv[i], kk[i] + kbase); }, s, n, 0u);
    if (tid == 0) bucket_tiles[bucket] = carry;
    SS_STAMP(dbg, 4);
    note_time(n);
}

// ---------------------------------------------------------------------------------------------------------
// ss_buckets, a WAVE per bucket (round 6).  A frame that samples cuts the depth order into buckets of <= 512 records on
// average (290 at configs[2]'s closed loop): for a workgroup of four waves that is one record per thread and some twenty
// workgroup barriers -- two LSD passes of (zero, count, scan, rank), the tie fix-up, the emit's scans -- on a launch that
// does not fill the chip: the kernel lasted as long as one such chain (21 us for the two frames of a closed-loop step;
// 1 024 workgroups of 862 records in the dense view: 38 us, "four workgroups per CU stand in each other's way").  Here a
// wave takes a bucket by itself: its records live in wave-private LDS (lane l, round r = record 64 r + l), the LSD passes
// rank with wave-wide match masks, scans are DPP scans, and nothing waits at a workgroup barrier; four buckets per
// workgroup.  Buckets above kWaveCap records (an unlucky sampled table, depth ties, a scene that jumped) are left to the
// workgroup path, which the four waves then run together, one such bucket after the other.  Same order, same outputs.
// ---------------------------------------------------------------------------------------------------------
constexpr int kWaveCap = 512;
constexpr int kWaveLds = 4 * kWaveCap + 256;  // words per wave: keys and values, two halves each, + 256 digit cursors

// one stable LSD pass (8-bit digit at `shift`) over the wave's n records
__device__ __forceinline__ void ss_wave_radix_pass(const uint32_t *kin, const uint32_t *vin, uint32_t *kout, uint32_t *vout,
                                                   int n, int shift, uint32_t *cur) {
    const int lane = gsr_lane();
    const uint64_t lt = gsr_lanemask_lt();
    const int rounds = (n + GSR_WAVE - 1) >> 6;
#pragma unroll
    for (int k = 0; k < 4; k++) cur[4 * lane + k] = 0u;
    __builtin_amdgcn_wave_barrier();
    for (int r = 0; r < rounds; r++) {
        const int i = (r << 6) + lane;
        if (i < n) atomicAdd(&cur[(kin[i] >> shift) & 255u], 1u);
    }
    __builtin_amdgcn_wave_barrier();
    {
        const uint32_t c0 = cur[4 * lane], c1 = cur[4 * lane + 1], c2 = cur[4 * lane + 2], c3 = cur[4 * lane + 3];
        const uint32_t sum = c0 + c1 + c2 + c3;
        const uint32_t excl = gsr_wave_incl_scan(sum) - sum;
        cur[4 * lane] = excl;
        cur[4 * lane + 1] = excl + c0;
        cur[4 * lane + 2] = excl + c0 + c1;
        cur[4 * lane + 3] = excl + c0 + c1 + c2;
    }
    __builtin_amdgcn_wave_barrier();
    for (int r = 0; r < rounds; r++) {
        const int i = (r << 6) + lane;
        const bool valid = i < n;
        const uint32_t key = valid ? kin[i] : 0u;
        const uint32_t val = valid ? vin[i] : 0u;
        const uint32_t d = (key >> shift) & 255u;
        const uint64_t same = ss_match(d, 8, valid);
        const uint32_t rank = (uint32_t)__popcll(same & lt);
        if (valid) {
            const uint32_t pos = cur[d] + rank;
            kout[pos] = key;
            vout[pos] = val;
        }
        __builtin_amdgcn_wave_barrier();  // every lane has read its cursor before the group leader moves it
        if (valid && rank == 0u) cur[d] += (uint32_t)__popcll(same);
        __builtin_amdgcn_wave_barrier();
    }
}

// the whole of ss_buckets_body for one bucket of n <= kWaveCap records, by ONE wave; lds: this wave's kWaveLds words
__device__ __forceinline__ void ss_bucket_wave(const int bucket, const int B, const uint32_t V, const uint32_t s, const int n,
                                               const uint2 *__restrict__ recs, uint32_t *__restrict__ order,
                                               uint32_t *__restrict__ splitters, const uint2 *__restrict__ rects,
                                               uint2 *__restrict__ rect_sorted, uint32_t *__restrict__ tile_cum,
                                               uint32_t *__restrict__ bucket_tiles, const int sshift,
                                               const int32_t *__restrict__ orig, uint32_t *lds) {
    const int lane = gsr_lane();
    uint32_t *s_k = lds, *s_v = lds + 2 * kWaveCap, *cur = lds + 4 * kWaveCap;
    if (n == 0) {
        if (lane == 0) bucket_tiles[bucket] = 0u;
        return;
    }
    constexpr int kR = kWaveCap / GSR_WAVE;
    const int rounds = (n + GSR_WAVE - 1) >> 6;
    // ---- the records, every load in flight at once; keys go to the LDS as offsets from the bucket's smallest
    uint2 r[kR];
    uint32_t kmn = 0xFFFFFFFFu, kmx = 0u;
#pragma unroll
    for (int e = 0; e < kR; e++) {
        const int i = (e << 6) + lane;
        r[e] = i < n ? recs[s + (uint32_t)i] : make_uint2(0u, 0xFFFFFFFFu);
        if (i < n) {
            kmn = min(kmn, r[e].y);
            kmx = max(kmx, r[e].y);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        kmn = min(kmn, (uint32_t)__shfl_xor((int)kmn, o, 64));
        kmx = max(kmx, (uint32_t)__shfl_xor((int)kmx, o, 64));
    }
#pragma unroll
    for (int e = 0; e < kR; e++) {
        const int i = (e << 6) + lane;
        if (i < n) {
            s_k[i] = r[e].y - kmn;
            s_v[i] = r[e].x;
        }
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t range = kmx - kmn;
    const int bits = range == 0u ? 0 : 32 - __builtin_clz(range);
    int src = 0;
    for (int shift = 0; shift < bits; shift += 8) {
        ss_wave_radix_pass(s_k + src * kWaveCap, s_v + src * kWaveCap, s_k + (src ^ 1) * kWaveCap, s_v + (src ^ 1) * kWaveCap, n,
                           shift, cur);
        src ^= 1;
    }
    uint32_t *kk = s_k + src * kWaveCap, *vv = s_v + src * kWaveCap;
    // ---- a permuted model: runs of equal depth in the order of the ORIGINAL numbers (ss_buckets_body fix_ties, wave form:
    // odd-even transposition over the pairs of equal keys, a ballot per round instead of a workgroup barrier)
    if (orig != nullptr) {
        uint32_t *oo = s_v + (src ^ 1) * kWaveCap;  // (the free half: original numbers of the run members)
        bool eq[kR];
        int any = 0;
#pragma unroll
        for (int e = 0; e < kR; e++) {
            const int i = (e << 6) + lane;
            eq[e] = e < rounds && i + 1 < n && kk[i] == kk[i + 1];
            const bool prev = e < rounds && i > 0 && i < n && kk[i - 1] == kk[i];
            if (eq[e] || prev) {
                oo[i] = (uint32_t)orig[vv[i]];
                any = 1;
            }
        }
        if (__builtin_amdgcn_ballot_w64(any != 0) != 0ull) {
            __builtin_amdgcn_wave_barrier();
            int quiet = 0;
            for (int round = 0; quiet < 2; round++) {
                int moved = 0;
#pragma unroll
                for (int e = 0; e < kR; e++) {
                    const int i = (e << 6) + lane;
                    if (eq[e] && ((i ^ round) & 1) == 0) {
                        const uint32_t o0 = oo[i], o1 = oo[i + 1];
                        if (o0 > o1) {
                            const uint32_t v0 = vv[i], v1 = vv[i + 1];
                            oo[i] = o1; oo[i + 1] = o0;
                            vv[i] = v1; vv[i + 1] = v0;
                            moved = 1;
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                quiet = __builtin_amdgcn_ballot_w64(moved != 0) != 0ull ? 0 : quiet + 1;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- what leaves the kernel: next frame's splitters (the exact B-quantiles of this frame's order: rank
    // floor((i + 1) V / B) for splitter i, written by whoever holds it), depth order, rects in that order, running cost
    {
        const uint64_t lo = ((uint64_t)s * (uint32_t)B + V - 1u) / V, hi = ((uint64_t)(s + (uint32_t)n) * (uint32_t)B + V - 1u) / V;
        for (uint64_t j = max(lo, (uint64_t)1) + (uint32_t)lane; j < hi && j < (uint64_t)B; j += GSR_WAVE) {
            const uint32_t q = (uint32_t)((j * V) >> ss_log2(B));  // rank of splitter j - 1 (B is a power of two)
            if (q >= s && q < s + (uint32_t)n) splitters[j - 1u] = (kk[q - s] + kmn) & kKeyMask;
        }
    }
    uint32_t gi[kR];
    uint2 rc[kR];
#pragma unroll
    for (int e = 0; e < kR; e++) {
        const int i = (e << 6) + lane;
        gi[e] = (e < rounds && i < n) ? vv[i] : 0u;
        rc[e] = (e < rounds && i < n) ? rects[gi[e]] : make_uint2(0u, 0u);
    }
    uint32_t carry = 0u;
#pragma unroll
    for (int e = 0; e < kR; e++) {
        if (e >= rounds) break;
        const int i = (e << 6) + lane;
        uint32_t t = 0u;
        if (i < n) {
            const uint2 r2 = ss_super_rect(rc[e], sshift);
            order[s + (uint32_t)i] = gi[e];
            rect_sorted[s + (uint32_t)i] = r2;
            t = ((r2.y & 0xffffu) - (r2.x & 0xffffu)) * ((r2.y >> 16) - (r2.x >> 16)) + kRankCost;
        }
        const uint32_t incl = gsr_wave_incl_scan(t);
        if (i < n) tile_cum[s + (uint32_t)i] = carry + incl;
        carry += (uint32_t)__shfl((int)incl, 63, 64);
    }
    if (lane == 0) bucket_tiles[bucket] = carry;
}

// ---- the four kernels: grid = (workgroups of one frame, frames); blockIdx.y picks the frame's argument block ----------
struct SsArgs {
    int P, nb1, bpw, bmax, nbc;
    uint2 *pair0, *pair1;           // compacted records | block-local records, then the bucketed ones
    const uint32_t *block_counts, *block_cand;
    uint32_t *table, *splitters, *splitters_new, *seg, *first, *totals, *bucket_start;
    GsrHeader *hdr;
    uint64_t *dbg;
    const float *view;
    uint32_t sig;
    const uint32_t *quad_work;      // (the prepare launch's second workgroup: the compositor's quadrant deal)
    int num_quads;
    uint32_t *quad_order;
    int cus_per_xcd;
    uint32_t *coop_list;  // cooperative quadrants of the compositor: one entry per cooperative workgroup (coop_cap; 0: none)
    int coop_cap;
    uint32_t *order;
    const uint2 *rects;
    uint2 *rect_sorted;
    uint32_t *tile_cum, *bucket_tiles;
    int sshift;
    const int32_t *orig;
    const uint32_t *tile_dirty;  // (ImageState::tile_dirty: the deal of the compositor's quadrants reads it, bit 1 of pc_enabled)
    const uint2 *block_recs;  // preprocess' block-local records (GeomState::block_recs)
    int pc_enabled;           // this frame's preprocess ran with the block cache: ss_prepare makes its slot current
};

__global__ __launch_bounds__(kPT) void ss_prepare_kernel(const GsrBatch<SsArgs> bt) {
    const SsArgs &a = bt.f[blockIdx.y];
    if (blockIdx.x == 2) {  // (only launched with a deal to make)
        ss_quad_order_1024(a.quad_work, a.num_quads, a.quad_order, a.cus_per_xcd, a.coop_list, a.coop_cap, a.hdr,
                           (a.pc_enabled & 2) != 0 ? a.tile_dirty : (const uint32_t *)nullptr);
        return;
    }
    ss_prepare_body(a.P, a.nb1, a.bmax, a.nbc, a.block_recs, a.block_counts, a.splitters, a.splitters_new, a.seg, a.first,
                    a.hdr, a.dbg, a.view, a.sig, (int)blockIdx.x, (a.pc_enabled & 1) != 0);
}
__global__ __launch_bounds__(kT) void ss_compact_kernel(const GsrBatch<SsArgs> bt) {
    const SsArgs &a = bt.f[blockIdx.y];
    ss_compact_body(a.bmax, a.block_recs, a.block_counts, a.pair0, a.table, a.splitters, a.splitters_new, a.seg, a.first,
                    a.hdr, a.dbg, (a.pc_enabled & 2) != 0);
}
__global__ __launch_bounds__(kColT) void ss_colscan_kernel(const GsrBatch<SsArgs> bt) {
    const SsArgs &a = bt.f[blockIdx.y];
    ss_colscan_body(a.bmax, a.nbc, a.table, a.totals, a.hdr);
}
__global__ __launch_bounds__(kT) void ss_partition_kernel(const GsrBatch<SsArgs> bt) {
    const SsArgs &a = bt.f[blockIdx.y];
    ss_partition_body(a.bmax, a.pair0, a.pair1, a.table, a.totals, a.splitters, a.splitters_new, a.seg, a.bucket_start,
                      a.hdr, a.dbg);
}
#ifndef GSR_SS_WAVE_BUCKETS
#define GSR_SS_WAVE_BUCKETS 0  // MEASURED AND NOT KEPT (1 = on): at configs[2]'s closed loop -- 2 x 512 buckets of ~290
                               // records -- ss_buckets lasted 47 us with a wave per bucket against 21 us with a workgroup
                               // per bucket (closed loop 7.5 against 8.4 k frames/s; every test green either way): a
                               // bucket's ten ranking rounds are a chain of dependent LDS round trips, and four waves
                               // walk a quarter of it each -- the barriers between them were never the price
#endif
__global__ __launch_bounds__(kT) void ss_buckets_kernel(const GsrBatch<SsArgs> bt) {
    const SsArgs &a = bt.f[blockIdx.y];
    const uint32_t V = a.hdr->V;
    if (V == 0u) return;
    const int B = (int)a.hdr->ss_B;
    // a frame that took its kept quantiles blind has buckets of up to 1024 records: a workgroup each, as before; a frame
    // that sampled (buckets of <= 512 on average): a wave each, four buckets per workgroup
    const bool by_wave = GSR_SS_WAVE_BUCKETS && a.hdr->ss_blind == 0u;
    if (!by_wave) {
        ss_buckets_body(a.bmax, a.pair1, a.pair0, a.bucket_start, a.order, a.splitters, a.rects, a.rect_sorted, a.tile_cum,
                        a.bucket_tiles, a.hdr, a.dbg, a.view, a.sig, a.sshift, a.orig, (int)blockIdx.x);
        return;
    }
    constexpr int NW = kT / GSR_WAVE;
    if ((int)blockIdx.x * NW >= B) return;
    extern __shared__ uint32_t smem[];
    static_assert(NW * kWaveLds <= 4 * kBucketCap + 4 * 256, "the waves' LDS fits the workgroup path's");
    __shared__ uint32_t s_big[NW];
    const int tid = (int)threadIdx.x, wave = gsr_wave();
    const int bucket = (int)blockIdx.x * NW + wave;
    if (blockIdx.x == 0 && tid == 0) {
        a.hdr->ss_magic = kSplitMagic;
        a.hdr->ss_buckets = (uint32_t)B;
        a.hdr->ss_P = a.sig;
    }
    if (blockIdx.x == 0 && tid < 16) a.hdr->ss_view[tid] = __float_as_uint(a.view[tid]);
    const uint32_t s0 = a.bucket_start[bucket];
    const int n = (int)(a.bucket_start[bucket + 1] - s0);
    // (above what the exact quantiles of the last frame give when nothing moved: see ss_buckets_body)
    if (gsr_lane() == 0) {
        if ((uint32_t)n > ss_share_limit(V, B, a.hdr->ss_moved != 0u)) a.hdr->ss_bad = 1u;
        s_big[wave] = n > kWaveCap ? 1u : 0u;
    }
    if (n <= kWaveCap)
        ss_bucket_wave(bucket, B, V, s0, n, a.pair1, a.order, a.splitters, a.rects, a.rect_sorted, a.tile_cum, a.bucket_tiles,
                       a.sshift, a.orig, smem + wave * kWaveLds);
    __syncthreads();
    for (int w = 0; w < NW; w++) {  // (rare: the buckets a wave could not hold, by the four waves together)
        if (s_big[w] == 0u) continue;
        ss_buckets_body(a.bmax, a.pair1, a.pair0, a.bucket_start, a.order, a.splitters, a.rects, a.rect_sorted, a.tile_cum,
                        a.bucket_tiles, a.hdr, a.dbg, a.view, a.sig, a.sshift, a.orig, (int)blockIdx.x * NW + w);
        __syncthreads();
    }
}

}  // namespace

// geometry of the sample sort for P Gaussians: compaction workgroups, preprocess blocks per workgroup, bucket capacity
int gsr_ss_nbc(int32_t P) {
    const int nb1 = GeomState::prep_blocks(P);
    int nbc = gsr_div_up(nb1, 8);
    if (nbc > 256) nbc = 256;  // (one per CU; 128 measured: same at config 2, dense view compaction + partition 103 -> 67 us)
    const int need = gsr_div_up(nb1, 1024);  // at most 1024 blocks per workgroup (LDS offsets)
    return nbc > need ? nbc : need;
}
// the sample sort's prepare workgroup keeps every block count of the model in LDS: models beyond that (8.4 M Gaussians)
// take the LSD radix depth sort (api.hip)
bool gsr_ss_supported(int32_t P) { return GeomState::prep_blocks(P) <= kPrepBlocks; }
int gsr_ss_bmax(int32_t P) {
    int b = 256;
    while (b < 2048 && (int64_t)b * 512 < (int64_t)P) b <<= 1;
    return b;
}

// preprocess left the block-local records (pair[1]) / block_counts / block_cand; the sorted depth order ends in g.order
// (order_early: the prepare launch's second workgroup deals the num_quads quadrants of the frame's compositor ->
//  img.quad_order; super_shift: 1 = rect_sorted in super-tile units, GsrSettings.forward_only)
int gsr_launch_sample_depth_sort(int B, const GsrFrame *fr, bool order_early, int num_quads, int super_shift,
                                 int coop_blocks, bool debug, hipStream_t stream) {
    const int32_t P = fr[0].in->P;
    const int nb1 = GeomState::prep_blocks(P);
    const int nbc = gsr_ss_nbc(P), bpw = gsr_div_up(nb1, nbc), bmax = gsr_ss_bmax(P);
    GsrBatch<SsArgs> bt{};  // (entries beyond B stay zero: nothing uninitialised travels in the kernarg)
    for (int k = 0; k < B; k++) {
        const GeomState &g = fr[k].g;
        SsArgs &a = bt.f[k];
        a.P = P; a.nb1 = nb1; a.bpw = bpw; a.bmax = bmax; a.nbc = nbc;
        a.pair0 = g.pair[0]; a.pair1 = g.pair[1];
        a.block_counts = g.block_counts; a.block_cand = g.block_cand;
        a.table = g.ss_table; a.splitters = g.ss_splitters; a.splitters_new = g.ss_splitters_new; a.seg = g.ss_seg;
        a.first = g.ss_first;
        a.totals = g.ss_totals; a.bucket_start = g.ss_bucket_start;
        a.hdr = g.hdr; a.dbg = g.ss_dbg; a.view = fr[k].in->viewmatrix;
        // what a kept splitter table is valid for: this model size AND this state layout (a buffer the allocator hands
        // back can carry a plausible header of another layout over arrays that have moved: a lean inference state after a
        // full one took a zero-filled "table" blind once, and one bucket of 175 k records went through the global-memory
        // sort)
        a.sig = (uint32_t)P * 2654435761u ^ (uint32_t)((char *)g.ss_splitters - (char *)g.hdr);
        a.quad_work = fr[k].img.quad_work;
        a.num_quads = num_quads;
        a.quad_order = fr[k].img.quad_order;
        a.cus_per_xcd = gsr_render_cus_per_xcd();
        a.coop_list = fr[k].img.split_list;  // (the split list, reused)
        a.coop_cap = coop_blocks;
        a.order = g.order; a.rects = g.rects; a.rect_sorted = g.rect_sorted; a.tile_cum = g.tile_cum;
        a.bucket_tiles = g.bucket_tiles; a.sshift = super_shift; a.orig = fr[k].in->orig_index;
        a.block_recs = g.block_recs;
        a.tile_dirty = fr[k].img.tile_dirty;
        a.pc_enabled = (fr[k].pc ? 1 : 0) | (fr[k].pc && fr[k].td ? 2 : 0);
    }
    hipLaunchKernelGGL(ss_prepare_kernel, dim3(order_early ? 3 : 2, B), dim3(kPT), 0, stream, bt);
    if (int e = gsr_check_launch("ss_prepare", debug, stream)) return e;
    const size_t lds1 = (size_t)(2 * bmax + 4 * kT + 1) * sizeof(uint32_t);
    hipLaunchKernelGGL(ss_compact_kernel, dim3(nbc, B), dim3(kT), lds1, stream, bt);
    if (int e = gsr_check_launch("ss_compact", debug, stream)) return e;
    hipLaunchKernelGGL(ss_colscan_kernel, dim3(gsr_div_up(bmax, GSR_WAVE), B), dim3(kColT), 0, stream, bt);
    if (int e = gsr_check_launch("ss_colscan", debug, stream)) return e;
    const size_t lds2 = (size_t)(6 * bmax) * sizeof(uint32_t);
    hipLaunchKernelGGL(ss_partition_kernel, dim3(nbc, B), dim3(kT), lds2, stream, bt);
    if (int e = gsr_check_launch("ss_partition", debug, stream)) return e;
#ifndef GSR_SS_BUCKETS_LDS_KB
#define GSR_SS_BUCKETS_LDS_KB 0  // (A/B: ask for this much LDS per bucket workgroup instead of the 36 KB it needs -- caps the
#endif                           //  workgroups resident per CU: 160 KB / this)
    size_t lds3 = (size_t)(4 * kBucketCap + 4 * 256) * sizeof(uint32_t);
    if ((size_t)GSR_SS_BUCKETS_LDS_KB * 1024 > lds3) lds3 = (size_t)GSR_SS_BUCKETS_LDS_KB * 1024;
    hipLaunchKernelGGL(ss_buckets_kernel, dim3(bmax, B), dim3(kT), lds3, stream, bt);
    return gsr_check_launch("ss_buckets", debug, stream);
}
