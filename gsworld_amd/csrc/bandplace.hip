// bandplace.hip -- counting placement by tile rows: the point list (upstream duplicateWithKeys + SortPairs +
// identifyTileRanges, rasterizer_impl.cu) without keys, without a sort of instances and
// without LDS atomics.
//
// The Gaussians arrive depth-sorted (depthsort.hip) and a Gaussian touches DISTINCT tiles, so the slot of instance
// (g, t) is  tile_start[t] + #{Gaussians before g in depth order that touch t}.  Round 1 evaluated that per chunk of 256
// depth ranks against ALL tiles (binning.hip tile_count / tile_place: 38 KiB of LDS counters per workgroup, one returning
// LDS atomic per instance in a serial walk per wave, 4-byte stores scattered over ~1000 tile lists per workgroup:
// 57 us for a 15.8 MB list at config 2 and 4.1x write amplification).  Here the work is cut the other way:
//
//   grid = (NR depth-rank ranges) x (tile rows).  Workgroup (r, y) looks at the Gaussians of rank range r that overlap
//   tile row y -- the depth-ordered rects are one coalesced 8-byte stream (rect_sorted, written by the depth sort) -- and
//   compacts those (Gaussian, row) pairs, in order.  The row's cursors live in REGISTERS, one tile column per lane, and
//   the wave takes its pairs one after the other: the lanes inside the pair's column span store the Gaussian at their
//   cursor and advance it.  ~11 instructions per pair, no counters in LDS, no atomics; depth order = program order.
//   (Measured alternatives on MI355X, same decomposition: a ballot per tile column and round of 64 pairs 59 us; lane
//   bits ORed into per-column LDS masks + popcount ranks 29 us; the same staged in LDS and copied out in whole lines
//   37 us -- the LDS atomics and the per-instance round trips cost more than the scattered stores they avoid.)
//
//   band_count   order-free: +1 / -1 at the ends of a pair's column range, running sum = pairs per column; per-wave
//                column counts -> wtable, per-workgroup -> table[t][r]
//   band_scan    exclusive scan of every tile's NR entries, totals[t]           (then tile_starts_kernel of binning.hip:
//                ranges, R, capacity check, compositing order)
//   band_place   cursors = ranges[t].x + table[t][r] + earlier waves of the workgroup, then the rounds with the stores
//
// A (tile, workgroup) run is R / (tiles x NR) ~ 50 consecutive slots at config 2 (200 B) and lanes of one store
// instruction write consecutive words, so the list is written in whole lines.  Tile rows up to 256 tiles wide (4096 px);
// wider grids use the older paths.
#include "gsr_internal.h"

#ifndef GSR_BAND_EXACT_CUTS
#define GSR_BAND_EXACT_CUTS 0
#endif

namespace {

constexpr int kBT = GSR_BLOCK;              // 256 threads = 4 waves
constexpr int kBW = kBT / GSR_WAVE;
constexpr int kRing = 128;                  // (Gaussian, row) pairs buffered per wave
constexpr uint32_t kSplit2 = 1024u, kSplit4 = 2048u;  // instances of a (share, row) unit above which 2 / 4 waves place it
constexpr int kMaxSegments = 4;

// First depth rank of every placement wave.  Equal RANK shares load the waves unevenly -- a wave's work is its
// instances, and the nearest splats (first in depth order) are the largest on screen: at config 5 the 5000 near-band
// splats cover the whole image and all sat in the first three waves of every row (290 us).  With the running sums the
// depth sort leaves (tiles per bucket, running sum inside each bucket) the order is cut at equal cumulative INSTANCE
// counts: wave j starts at the first rank whose inclusive running sum exceeds j R / waves.
constexpr uint32_t kCutsMagic = 0x43555453u;  // 'CUTS'

// Since round 6 the cuts have no launch of their own and, from the second frame on a state, no place on the frame's
// critical path: any ascending cuts give the same point list, only the balance of the placement waves depends on them, and
// consecutive frames of a state -- a sensor camera, a wrist camera riding on the arm -- have nearly the same depth order.
// So every counting workgroup (rank range r, tile row y) takes the five cuts of its four waves from the table the
// PREVIOUS frame left (wave_lo_base: exact equal-cost cuts of that frame's order), rescaled to this frame's V -- one load
// per thread for the check, one rescale per cut -- and the workgroups of tile row 0 leave them in wave_lo for the
// placement; the exact cuts of THIS frame's order are computed beside the scan that follows the counting pass
// (band_cuts_next: a workgroup of its own, nobody waits for it) for the next frame.  A state without a valid table (first
// frame, another model) computes its cuts on the spot, in every counting workgroup, from the 256 .. 2048 bucket totals
// (block scan in LDS, the bucket of a cut found by its owner).  Round 5: band_ranges_kernel, one workgroup per frame
// between the depth sort and the counting pass, 5.2 us of launch floor per closed-loop step -- kept tables were only
// trusted while camera and scene stood still.  First version of the fold (every workgroup computing every frame):
// +4.5 us on the counting kernel, whose 3 840 workgroups are two rounds of the chip and paid the prologue twice.
// -> route: 0 = equal rank shares (no running sums), 1 = the previous frame's cuts rescaled, 2 = cuts computed on the
// spot.  Nothing the routes are decided by is written while the counting kernel runs -- not the header, not wave_lo_base.
__device__ __forceinline__ uint32_t band_cuts_block(const GsrHeader *__restrict__ hdr, int bmax,
                                                    const uint32_t *__restrict__ bucket_start,
                                                    const uint32_t *__restrict__ bucket_tiles,
                                                    const uint32_t *__restrict__ tile_cum,
                                                    const uint32_t *__restrict__ wave_lo_base, uint32_t sig, uint32_t r,
                                                    uint32_t *s_cut /*[kBW + 1]*/, uint32_t *s_pre /*[2048 + 1]*/,
                                                    uint32_t *s_bs /*[2048 + 1]*/, uint32_t *s_w /*[4]*/) {
    constexpr int waves = GSR_BAND_RANGES * kBW;
    static_assert(waves == kBT, "a thread per kept cut");
    const int tid = (int)threadIdx.x;
    // ONE round trip for what the usual route needs: the header words, this thread's pair of the kept cuts (the check) and
    // the cut this thread may rescale
    const uint32_t V = hdr->V, base_V = hdr->br_V, magic = hdr->br_magic, hP = hdr->br_P;
    const int B = (int)hdr->ss_B;  // (the frame's bucket count: depthsort.hip ss_prepare)
    const uint32_t wb0 = wave_lo_base[tid], wb1 = wave_lo_base[tid + 1];
    const uint32_t wbj = tid <= kBW ? wave_lo_base[r * (uint32_t)kBW + (uint32_t)tid] : 0u;
    if (tile_cum == nullptr || V == 0u) {  // no running sums (LSD radix variant of the depth sort): equal rank shares
        const uint32_t per = (((V + (uint32_t)waves - 1u) / (uint32_t)waves) + 63u) & ~63u;
        if (tid <= kBW) s_cut[tid] = min(V, (r * (uint32_t)kBW + (uint32_t)tid) * per);
        __syncthreads();
        return 0u;
    }
    bool keep = magic == kCutsMagic && hP == sig && base_V != 0u;
    if (keep) {
        // (the kept table is checked before it is used -- ascending from 0 to base_V -- see ss_prepare_body)
        uint32_t bad = wb0 > wb1 ? 1u : 0u;
        if (tid == 0) bad |= wb0 != 0u ? 1u : 0u;
        if (tid == waves - 1) bad |= wb1 != base_V ? 1u : 0u;
        keep = __syncthreads_or((int)bad) == 0;
    }
    if (keep) {
        if (tid <= kBW) {
            const uint32_t j = r * (uint32_t)kBW + (uint32_t)tid;
            // (monotone in wbj, 0 -> 0 and base_V -> V: every operation of the chain is; no 64-bit division)
            s_cut[tid] = j == (uint32_t)waves ? V : min(V, (uint32_t)((double)wbj * ((double)V / (double)base_V)));
        }
        __syncthreads();
        return 1u;
    }
    // ---- no valid table (first frame on this state, another model): the cuts of THIS frame's order, on the spot
    const int PER = B / kBT;  // 1, 2, 4 or 8 buckets per thread
    uint32_t bt[8], bs[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        bt[k] = k < PER ? bucket_tiles[tid + k * kBT] : 0u;
        bs[k] = k < PER ? bucket_start[tid + k * kBT] : 0u;
    }
    const uint32_t bs_B = tid == 0 ? bucket_start[B] : 0u;
    // the rows arrive strided (entry tid + 256 k); a thread scans PER consecutive entries: through the LDS
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (k < PER) {
            s_pre[tid + k * kBT] = bt[k];
            s_bs[tid + k * kBT] = bs[k];
        }
    if (tid == 0) s_bs[B] = bs_B;
    __syncthreads();
    uint32_t t[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        t[k] = k < PER ? s_pre[tid * PER + k] : 0u;
        sum += t[k];
    }
    uint32_t total;
    uint32_t run = gsr_block_incl_scan(sum, s_w, total) - sum;
    // the bucket that holds a cut's crossing finds it itself: target in [exclusive sum, + the bucket's total) -- one
    // non-empty bucket per target below the grand total (a binary search was nine dependent LDS round trips)
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k < PER && t[k] != 0u) {
            const int b = tid * PER + k;
#pragma unroll
            for (int c = 0; c <= kBW; c++) {
                const uint32_t j = r * (uint32_t)kBW + (uint32_t)c;
                const uint32_t target = (uint32_t)(((uint64_t)total * j) / (uint32_t)waves);
                if (j != (uint32_t)waves && target >= run && target - run < t[k]) {
                    // inside the bucket the cut is placed by proportion (its records are ~V / B consecutive depth ranks
                    // of similar size)
                    const uint32_t s0 = s_bs[b], n = s_bs[b + 1] - s0, rest = target - run;
                    s_cut[c] = s0 + min(n, (uint32_t)((float)n * ((float)rest / (float)t[k])));
                }
            }
        }
        run += t[k];
    }
    if (tid <= kBW) {
        const uint32_t j = r * (uint32_t)kBW + (uint32_t)tid;
        if (j == (uint32_t)waves || total == 0u) s_cut[tid] = j == (uint32_t)waves ? V : 0u;
    }
    __syncthreads();
    return 2u;
}

// The exact equal-cost cuts of THIS frame's depth order, for the NEXT frame on the state (band_cuts_block route 1): one
// workgroup of 256 threads beside the scan that follows the counting pass -- nobody in this frame waits for it.  With the
// running sums the depth sort leaves (tiles per bucket) the order is cut at equal cumulative placement cost: cut j is
// where the running cost crosses j / 256 of the total; the bucket of the crossing by a search over the scanned bucket
// totals, the place inside the bucket by proportion (round 5 measured the exact place -- a binary search over the
// bucket's running sums, ten dependent global round trips -- against the proportion: same balance).
__device__ __forceinline__ void band_cuts_next(GsrHeader *__restrict__ hdr, const uint32_t *__restrict__ bucket_start,
                                               const uint32_t *__restrict__ bucket_tiles,
                                               const uint32_t *__restrict__ tile_cum, uint32_t *__restrict__ wave_lo_base,
                                               uint32_t sig) {
    constexpr int waves = GSR_BAND_RANGES * kBW;
    __shared__ uint32_t s_pre[2048 + 1];  // exclusive running sum of the bucket totals
    __shared__ uint32_t s_w[4];
    const int tid = (int)threadIdx.x;
    const uint32_t V = hdr->V;
    if (tile_cum == nullptr) return;  // (LSD radix variant of the depth sort: no running sums, equal rank shares)
    if (V == 0u) {
        if (tid == 0) hdr->br_magic = 0u;
        return;
    }
    const int B = (int)hdr->ss_B;
    const int PER = B / kBT;
    uint32_t t[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        t[k] = k < PER ? bucket_tiles[tid * PER + k] : 0u;
        sum += t[k];
    }
    uint32_t total;
    uint32_t run = gsr_block_incl_scan(sum, s_w, total) - sum;
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (k < PER) {
            s_pre[tid * PER + k] = run;
            run += t[k];
        }
    if (tid == 0) s_pre[B] = total;
    __syncthreads();
    for (int j = tid; j <= waves; j += kBT) {
        const uint32_t target = (uint32_t)(((uint64_t)total * (uint32_t)j) / (uint32_t)waves);
        int b = 0;  // the last bucket whose exclusive sum is <= target
        for (int step = B >> 1; step > 0; step >>= 1)
            if (s_pre[b + step] <= target) b += step;
        const uint32_t s0 = bucket_start[b], n = bucket_start[b + 1] - s0, rest = target - s_pre[b];
        const uint32_t tb = s_pre[b + 1] - s_pre[b];
#if GSR_BAND_EXACT_CUTS
        uint32_t lo = 0, hi = n;  // count of entries <= rest: first index with tile_cum > rest
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (tile_cum[s0 + mid] <= rest) lo = mid + 1u; else hi = mid;
        }
#else
        const uint32_t lo = tb != 0u ? min(n, (uint32_t)((float)n * ((float)rest / (float)tb))) : 0u;
#endif
        wave_lo_base[j] = j == waves ? V : s0 + lo;
    }
    if (tid == 0) {
        hdr->br_magic = kCutsMagic;
        hdr->br_P = sig;
        hdr->br_V = V;
        hdr->br_age = 0u;
    }
}

constexpr int kBatch = 12;  // 64-rank rows of the stream requested together (704 ranks per wave at config 2: one batch)

// Counting needs no order at all: a pair adds +1 at its first column and -1 behind its last one; the running sum over
// the columns is the number of pairs covering each.  Two LDS atomics per pair, per-wave difference arrays.
template <int NC>
__device__ __forceinline__ void band_count_body(const uint2 *__restrict__ rect_sorted, GsrHeader *__restrict__ hdr,
                                                         int bmax, const uint32_t *__restrict__ bucket_start,
                                                         const uint32_t *__restrict__ bucket_tiles,
                                                         const uint32_t *__restrict__ tile_cum, int waves,
                                                         uint32_t *__restrict__ wave_lo,
                                                         uint32_t *__restrict__ wave_lo_base, uint32_t sig, int gx, int NR,
                                                         uint32_t *__restrict__ table, uint32_t *__restrict__ wtable,
                                                         uint32_t *__restrict__ nseg_tab) {
    __shared__ int s_diff[kBW][NC * 64 + 1];
    __shared__ uint32_t s_tot[kBW][NC * 64];
    __shared__ uint32_t s_pre[2048 + 1];  // exclusive running sum of the bucket totals
    __shared__ uint32_t s_bs[2048 + 1];   // the bucket starts
    __shared__ uint32_t s_cut[kBW + 1], s_w[4];
    const int lane = gsr_lane(), wave = gsr_wave();
    const uint32_t r = blockIdx.x, y = blockIdx.y;
    const uint32_t route = band_cuts_block(hdr, bmax, bucket_start, bucket_tiles, tile_cum, wave_lo_base, sig, r, s_cut,
                                           s_pre, s_bs, s_w);
    const uint32_t lo = s_cut[wave], hi = s_cut[wave + 1];
    if (y == 0u) {  // tile row 0 leaves the cuts for the placement (and, freshly computed ones, for the frames to come)
        const int tid = (int)threadIdx.x;
        if (tid < kBW || (tid == kBW && r + 1u == gridDim.x)) {
            wave_lo[r * kBW + (uint32_t)tid] = s_cut[tid];
        }
        if (r == 0u && tid == 0) hdr->br_route = route;  // (tools: which route the frame took)
    }
    int *diff = s_diff[wave];
#pragma unroll
    for (int k = 0; k < NC; k++) diff[k * 64 + lane] = 0;
    if (lane == 0) diff[NC * 64] = 0;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t base = lo; base < hi; base += (uint32_t)(kBatch * GSR_WAVE)) {
        uint2 rc[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; u++) {
            const uint32_t i = base + (uint32_t)(u * GSR_WAVE + lane);
            rc[u] = i < hi ? rect_sorted[i] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < kBatch; u++) {
            const uint32_t miny = rc[u].x >> 16, maxy = rc[u].y >> 16;
            if (miny <= y && y < maxy) {  // (an empty slot has maxy = 0)
                atomicAdd(&diff[rc[u].x & 0xffffu], 1);
                atomicAdd(&diff[rc[u].y & 0xffffu], -1);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // running sum over the columns: lane l of round k = column 64 k + l
    uint32_t *wrow = wtable + ((size_t)(y * (uint32_t)NR + r) * kBW + (uint32_t)wave) * (NC * 64);
    uint32_t carry = 0, unit = 0;
#pragma unroll
    for (int k = 0; k < NC; k++) {
        const uint32_t incl = gsr_wave_incl_scan((uint32_t)diff[k * 64 + lane]) + carry;
        wrow[k * 64 + lane] = incl;
        s_tot[wave][k * 64 + lane] = incl;
        carry = (uint32_t)__shfl((int)incl, 63, 64);
        unit += k * 64 + lane < gx ? incl : 0u;
    }
    // instances of this wave's (share, row) unit -> into how many column segments the placement cuts it (band_place_body:
    // a byte per wave; three of its four workgroups per unit leave on this byte alone)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) unit += (uint32_t)__shfl_xor((int)unit, o, 64);
    if (lane == 0)
        reinterpret_cast<uint8_t *>(nseg_tab)[((size_t)y * (uint32_t)NR + r) * kBW + (uint32_t)wave] =
            unit > kSplit4 ? 4 : (unit > kSplit2 ? 2 : 1);
    __syncthreads();
    for (int x = (int)threadIdx.x; x < gx; x += kBT)
        table[((size_t)y * gx + x) * NR + r] = s_tot[0][x] + s_tot[1][x] + s_tot[2][x] + s_tot[3][x];
}

// exclusive scan of every tile's NR (<= 64) entries by one wave; totals[t] = instances of tile t
__device__ __forceinline__ void band_scan_body(uint32_t *__restrict__ table, int T,
                                                        uint32_t *__restrict__ totals) {
    constexpr int PL = (GSR_BAND_RANGES + GSR_WAVE - 1) / GSR_WAVE;  // consecutive entries per lane
    const int t = (int)blockIdx.x * kBW + gsr_wave();
    if (t >= T) return;  // (inlined: the caller goes on)
    const int lane = gsr_lane();
    uint32_t *row = table + (size_t)t * GSR_BAND_RANGES;
    uint32_t v[PL], sum = 0;
#pragma unroll
    for (int k = 0; k < PL; k++) {
        v[k] = lane * PL + k < GSR_BAND_RANGES ? row[lane * PL + k] : 0u;
        sum += v[k];
    }
    const uint32_t incl = gsr_wave_incl_scan(sum);
    uint32_t run = incl - sum;
#pragma unroll
    for (int k = 0; k < PL; k++) {
        if (lane * PL + k < GSR_BAND_RANGES) row[lane * PL + k] = run;
        run += v[k];
    }
    if (lane == 63) totals[t] = incl;
}

// 64 x 64 bit-matrix transpose across the wave: lane l brings row R_l, lane x leaves with column x (bit l = bit x of
// R_l).  Six butterfly steps (swap the off-diagonal s x s blocks, s = 32 .. 1), registers and cross-lane moves only.
__device__ __forceinline__ uint64_t wave_bit_transpose(uint64_t R) {
    const uint32_t lane = (uint32_t)gsr_lane();
    constexpr uint64_t kLow[6] = {0x00000000FFFFFFFFull, 0x0000FFFF0000FFFFull, 0x00FF00FF00FF00FFull,
                                  0x0F0F0F0F0F0F0F0Full, 0x3333333333333333ull, 0x5555555555555555ull};
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const int s = 32 >> i;
        const uint64_t m = kLow[i];
        const uint32_t tlo = (uint32_t)__shfl_xor((int)(uint32_t)R, s, 64);
        const uint32_t thi = (uint32_t)__shfl_xor((int)(uint32_t)(R >> 32), s, 64);
        const uint64_t t = ((uint64_t)thi << 32) | tlo;
        R = (lane & (uint32_t)s) == 0u ? ((R & m) | ((t & m) << s)) : ((R & ~m) | ((t & ~m) >> s));
    }
    return R;
}

// One placing round: lanes [0, n) hold a (Gaussian, row) pair each, in depth order.  The pairs' column spans are the
// rows of a 64 x gx bit matrix; its transpose gives every column the set of pairs that cover it, in lane = depth order,
// so the slot of instance (pair l, column x) is cursor[x] + popcount(column_mask[x] below lane l).  The masks and cursors
// go through a wave-private LDS row; a pair then walks its OWN columns (~5), four per step.
template <int NC>
__device__ __forceinline__ void band_place_round(uint32_t span, uint32_t g, bool valid, int gx, uint32_t *cur,
                                                 unsigned long long *colmask, uint32_t *__restrict__ point_list) {
    const int lane = gsr_lane();
    const uint64_t lt = gsr_lanemask_lt();
    const uint32_t minx = span & 0xffffu, maxx = span >> 16, w = valid ? maxx - minx : 0u;
    // a pair walks up to kNarrow of its own columns; the rest of a WIDE pair (a splat spanning much of the row: 1 % of
    // the Gaussians at config 5, but one of them sits in nearly every round of 64 pairs) is spread over the whole wave
    // instead of keeping 63 idle lanes looping with it
    constexpr uint32_t kNarrow = 16;
    const uint32_t wn = min(w, kNarrow);
    uint32_t wmax = wn;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, o, 64));
    wmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)wmax);
    uint32_t add[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) {
        const uint32_t lo = max(minx, (uint32_t)(64 * k)), hi = min(maxx, (uint32_t)(64 * k + 64));
        uint64_t row = 0ull;
        if (valid && hi > lo) row = (hi - lo == 64u ? ~0ull : ((1ull << (hi - lo)) - 1ull)) << (lo - (uint32_t)(64 * k));
        const uint64_t col = wave_bit_transpose(row);
        colmask[k * 64 + lane] = col;
        add[k] = (uint32_t)__popcll(col);
    }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t j0 = 0; j0 < wmax; j0 += 4u) {
        unsigned long long m[4];
        uint32_t cx[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) {
            m[u] = j0 + u < wn ? colmask[minx + j0 + u] : 0ull;
            cx[u] = j0 + u < wn ? cur[minx + j0 + u] : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++)
            if (j0 + u < wn) point_list[cx[u] + (uint32_t)__popcll(m[u] & lt)] = g;
    }
    for (uint64_t wide = __builtin_amdgcn_ballot_w64(w > kNarrow); wide;) {
        const int src = __ffsll((unsigned long long)wide) - 1;
        wide &= wide - 1ull;
        const uint32_t sw = (uint32_t)__builtin_amdgcn_readlane((int)w, src);
        const uint32_t sx = (uint32_t)__builtin_amdgcn_readlane((int)minx, src);
        const uint32_t sg = (uint32_t)__builtin_amdgcn_readlane((int)g, src);
        const uint64_t below = (1ull << src) - 1ull;
        for (uint32_t j = kNarrow + (uint32_t)lane; j < sw; j += 64u)
            point_list[cur[sx + j] + (uint32_t)__popcll(colmask[sx + j] & below)] = sg;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < NC; k++) cur[k * 64 + lane] += add[k];
    __builtin_amdgcn_wave_barrier();
}

// 64-rank rows requested together by the placement: 6, not the 12 of the counting pass -- with 60 VGPRs eight waves fit
// a SIMD and all 7 680+ waves of config 2 are resident at once (12 rows: 82 VGPRs, five waves, two rounds)
constexpr int kPlaceBatch = 6;

template <int NC>
__device__ __forceinline__ void band_place_body(const uint32_t seg, const uint2 *__restrict__ rect_sorted,
                                                         const uint32_t *__restrict__ order,
                                                         const GsrHeader *__restrict__ hdr,
                                                         const uint32_t *__restrict__ wave_lo, int gx, int NR,
                                                         const uint32_t *__restrict__ table,
                                                         const uint32_t *__restrict__ wtable,
                                                         const uint2 *__restrict__ ranges,
                                                         uint32_t *__restrict__ point_list,
                                                         const uint32_t *__restrict__ nseg_tab) {
    __shared__ uint2 s_ring[kBW][kRing];
    __shared__ unsigned long long s_mask[kBW][NC * 64];
    __shared__ uint32_t s_cur[kBW][NC * 64];
    const int lane = gsr_lane(), wave = gsr_wave();
    const uint32_t r = blockIdx.x, y = blockIdx.y;
    // (the counting pass left, per wave of every unit, into how many column segments the unit is cut: the waves of the
    //  spare segment workgroups -- three in four -- leave on one scalar word)
    const uint32_t nseg = (nseg_tab[(size_t)y * (uint32_t)NR + r] >> (8u * (uint32_t)__builtin_amdgcn_readfirstlane(wave))) & 255u;
    if (seg >= nseg) return;
    if (hdr->overflow) return;
    const uint32_t lo = wave_lo[r * kBW + (uint32_t)wave], hi = wave_lo[r * kBW + (uint32_t)wave + 1u];
    if (lo >= hi) return;
    uint2 *ring = s_ring[wave];
    uint32_t *cur = s_cur[wave];
    unsigned long long *colmask = s_mask[wave];
    const uint32_t *wbase = wtable + (size_t)(y * (uint32_t)NR + r) * kBW * (NC * 64);
    // Column segments.  The shares are cut at equal cost over the whole image, but depth correlates with the image row
    // (a table top recedes upwards), so a share's instances pile up in a few rows: at config 2 a (share, row) unit
    // holds 514 instances on average and 3 683 at most, and the kernel lasted as long as that one wave.  The counting
    // pass left every unit's instances per column, so a heavy unit is cut into 2 or 4 column segments of equal
    // instance count, each placed by its own wave (grid z): the cursors are per column, so a wave that only takes
    // the pairs overlapping its columns, clipped to them, writes exactly the slots the whole unit's wave would.
    uint32_t x0 = 0u, x1 = (uint32_t)gx;
    if (nseg > 1u) {
        uint32_t mine[NC], total = 0u;
#pragma unroll
        for (int k = 0; k < NC; k++) {
            const int x = k * 64 + lane;
            mine[k] = x < gx ? wbase[wave * (NC * 64) + x] : 0u;
            total += mine[k];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) total += (uint32_t)__shfl_xor((int)total, o, 64);
        // (measured: thresholds of 768 / 1536 the same, 512 / 1024 and 350 / 700 slower -- more waves repeat the filter)
        {
            // column x belongs to segment floor(instances before x * nseg / total): contiguous, equal-count segments
            uint32_t before = 0u, first = 0xFFFFFFFFu, last = 0u;
#pragma unroll
            for (int k = 0; k < NC; k++) {
                const uint32_t incl = gsr_wave_incl_scan(mine[k]);
                const uint32_t excl = before + incl - mine[k];
                const bool in = k * 64 + lane < gx && (uint32_t)(((uint64_t)excl * nseg) / total) == seg;
                const uint64_t m = __builtin_amdgcn_ballot_w64(in);
                if (m != 0ull) {
                    first = min(first, (uint32_t)(k * 64 + __builtin_ctzll(m)));
                    last = max(last, (uint32_t)(k * 64 + 64 - __builtin_clzll(m)));
                }
                before += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
            if (first == 0xFFFFFFFFu) return;  // (no column falls into this segment)
            x0 = first;
            x1 = last;
        }
    }
    // cursors: first slot of the tile + rank ranges before mine + earlier waves of this workgroup
#pragma unroll
    for (int k = 0; k < NC; k++) {
        const int x = k * 64 + lane;
        uint32_t s = 0u;
        if (x < gx) {
            const size_t tile = (size_t)y * gx + x;
            s = ranges[tile].x + table[tile * NR + r];
            for (int w = 0; w < wave; w++) s += wbase[w * (NC * 64) + x];
        }
        cur[x] = s;
    }
    __builtin_amdgcn_wave_barrier();
    // the stream: pairs of row y, in depth order, compacted through the ring; a round per 64 pairs
    uint32_t head = 0, tail = 0;  // wave-uniform: pairs consumed / produced
    for (uint32_t base = lo; base < hi; base += (uint32_t)(kPlaceBatch * GSR_WAVE)) {
        uint2 rc[kPlaceBatch];
        uint32_t g[kPlaceBatch];
#pragma unroll
        for (int u = 0; u < kPlaceBatch; u++) {
            const uint32_t i = base + (uint32_t)(u * GSR_WAVE + lane);
            rc[u] = i < hi ? rect_sorted[i] : make_uint2(0u, 0u);
            g[u] = i < hi ? order[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < kPlaceBatch; u++) {
            const uint32_t miny = rc[u].x >> 16, maxy = rc[u].y >> 16;
            const uint32_t minx = max(rc[u].x & 0xffffu, x0), maxx = min(rc[u].y & 0xffffu, x1);
            const bool keep = miny <= y && y < maxy && minx < maxx;
            const uint64_t mask = __builtin_amdgcn_ballot_w64(keep);
            if (mask == 0ull) continue;
            if (keep) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                ring[(tail + rank) & (kRing - 1)] = make_uint2(minx | (maxx << 16), g[u]);
            }
            tail += (uint32_t)__popcll(mask);
            __builtin_amdgcn_wave_barrier();
            if (tail - head >= (uint32_t)GSR_WAVE) {
                const uint2 p = ring[(head + (uint32_t)lane) & (kRing - 1)];
                __builtin_amdgcn_wave_barrier();
                band_place_round<NC>(p.x, p.y, true, gx, cur, colmask, point_list);
                head += (uint32_t)GSR_WAVE;
            }
        }
    }
    if (tail != head) {
        const bool valid = (uint32_t)lane < tail - head;
        const uint2 p = valid ? ring[(head + (uint32_t)lane) & (kRing - 1)] : make_uint2(0u, 0u);
        band_place_round<NC>(p.x, p.y, valid, gx, cur, colmask, point_list);
    }
}

// ---- the four kernels with B frames per launch: the frame's argument block by blockIdx.y (ranges, scan) or, where the
// grid already uses y for the tile row, by blockIdx.z (count: z = frame; place: z = 4 x frame + column segment) --------
struct BandArgs {
    GsrHeader *hdr;
    int bmax;
    const uint32_t *bucket_start, *bucket_tiles, *tile_cum;  // (tile_cum nullptr: equal rank shares)
    int waves;
    uint32_t *wave_lo, *wave_lo_base;
    uint32_t sig;
    const uint2 *rect_sorted;
    int gx, NR, T;
    uint32_t *table, *wtable, *totals, *nseg;
    const uint32_t *order;
    const uint2 *ranges;
    uint32_t *point_list;
    uint32_t r_capacity;  // (the merged scan + starts kernel: what tile_starts_kernel checks R against)
    uint2 *ranges_w;
    const uint32_t *quad_work;  // (... and its tile-order workgroup: the tiles' costs in the previous frame -> tile_order)
    uint32_t *tile_order;
    uint32_t *mirror;  // (GsrOutputs.overflow_mirror: where the capacity check leaves the state's overflow count for the host)
};

template <int NC>
__global__ __launch_bounds__(kBT) void band_count_kernel(const GsrBatch<BandArgs> bt) {
    const BandArgs &a = bt.f[blockIdx.z];
    band_count_body<NC>(a.rect_sorted, a.hdr, a.bmax, a.bucket_start, a.bucket_tiles, a.tile_cum, a.waves, a.wave_lo, a.wave_lo_base,
                        a.sig, a.gx, a.NR, a.table, a.wtable, a.nseg);
}
// (grid.x = workgroups of the scan + 1: the last one computes the next frame's cuts)
__global__ __launch_bounds__(kBT) void band_scan_kernel(const GsrBatch<BandArgs> bt) {
    const BandArgs &a = bt.f[blockIdx.y];
    if (blockIdx.x == gridDim.x - 1) {
        band_cuts_next(a.hdr, a.bucket_start, a.bucket_tiles, a.tile_cum, a.wave_lo_base, a.sig);
        return;
    }
    band_scan_body(a.table, a.T, a.totals);
}
template <int NC>
__global__ __launch_bounds__(kBT, 8) void band_place_kernel(const GsrBatch<BandArgs> bt) {
    const BandArgs &a = bt.f[blockIdx.z / (uint32_t)kMaxSegments];
    band_place_body<NC>(blockIdx.z % (uint32_t)kMaxSegments, a.rect_sorted, a.order, a.hdr, a.wave_lo, a.gx, a.NR, a.table,
                        a.wtable, a.ranges, a.point_list, a.nseg);
}

// ---- band_scan + tile_starts in ONE launch (round 6) --------------------------------------------------------------
// Between the counting pass and the placement the frame needs (a) the exclusive scan of every tile's NR entries and the
// tile totals, (b) the exclusive scan of the totals = the ranges, R and the capacity check.  Until round 5 those were two
// launches at the launch floor (150 workgroups of four waves, then one workgroup: 5.0 + 4.9 us per closed-loop step).
// GSR_SCAN_MODE 1 (default): ONE workgroup of sixteen waves per frame does both -- a wave takes every sixteenth tile,
// sixteen rows in flight per lane, the totals stay in LDS for the second scan.  GSR_SCAN_MODE 2 (measured against it:
// DESIGN.md section 7): the 150 workgroups of (a) as before, and the LAST of them to finish -- a ticket from a
// device-scope atomic behind an agent-scope release, an acquire in front of the reads -- does (b).  0: the two launches.
#ifndef GSR_SCAN_MODE
#define GSR_SCAN_MODE 1
#endif
constexpr int kSST = 1024, kSSW = kSST / GSR_WAVE;  // the merged kernel's workgroup
constexpr int kSSMaxTiles = 8 * kSST;               // (what the merged kernel's LDS and registers are sized for)
// ... but ONE workgroup scanning every tile's row is only worth it while the rows are few: 600 super-tiles (640 x 480
// inference frames) 8.4 us against 5.0 + 4.9 for the two launches; 2 500 tiles (800 x 800 training frames) 24 us against
// 5.1 + 9.3 -- larger grids keep the two launches
constexpr int kSSMergeTiles = 1024;
static_assert(GSR_BAND_RANGES == GSR_WAVE, "a tile's row of the band table is one entry per lane");

// totals[T] (LDS or global) -> ranges, R, capacity check: THREADS threads, each owns `per` consecutive tiles
template <int THREADS, int MAXPER>
__device__ __forceinline__ void band_starts_block(const uint32_t *totals, int T, GsrHeader *__restrict__ hdr,
                                                  uint32_t r_capacity, uint2 *__restrict__ ranges, uint32_t *s_wv,
                                                  uint32_t *__restrict__ mirror) {
    constexpr int NWV = THREADS / GSR_WAVE;
    const int tid = (int)threadIdx.x, lane = gsr_lane(), wave = tid >> 6;
    const int per = (T + THREADS - 1) / THREADS, t0 = tid * per;
    uint32_t v[MAXPER], sum = 0;
#pragma unroll
    for (int k = 0; k < MAXPER; k++) {
        v[k] = (k < per && t0 + k < T) ? totals[t0 + k] : 0u;
        sum += v[k];
    }
    const uint32_t incl = gsr_wave_incl_scan(sum);
    if (lane == 63) s_wv[wave] = incl;
    __syncthreads();
    uint32_t add = 0, grand = 0;
#pragma unroll
    for (int w = 0; w < NWV; w++) {
        const uint32_t x = s_wv[w];
        add += w < wave ? x : 0u;
        grand += x;
    }
    uint32_t run = incl - sum + add;
    const bool overflow = grand > r_capacity;
#pragma unroll
    for (int k = 0; k < MAXPER; k++) {
        const int t = t0 + k;
        if (k < per && t < T) ranges[t] = (v[k] == 0u || overflow) ? make_uint2(0u, 0u) : make_uint2(run, run + v[k]);
        run += v[k];
    }
    if (tid == 0) {
        hdr->R_raw = grand;
        hdr->r_capacity = r_capacity;
        gsr_set_overflow(hdr, overflow, mirror);
        hdr->R = overflow ? 0u : grand;
    }
}

__device__ __forceinline__ void band_scan_starts_body(uint32_t *__restrict__ table, int T, GsrHeader *__restrict__ hdr,
                                                      uint32_t r_capacity, uint2 *__restrict__ ranges,
                                                      uint32_t *__restrict__ mirror) {
    __shared__ uint32_t s_tot[kSSMaxTiles];
    __shared__ uint32_t s_wv[kSSW];
    // A wave takes every sixteenth tile, sixteen rows in flight per lane (lane = rank range: one coalesced 256-byte row per
    // load), a six-instruction DPP scan per row.  (Measured on the way: the same with ds_bpermute scans -- 38 rows per wave,
    // six dependent LDS-crossbar round trips each -- 15.4 us; a THREAD per tile scanning its row in registers -- every
    // load a quarter-used 64-byte request per lane, 20 k requests through one CU's L1 -- 13.6 us.)
    const int lane = gsr_lane(), wave = (int)(threadIdx.x >> 6);
    constexpr int kU = 16;
    for (int t0 = wave; t0 < T; t0 += kSSW * kU) {
        uint32_t v[kU];
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const int t = t0 + u * kSSW;
            v[u] = t < T ? table[(size_t)t * GSR_BAND_RANGES + lane] : 0u;
        }
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const int t = t0 + u * kSSW;
            const uint32_t incl = gsr_wave_incl_scan(v[u]);
            if (t < T) {
                table[(size_t)t * GSR_BAND_RANGES + lane] = incl - v[u];
                if (lane == 63) s_tot[t] = incl;
            }
        }
    }
    __syncthreads();
    band_starts_block<kSST, kSSMaxTiles / kSST>(s_tot, T, hdr, r_capacity, ranges, s_wv, mirror);
}

// (grid.x = 2 or 3: workgroup 1 -- its first four waves -- computes the next frame's cuts; workgroup 2, where the frame's
//  compositor takes its tiles in the order of their cost in the previous frame and nobody dealt them earlier -- training
//  frames, grids above 2048 tiles -- sorts the tiles by that cost: tile_starts_kernel's second workgroup)
__global__ __launch_bounds__(kSST) void band_scan_starts_kernel(const GsrBatch<BandArgs> bt) {
    const BandArgs &a = bt.f[blockIdx.y];
    if (blockIdx.x == 1) {
        if (threadIdx.x >= kBT) return;  // (the waves that stay meet at their own barriers: finished waves do not count)
        band_cuts_next(a.hdr, a.bucket_start, a.bucket_tiles, a.tile_cum, a.wave_lo_base, a.sig);
        return;
    }
    if (blockIdx.x == 2) {
        if (threadIdx.x >= kBT) return;
        __shared__ uint32_t s_bins[64], s_red[4];
        uint32_t key[32];
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const int t = (int)threadIdx.x + i * kBT;
            key[i] = t < a.T ? gsr_tile_order_key(nullptr, a.quad_work, t) : 0u;
        }
        gsr_tile_order_block_keys<32>(key, a.T, a.tile_order, s_bins, s_red);
        return;
    }
    band_scan_starts_body(a.table, a.T, a.hdr, a.r_capacity, a.ranges_w, a.mirror);
}
// GSR_SCAN_MODE 2: band_scan's grid (+ the workgroup of the cuts); the last scanning workgroup of a frame to finish turns
// the totals into ranges
__global__ __launch_bounds__(kBT) void band_scan_tail_kernel(const GsrBatch<BandArgs> bt) {
    const BandArgs &a = bt.f[blockIdx.y];
    __shared__ uint32_t s_last, s_wv[kBW];
    if (blockIdx.x == gridDim.x - 1) {
        band_cuts_next(a.hdr, a.bucket_start, a.bucket_tiles, a.tile_cum, a.wave_lo_base, a.sig);
        return;
    }
    band_scan_body(a.table, a.T, a.totals);
    __threadfence();  // release: this workgroup's totals are visible device-wide before its ticket is
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&a.hdr->tile_queue, 1u) == gridDim.x - 2u ? 1u : 0u;
    __syncthreads();
    if (s_last == 0u) return;
    __threadfence();  // acquire: the other workgroups' totals, not what this CU's caches held before
    if (threadIdx.x == 0) a.hdr->tile_queue = 0u;
    band_starts_block<kBT, kSSMaxTiles / kBT>(a.totals, a.T, a.hdr, a.r_capacity, a.ranges_w, s_wv, a.mirror);
}

// depth-ordered rects for depth sorts that do not write them themselves (the LSD radix variant)
__global__ __launch_bounds__(kBT) void gather_rects_kernel(const uint32_t *__restrict__ order,
                                                           const uint2 *__restrict__ rects,
                                                           const GsrHeader *__restrict__ hdr,
                                                           uint2 *__restrict__ rect_sorted) {
    const uint32_t V = hdr->V;
    for (uint32_t i = blockIdx.x * (uint32_t)kBT + threadIdx.x; i < V; i += gridDim.x * (uint32_t)kBT)
        rect_sorted[i] = rects[order[i]];
}

}  // namespace

bool gsr_band_supported(int gx) { return gx <= 256; }

int gsr_launch_gather_rects(int32_t P, const GeomState &g, bool debug, hipStream_t stream) {
    const int blocks = GeomState::prep_blocks(P) < 1024 ? GeomState::prep_blocks(P) : 1024;
    hipLaunchKernelGGL(gather_rects_kernel, dim3(blocks), dim3(kBT), 0, stream, g.order, g.rects, g.hdr, g.rect_sorted);
    return gsr_check_launch("gather_rects", debug, stream);
}

static void band_args(int B, const GsrFrame *fr, bool balanced, bool place, GsrBatch<BandArgs> &bt) {
    for (int k = 0; k < B; k++) {
        const GeomState &g = fr[k].g;
        const GsrSettings &st = *fr[k].st_bin;
        BandArgs &a = bt.f[k];
        const int32_t P = fr[k].in->P;
        a.hdr = g.hdr;
        a.bmax = gsr_ss_bmax(P);
        a.bucket_start = g.ss_bucket_start;
        a.bucket_tiles = g.bucket_tiles;
        a.tile_cum = balanced ? g.tile_cum : (const uint32_t *)nullptr;
        a.waves = GSR_BAND_RANGES * kBW;
        a.wave_lo = g.wave_lo;
        a.wave_lo_base = g.wave_lo_base;
        // (model size and state layout the kept cuts belong to: see gsr_launch_sample_depth_sort)
        a.sig = (uint32_t)P * 2654435761u ^ (uint32_t)((char *)g.wave_lo_base - (char *)g.hdr);
        a.rect_sorted = g.rect_sorted;
        a.gx = gsr_div_up(st.image_width, GSR_TILE);
        a.NR = GSR_BAND_RANGES;
        a.T = a.gx * gsr_div_up(st.image_height, GSR_TILE);
        a.table = g.band_table;
        a.wtable = g.band_wtable;
        a.totals = g.tile_totals;
        a.nseg = g.band_nseg;
        a.order = g.order;
        a.ranges = fr[k].img.ranges;
        a.ranges_w = fr[k].img.ranges;
        a.r_capacity = fr[k].cap32;
        a.quad_work = fr[k].img.quad_work;
        a.tile_order = fr[k].img.tile_order;
        a.mirror = fr[k].out ? fr[k].out->overflow_mirror : (uint32_t *)nullptr;
        a.point_list = place ? fr[k].b.gidx[0] : (uint32_t *)nullptr;
    }
}

// counts -> ranges, R (tile_starts_kernel lives in binning.hip)
int gsr_launch_band_count(int B, const GsrFrame *fr, bool balanced, bool merge_starts, bool *starts_done, bool debug,
                          hipStream_t stream) {
    const GsrSettings &st = *fr[0].st_bin;
    const int gx = gsr_div_up(st.image_width, GSR_TILE), gy = gsr_div_up(st.image_height, GSR_TILE);
    GsrBatch<BandArgs> bt{};  // (entries beyond B stay zero: nothing uninitialised travels in the kernarg)
    band_args(B, fr, balanced, false, bt);
    const dim3 grid(GSR_BAND_RANGES, gy, B);
    if (gx <= 64)
        hipLaunchKernelGGL(band_count_kernel<1>, grid, dim3(kBT), 0, stream, bt);
    else if (gx <= 128)
        hipLaunchKernelGGL(band_count_kernel<2>, grid, dim3(kBT), 0, stream, bt);
    else
        hipLaunchKernelGGL(band_count_kernel<4>, grid, dim3(kBT), 0, stream, bt);
    if (int e = gsr_check_launch("band_count", debug, stream)) return e;
    const int T = gx * gy;
    // (merged: nothing but the ranges is asked of tile_starts_kernel -- the compositing order was dealt earlier in the frame
    //  (merge_starts) and there are no quadrants to split: the default path)
    // what else tile_starts_kernel would do for this frame: nothing (the quadrants were dealt beside the depth sort:
    // merge_starts), or the tile order by last frame's costs (a compositor that wants one and has no quadrant deal)
    const bool by_tiles = !merge_starts && gsr_render_wants_tile_order(st, T) && !gsr_render_uses_quad_order(st, T);
    const bool merged = GSR_SCAN_MODE != 0 && (merge_starts || (by_tiles && GSR_SCAN_MODE == 1)) && T <= kSSMergeTiles &&
                        gsr_render_split_blocks(st, T) == 0;
    if (starts_done) *starts_done = merged;
    if (merged && GSR_SCAN_MODE == 1) {
        hipLaunchKernelGGL(band_scan_starts_kernel, dim3(by_tiles ? 3 : 2, B), dim3(kSST), 0, stream, bt);
        return gsr_check_launch("band_scan_starts", debug, stream);
    }
    if (merged) {
        hipLaunchKernelGGL(band_scan_tail_kernel, dim3(gsr_div_up(T, kBW) + 1, B), dim3(kBT), 0, stream, bt);
        return gsr_check_launch("band_scan_tail", debug, stream);
    }
    hipLaunchKernelGGL(band_scan_kernel, dim3(gsr_div_up(T, kBW) + 1, B), dim3(kBT), 0, stream, bt);
    return gsr_check_launch("band_scan", debug, stream);
}

int gsr_launch_band_place(int B, const GsrFrame *fr, bool debug, hipStream_t stream) {
    const GsrSettings &st = *fr[0].st_bin;
    const int gx = gsr_div_up(st.image_width, GSR_TILE), gy = gsr_div_up(st.image_height, GSR_TILE);
    GsrBatch<BandArgs> bt{};  // (entries beyond B stay zero: nothing uninitialised travels in the kernarg)
    band_args(B, fr, true, true, bt);
    const dim3 grid(GSR_BAND_RANGES, gy, kMaxSegments * B);
    if (gx <= 64)
        hipLaunchKernelGGL(band_place_kernel<1>, grid, dim3(kBT), 0, stream, bt);
    else if (gx <= 128)
        hipLaunchKernelGGL(band_place_kernel<2>, grid, dim3(kBT), 0, stream, bt);
    else
        hipLaunchKernelGGL(band_place_kernel<4>, grid, dim3(kBT), 0, stream, bt);
    return gsr_check_launch("band_place", debug, stream);
}
