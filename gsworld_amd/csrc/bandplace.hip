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
// counts: wave j starts at the first rank whose inclusive running sum exceeds j R / waves.  One workgroup.
// While the camera rests (the depth sort took its splitters unchecked: hdr->ss_blind) the cuts of the last exact
// computation are kept, rescaled to this frame's V, for up to kCutsMaxAge frames: the searches below are a chain of
// dependent global round trips on the frame's critical path, and any monotone cut gives the same point list.
constexpr uint32_t kCutsMagic = 0x43555453u;  // 'CUTS'
constexpr uint32_t kCutsMaxAge = 15u;

__device__ __forceinline__ void band_ranges_body(GsrHeader *__restrict__ hdr, int bmax,
                                                          const uint32_t *__restrict__ bucket_start,
                                                          const uint32_t *__restrict__ bucket_tiles,
                                                          const uint32_t *__restrict__ tile_cum, int waves,
                                                          uint32_t *__restrict__ wave_lo,
                                                          uint32_t *__restrict__ wave_lo_base, uint32_t sig) {
    __shared__ uint32_t s_pre[2048 + 1];  // exclusive running sum of the bucket totals
    __shared__ uint32_t s_w[4];
    const int tid = (int)threadIdx.x;
    const uint32_t V = hdr->V;
    if (tile_cum == nullptr || V == 0u) {  // no running sums (LSD radix variant of the depth sort): equal rank shares
        const uint32_t per = (((V + (uint32_t)waves - 1u) / (uint32_t)waves) + 63u) & ~63u;
        for (int j = tid; j <= waves; j += kBT) wave_lo[j] = min(V, (uint32_t)j * per);
        return;
    }
    const uint32_t base_V = hdr->br_V, age = hdr->br_age;
    bool keep = hdr->ss_blind != 0u && hdr->br_magic == kCutsMagic && hdr->br_P == sig && age < kCutsMaxAge &&
                base_V != 0u;
    if (keep) {
        // (the kept table is checked before it is used -- ascending from 0 to base_V -- see ss_compact_kernel)
        uint32_t bad = 0u;
        for (int j = tid; j < waves; j += kBT) bad |= wave_lo_base[j] > wave_lo_base[j + 1] ? 1u : 0u;
        if (tid == 0) bad |= (wave_lo_base[0] != 0u || wave_lo_base[waves] != base_V) ? 1u : 0u;
        keep = __syncthreads_or((int)bad) == 0;
    }
    if (keep) {
        for (int j = tid; j <= waves; j += kBT)
            wave_lo[j] = j == waves ? V : (uint32_t)(((uint64_t)wave_lo_base[j] * V) / base_V);
        if (tid == 0) hdr->br_age = age + 1u;
        return;
    }
    const int B = (int)hdr->ss_B;  // (the frame's bucket count: depthsort.hip ss_prepare)
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
        // bucket that holds the crossing: the last one whose exclusive sum is <= target
        int b = 0;
        for (int step = B >> 1; step > 0; step >>= 1)
            if (s_pre[b + step] <= target) b += step;
        // inside that bucket the cut is placed by proportion (the bucket's records are ~V / B consecutive depth ranks of
        // similar size): any ascending cut gives the same point list, and the exact one -- a binary search over the
        // bucket's running sums -- was ten dependent global round trips on the critical path of every frame whose
        // camera or scene moves (band_ranges 10 -> 5 us there)
        const uint32_t s0 = bucket_start[b], n = bucket_start[b + 1] - s0, rest = target - s_pre[b];
        const uint32_t tb = s_pre[b + 1] - s_pre[b];
#if GSR_BAND_EXACT_CUTS
        uint32_t lo = 0, hi = n;  // count of entries <= rest: first index with tile_cum > rest
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (tile_cum[s0 + mid] <= rest) lo = mid + 1u; else hi = mid;
        }
#else
        const uint32_t lo = tb != 0u ? (uint32_t)min((uint64_t)n, ((uint64_t)n * rest) / tb) : 0u;
#endif
        const uint32_t cut = j == waves ? V : s0 + lo;
        wave_lo[j] = cut;
        wave_lo_base[j] = cut;
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
__device__ __forceinline__ void band_count_body(const uint2 *__restrict__ rect_sorted,
                                                         const uint32_t *__restrict__ wave_lo, int gx, int NR,
                                                         uint32_t *__restrict__ table, uint32_t *__restrict__ wtable,
                                                         uint32_t *__restrict__ nseg_tab) {
    __shared__ int s_diff[kBW][NC * 64 + 1];
    __shared__ uint32_t s_tot[kBW][NC * 64];
    const int lane = gsr_lane(), wave = gsr_wave();
    const uint32_t r = blockIdx.x, y = blockIdx.y;
    const uint32_t lo = wave_lo[r * kBW + (uint32_t)wave], hi = wave_lo[r * kBW + (uint32_t)wave + 1u];
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
    if (t >= T) return;
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
};

__global__ __launch_bounds__(kBT) void band_ranges_kernel(const GsrBatch<BandArgs> bt) {
    const BandArgs &a = bt.f[blockIdx.y];
    band_ranges_body(a.hdr, a.bmax, a.bucket_start, a.bucket_tiles, a.tile_cum, a.waves, a.wave_lo, a.wave_lo_base, a.sig);
}
template <int NC>
__global__ __launch_bounds__(kBT) void band_count_kernel(const GsrBatch<BandArgs> bt) {
    const BandArgs &a = bt.f[blockIdx.z];
    band_count_body<NC>(a.rect_sorted, a.wave_lo, a.gx, a.NR, a.table, a.wtable, a.nseg);
}
__global__ __launch_bounds__(kBT) void band_scan_kernel(const GsrBatch<BandArgs> bt) {
    const BandArgs &a = bt.f[blockIdx.y];
    band_scan_body(a.table, a.T, a.totals);
}
template <int NC>
__global__ __launch_bounds__(kBT, 8) void band_place_kernel(const GsrBatch<BandArgs> bt) {
    const BandArgs &a = bt.f[blockIdx.z / (uint32_t)kMaxSegments];
    band_place_body<NC>(blockIdx.z % (uint32_t)kMaxSegments, a.rect_sorted, a.order, a.hdr, a.wave_lo, a.gx, a.NR, a.table,
                        a.wtable, a.ranges, a.point_list, a.nseg);
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
        a.point_list = place ? fr[k].b.gidx[0] : (uint32_t *)nullptr;
    }
}

// counts -> ranges, R (tile_starts_kernel lives in binning.hip)
int gsr_launch_band_count(int B, const GsrFrame *fr, bool balanced, bool debug, hipStream_t stream) {
    const GsrSettings &st = *fr[0].st_bin;
    const int gx = gsr_div_up(st.image_width, GSR_TILE), gy = gsr_div_up(st.image_height, GSR_TILE);
    GsrBatch<BandArgs> bt;
    band_args(B, fr, balanced, false, bt);
    const dim3 grid(GSR_BAND_RANGES, gy, B);
    hipLaunchKernelGGL(band_ranges_kernel, dim3(1, B), dim3(kBT), 0, stream, bt);
    if (int e = gsr_check_launch("band_ranges", debug, stream)) return e;
    if (gx <= 64)
        hipLaunchKernelGGL(band_count_kernel<1>, grid, dim3(kBT), 0, stream, bt);
    else if (gx <= 128)
        hipLaunchKernelGGL(band_count_kernel<2>, grid, dim3(kBT), 0, stream, bt);
    else
        hipLaunchKernelGGL(band_count_kernel<4>, grid, dim3(kBT), 0, stream, bt);
    if (int e = gsr_check_launch("band_count", debug, stream)) return e;
    const int T = gx * gy;
    hipLaunchKernelGGL(band_scan_kernel, dim3(gsr_div_up(T, kBW), B), dim3(kBT), 0, stream, bt);
    return gsr_check_launch("band_scan", debug, stream);
}

int gsr_launch_band_place(int B, const GsrFrame *fr, bool debug, hipStream_t stream) {
    const GsrSettings &st = *fr[0].st_bin;
    const int gx = gsr_div_up(st.image_width, GSR_TILE), gy = gsr_div_up(st.image_height, GSR_TILE);
    GsrBatch<BandArgs> bt;
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
