// chunkplace.hip -- counting placement by CHUNKS of the depth order (upstream duplicateWithKeys + SortPairs +
// identifyTileRanges, rasterizer_impl.cu): the point list without keys and without a sort
// of instances, in three launches that each fill the chip.
//
// The Gaussians arrive depth-sorted (depthsort.hip: `order`, `rect_sorted`), so the slot of instance (g, t) is
//     ranges[t].x + #{Gaussians before g in depth order that touch t}.
// The band placement (bandplace.hip) cuts that count by (tile row, share of the depth order): every one of the 30 tile
// rows re-streams and re-tests all V rects, the shares have to be cut at equal cost first (band_ranges: a chain of
// dependent searches) and a wave ranks 64 pairs at a time through 64 x 64 bit transposes: 5 launches, 48 us at
// config 2, a third of it the tail of the heaviest (share, row) units.  Here the depth order is cut into fixed chunks
// of CH ranks and the count is split as
//     (instances of t in earlier chunks)  +  (ranks below g inside its chunk that touch t):
//
//   cp_count   one workgroup per chunk: +1 / -1 at the ends of every (Gaussian, tile row) span in an LDS difference
//              image of the whole tile grid, running sums per row -> table[chunk][tile]
//   cp_scan    one workgroup per 64 tiles: exclusive running sum of every tile's column over the chunks (in place)
//              and the tile totals                      (then tile_starts_kernel of binning.hip: ranges, R, capacity)
//   cp_place   one workgroup per (chunk, group of tile rows <= 256 tiles): a bit mask per tile over the chunk's CH ranks
//              (one LDS atomic OR per instance), popcount prefixes per 64 ranks, then
//              slot = ranges[t].x + table[chunk][t] + prefix[word][t] + popcount(mask[word][t] below the rank's bit).
//
// Work is enumerated as (Gaussian, tile row) spans of at most 16 columns, compacted into an LDS list and dealt
// round-robin to the threads, so a thread's inner loop is a handful of columns whatever the splat sizes are (a splat
// that covers the whole image is ~100 spans spread over the workgroup, not one lane looping 1200 times).
// Any chunking gives the same list: depth order = (chunk, rank in chunk) order.  Tile grids up to 4096 tiles and 256
// tiles wide; beyond that the band / older placements run.
#include "gsr_internal.h"

// -DCP_TIMING=<unit>: cycle stamps of that unit's phases into the state's debug words 48.. (tools/scratch/cp_stamps.py)
#ifdef CP_TIMING
#define CP_STAMP(k) do { __syncthreads(); if (unit == (uint32_t)(CP_TIMING) && threadIdx.x == 0) dbg[48 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CP_STAMP(k) do { } while (0)
#endif
#ifndef CP_EXP
#define CP_EXP 0  // tuning experiments (tools/gpu_variants.sh): 2 = no mask atomics, 3 = loads + scan only
#endif

namespace {

constexpr int kT = GSR_BLOCK;          // 256 threads = 4 waves
constexpr int kCH = 512;               // depth ranks per chunk: 8 mask words per tile
constexpr int kNW = kCH / GSR_WAVE;
constexpr int kPT = kCH / kT;          // ranks per thread
#ifndef CP_TG
#define CP_TG 128
#endif
#ifndef CP_CAP
#define CP_CAP 1024
#endif
constexpr int kTG = CP_TG;             // tiles per placement group (whole tile rows)
constexpr int kSpan = 8;               // columns per span: a wave takes 64 / kSpan spans at a time, one lane per column
constexpr int kPairCap = CP_CAP;       // spans staged in LDS at a time
constexpr int kMaxGrid = 8192;         // workgroups launched at most (they loop over their units)

__device__ __forceinline__ void cp_unpack(uint2 rc, uint32_t &minx, uint32_t &miny, uint32_t &maxx, uint32_t &maxy) {
    minx = rc.x & 0xffffu; miny = rc.x >> 16; maxx = rc.y & 0xffffu; maxy = rc.y >> 16;
}

// ---------------------------------------------------------------------------------------------------------
// cp_count: table[chunk][tile] = Gaussians of the chunk that touch the tile.  Order-free: a (Gaussian, row) span adds
// +1 at its first column and -1 behind its last one; the running sum over a row's columns is the cover count.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kT) void cp_count_kernel(const uint2 *__restrict__ rect_sorted,
                                                      const GsrHeader *__restrict__ hdr, int gx, int gy,
                                                      uint32_t *__restrict__ table) {
    extern __shared__ int s_diff[];  // [gy][gx + 1]
    const int tid = (int)threadIdx.x, lane = gsr_lane(), wave = gsr_wave();
    const uint32_t V = hdr->V;
    // (the grid is sized by P -- V is only known on the device -- and capped: a workgroup takes chunks b, b + grid, ...)
    for (uint32_t chunk = blockIdx.x; chunk * (uint32_t)kCH < V; chunk += gridDim.x) {
    const uint32_t base = chunk * (uint32_t)kCH;
    uint2 rc[kPT];
#pragma unroll
    for (int k = 0; k < kPT; k++) {  // (requested before the LDS image is cleared: the loads fly under it)
        const uint32_t i = base + (uint32_t)(k * kT + tid);
        rc[k] = i < V ? rect_sorted[i] : make_uint2(0u, 0u);
    }
    const int stride = gx + 1, T = gx * gy;
    for (int i = tid; i < gy * stride; i += kT) s_diff[i] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPT; k++) {
        uint32_t minx, miny, maxx, maxy;
        cp_unpack(rc[k], minx, miny, maxx, maxy);  // (an empty slot has maxy = 0)
        for (uint32_t y = miny; y < maxy; y++) {
            atomicAdd(&s_diff[(int)y * stride + (int)minx], 1);
            atomicAdd(&s_diff[(int)y * stride + (int)maxx], -1);
        }
    }
    __syncthreads();
    uint32_t *row_out = table + (size_t)chunk * T;
    for (int y = wave; y < gy; y += kT / GSR_WAVE) {
        uint32_t carry = 0;
        for (int x0 = 0; x0 < gx; x0 += GSR_WAVE) {
            const int x = x0 + lane;
            const uint32_t incl = gsr_wave_incl_scan(x < gx ? (uint32_t)s_diff[y * stride + x] : 0u) + carry;
            if (x < gx) row_out[y * gx + x] = incl;
            carry = (uint32_t)__shfl((int)incl, 63, 64);
        }
    }
    __syncthreads();  // (the image is cleared again by the next chunk)
    }
}

// ---------------------------------------------------------------------------------------------------------
// cp_scan: per tile, exclusive running sum over the chunks (in place) and the total.  Lane = tile, the four waves take
// contiguous quarters of the chunks: partial sums first, then the running sums on a second sweep (the rows are L2 /
// L1 resident by then).
// ---------------------------------------------------------------------------------------------------------
constexpr int kScanT = 1024;  // 16 waves: the chunks of a 64-tile column are split 16 ways, one or two round trips each
constexpr int kScanB = 24;    // rows requested together per lane

__global__ __launch_bounds__(kScanT) void cp_scan_kernel(uint32_t *__restrict__ table,
                                                         const GsrHeader *__restrict__ hdr, int T,
                                                         uint32_t *__restrict__ totals) {
    constexpr int NWV = kScanT / GSR_WAVE;
    __shared__ uint32_t s_sum[NWV][GSR_WAVE];
    const int lane = gsr_lane(), wave = (int)(threadIdx.x >> 6);
    const int t = (int)blockIdx.x * GSR_WAVE + lane;
    const bool ok = t < T;
    const uint32_t V = hdr->V;
    const int nch = (int)((V + (uint32_t)kCH - 1u) / (uint32_t)kCH);
    const int q = (nch + NWV - 1) / NWV, c0 = min(nch, wave * q), c1 = min(nch, c0 + q);
    uint32_t sum = 0;
    for (int c = c0; c < c1; c += kScanB) {
        uint32_t v[kScanB];
#pragma unroll
        for (int u = 0; u < kScanB; u++) v[u] = (ok && c + u < c1) ? table[(size_t)(c + u) * T + t] : 0u;
#pragma unroll
        for (int u = 0; u < kScanB; u++) sum += v[u];
    }
    s_sum[wave][lane] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NWV; w++) {
        const uint32_t s = s_sum[w][lane];
        if (w < wave) run += s;
        total += s;
    }
    for (int c = c0; c < c1; c += kScanB) {
        uint32_t v[kScanB];
#pragma unroll
        for (int u = 0; u < kScanB; u++) v[u] = (ok && c + u < c1) ? table[(size_t)(c + u) * T + t] : 0u;
#pragma unroll
        for (int u = 0; u < kScanB; u++) {
            if (ok && c + u < c1) table[(size_t)(c + u) * T + t] = run;
            run += v[u];
        }
    }
    if (wave == 0 && ok) totals[t] = total;
}

// ---------------------------------------------------------------------------------------------------------
// cp_place: the point list.  Workgroup = (chunk, group of whole tile rows with at most kTG tiles).
// ---------------------------------------------------------------------------------------------------------
struct CpRect {
    uint32_t minx, maxx, y0, ny, nsx;  // clipped to the group's rows; ny * nsx spans
};

__device__ __forceinline__ CpRect cp_clip(uint2 rc, uint32_t gy0, uint32_t gy1) {
    uint32_t minx, miny, maxx, maxy;
    cp_unpack(rc, minx, miny, maxx, maxy);
    CpRect r;
    r.minx = minx;
    r.maxx = maxx;
    r.y0 = max(miny, gy0);
    const uint32_t y1 = min(maxy, gy1);
    r.ny = y1 > r.y0 ? y1 - r.y0 : 0u;
    r.nsx = (maxx - minx + (uint32_t)kSpan - 1u) / (uint32_t)kSpan;
    if (r.ny == 0u) r.nsx = 0u;
    return r;
}

constexpr int kPlaceT = kCH;                    // one thread per rank of the chunk: 8 waves
constexpr int kPlaceW = kPlaceT / GSR_WAVE;
constexpr int kSlots = GSR_WAVE / kSpan;        // spans a wave takes per step, one lane per column
constexpr int kUnroll = 4;                      // independent steps in flight per wave (the loop is LDS round trips)

// thread-private span generator: writes those of the thread's spans whose number falls into [w0, w0 + kPairCap)
__device__ __forceinline__ void cp_emit(const CpRect &r, uint32_t first, uint32_t p, uint32_t gy0, uint32_t w0,
                                        uint2 *pairs) {
    const uint32_t n = r.ny * r.nsx;
    if (n == 0u || first + n <= w0 || first >= w0 + (uint32_t)kPairCap) return;
    uint32_t j = first < w0 ? w0 - first : 0u;  // first span of this rect inside the window
    uint32_t yy = 0u, sx = 0u;
    if (j != 0u) { yy = j / r.nsx; sx = j - yy * r.nsx; }
    for (; j < n && first + j < w0 + (uint32_t)kPairCap; j++) {
        const uint32_t x0 = r.minx + sx * (uint32_t)kSpan, x1 = min(r.maxx, x0 + (uint32_t)kSpan);
        pairs[first + j - w0] = make_uint2(p | ((r.y0 + yy - gy0) << 16), x0 | (x1 << 16));
        if (++sx == r.nsx) { sx = 0u; yy++; }
    }
}

__global__ __launch_bounds__(kPlaceT) void cp_place_kernel(const uint2 *__restrict__ rect_sorted,
                                                           const uint32_t *__restrict__ order,
                                                           const GsrHeader *__restrict__ hdr, int gx, int gy,
                                                           int rows_per_group, int groups,
                                                           const uint32_t *__restrict__ table,
                                                           const uint2 *__restrict__ ranges,
                                                           uint32_t *__restrict__ point_list,
                                                           uint64_t *__restrict__ dbg) {
    __shared__ unsigned long long s_mask[kNW][kTG];
    __shared__ uint32_t s_cp[kNW][kTG];  // first slot of the chunk in the tile's list + ranks below the 64-rank word
    __shared__ uint32_t s_g[kCH];
    __shared__ uint2 s_pair[kPairCap];
    __shared__ uint32_t s_w[kPlaceW];
    const int tid = (int)threadIdx.x, lane = gsr_lane(), wave = (int)(threadIdx.x >> 6);
    if (hdr->overflow) return;
    const uint32_t V = hdr->V;
    const uint32_t nch = (V + (uint32_t)kCH - 1u) / (uint32_t)kCH;
    // units = (chunk, row group), chunk-major so that the workgroups of one chunk run close together (they read the
    // same rects); the grid is capped and sized by P, a workgroup takes units b, b + grid, ...
    for (uint32_t unit = blockIdx.x; unit < nch * (uint32_t)groups; unit += gridDim.x) {
    const uint32_t chunk = unit / (uint32_t)groups, grp = unit - chunk * (uint32_t)groups;
    const uint32_t base = chunk * (uint32_t)kCH;
    const uint32_t gy0 = grp * (uint32_t)rows_per_group, gy1 = min((uint32_t)gy, gy0 + (uint32_t)rows_per_group);
    const int ntl = (int)(gy1 - gy0) * gx, tile0 = (int)gy0 * gx, T = gx * gy;
    CP_STAMP(0);
    const uint32_t i = base + (uint32_t)tid;
    const uint2 rc = i < V ? rect_sorted[i] : make_uint2(0u, 0u);
    const uint32_t gi = i < V ? order[i] : 0u;
    uint32_t cur0 = 0u;
    if (tid < ntl) cur0 = ranges[tile0 + tid].x + table[(size_t)chunk * T + tile0 + tid];
    for (int k = tid; k < kNW * kTG; k += kPlaceT) (&s_mask[0][0])[k] = 0ull;
    const CpRect r = cp_clip(rc, gy0, gy1);
    const uint32_t mine = r.ny * r.nsx;
    s_g[tid] = gi;
    // spans before mine / spans of the unit (block scan over 8 waves)
    uint32_t first, Q;
    {
        const uint32_t incl = gsr_wave_incl_scan(mine);
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        uint32_t add = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < kPlaceW; w++) {
            const uint32_t v = s_w[w];
            if (w < wave) add += v;
            tot += v;
        }
        first = add + incl - mine;
        Q = tot;
    }
    if (Q == 0u) { __syncthreads(); continue; }
    CP_STAMP(1);
#ifdef CP_TIMING
    if (unit == (uint32_t)(CP_TIMING) && tid == 0) dbg[48 + 8] = Q;
#endif
#if CP_EXP == 3
    __syncthreads();
    continue;
#endif
    // ---- sweep 1: the masks.  lane = (span slot, column of the span): the kSpan lanes of a slot touch consecutive
    // tiles; kUnroll independent steps per wave are in flight (a step is a chain of LDS round trips)
    for (uint32_t w0 = 0; w0 < Q; w0 += (uint32_t)kPairCap) {
        if (w0 > 0u) __syncthreads();
        cp_emit(r, first, (uint32_t)tid, gy0, w0, s_pair);
        __syncthreads();
        if (w0 == 0u) CP_STAMP(2);
        const uint32_t nq = min((uint32_t)kPairCap, Q - w0);
        for (uint32_t j0 = (uint32_t)(wave * kSlots * kUnroll); j0 < nq; j0 += (uint32_t)(kPlaceW * kSlots * kUnroll)) {
            uint2 pr[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                const uint32_t j = j0 + (uint32_t)(u * kSlots + lane / kSpan);
                pr[u] = j < nq ? s_pair[j] : make_uint2(0u, 0u);
            }
#if CP_EXP != 2
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                const uint32_t p = pr[u].x & 0xffffu, x = (pr[u].y & 0xffffu) + (uint32_t)(lane % kSpan);
                if (x < (pr[u].y >> 16))
                    atomicOr(&s_mask[p >> 6][(pr[u].x >> 16) * (uint32_t)gx + x], 1ull << (p & 63u));
            }
#endif
        }
    }
    __syncthreads();
    CP_STAMP(3);
    // ---- ranks below every 64-rank word
    if (tid < ntl) {
        uint32_t run = cur0;  // first slot of the chunk in this tile's list
#pragma unroll
        for (int w = 0; w < kNW; w++) {
            s_cp[w][tid] = run;
            run += (uint32_t)__popcll(s_mask[w][tid]);
        }
    }
    __syncthreads();
    CP_STAMP(4);
    // ---- sweep 2: the slots.  (Staging the unit's instances tile-major in LDS and writing them out in runs of
    // consecutive words was built and measured: 39 us against 33 us for the direct stores below -- a (chunk, tile) run
    // is ~10 words, too short for the extra pass to pay.)
    for (uint32_t w0 = 0; w0 < Q; w0 += (uint32_t)kPairCap) {
        if (Q > (uint32_t)kPairCap) {  // (a single window is still staged from sweep 1)
            __syncthreads();
            cp_emit(r, first, (uint32_t)tid, gy0, w0, s_pair);
            __syncthreads();
        }
        const uint32_t nq = min((uint32_t)kPairCap, Q - w0);
        for (uint32_t j0 = (uint32_t)(wave * kSlots * kUnroll); j0 < nq; j0 += (uint32_t)(kPlaceW * kSlots * kUnroll)) {
            uint2 pr[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                const uint32_t j = j0 + (uint32_t)(u * kSlots + lane / kSpan);
                pr[u] = j < nq ? s_pair[j] : make_uint2(0u, 0u);
            }
            uint32_t cp[kUnroll], g[kUnroll];
            unsigned long long m[kUnroll];
            bool on[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                const uint32_t p = pr[u].x & 0xffffu, x = (pr[u].y & 0xffffu) + (uint32_t)(lane % kSpan);
                on[u] = x < (pr[u].y >> 16);
                const uint32_t tl = on[u] ? (pr[u].x >> 16) * (uint32_t)gx + x : 0u;
                cp[u] = s_cp[p >> 6][tl];
                m[u] = s_mask[p >> 6][tl];
                g[u] = s_g[p];
            }
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                const uint32_t p = pr[u].x & 0xffffu;
                if (on[u]) point_list[cp[u] + (uint32_t)__popcll(m[u] & ((1ull << (p & 63u)) - 1ull))] = g[u];
            }
        }
    }
    __syncthreads();  // (the next unit rewrites the LDS state)
    CP_STAMP(5);
    }
}

}  // namespace

bool gsr_chunk_supported(int gx, int gy) { return gx <= kTG && gx * gy <= 4096; }


// counts -> per-chunk offsets + tile totals (tile_starts_kernel of binning.hip turns the totals into ranges, R)
int gsr_launch_chunk_count(const GsrSettings &st, int32_t P, const GeomState &g, bool debug, hipStream_t stream) {
    const int gx = gsr_div_up(st.image_width, GSR_TILE), gy = gsr_div_up(st.image_height, GSR_TILE), T = gx * gy;
    const int chunks = gsr_div_up(P > 0 ? P : 1, kCH);
    // table = g.tile_table: tiles x ceil(P / 256) words were carved for the round-1 counting placement; this one
    // needs tiles x ceil(P / kCH)
    hipLaunchKernelGGL(cp_count_kernel, dim3(chunks < kMaxGrid ? chunks : kMaxGrid), dim3(kT), (size_t)gy * (gx + 1) * sizeof(int), stream, g.rect_sorted,
                       g.hdr, gx, gy, g.tile_table);
    if (int e = gsr_check_launch("cp_count", debug, stream)) return e;
    hipLaunchKernelGGL(cp_scan_kernel, dim3(gsr_div_up(T, GSR_WAVE)), dim3(kScanT), 0, stream, g.tile_table, g.hdr, T,
                       g.tile_totals);
    return gsr_check_launch("cp_scan", debug, stream);
}

int gsr_launch_chunk_place(const GsrSettings &st, int32_t P, const GeomState &g, const BinningState &b,
                           const ImageState &img, bool debug, hipStream_t stream) {
    const int gx = gsr_div_up(st.image_width, GSR_TILE), gy = gsr_div_up(st.image_height, GSR_TILE);
    const int chunks = gsr_div_up(P > 0 ? P : 1, kCH);
    const int rows = kTG / gx > 0 ? kTG / gx : 1, groups = gsr_div_up(gy, rows);
    const int64_t units = (int64_t)chunks * groups;
    hipLaunchKernelGGL(cp_place_kernel, dim3((unsigned)(units < kMaxGrid ? units : kMaxGrid)), dim3(kPlaceT), 0, stream,
                       g.rect_sorted, g.order, g.hdr, gx, gy, rows, groups, g.tile_table, img.ranges, b.gidx[0], g.ss_dbg);
    return gsr_check_launch("cp_place", debug, stream);
}
