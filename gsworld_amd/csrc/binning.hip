// binning.hip -- instance emission in depth order, stable tile sort, tile ranges
// (upstream rasterizer_impl.cu duplicateWithKeys + SortPairs + identifyTileRanges).
//
// MI355X-first restructuring: instead of one 43-bit sort of R (u64 key, u32 value) pairs (>= 6 passes x 24 B x R),
// the V visible Gaussians are depth-sorted first (sort.hip; 8 B x V per pass), instances are emitted in that
// order as (tile id, Gaussian index) and only the tile id (<= 16 bits) is radix-sorted, stably.  The resulting
// point list is identical to the reference's: tile-major, ascending depth bits, ties by ascending index.
#include "gsr_internal.h"

namespace {

// One workgroup per chunk of GSR_SORT_CHUNK depth-ordered Gaussians.  The chunk's first output slot comes from
// the scanned per-chunk sums; inside the chunk a block scan per 256 Gaussians gives each its slot range.
// Emission is hybrid: a lane writes a short tile list (<= 4) itself, longer lists are written by the whole
// wave, 64 tiles per step, so that a few large splats do not serialise the wave.
__global__ __launch_bounds__(GSR_BLOCK) void emit_kernel(const uint32_t *__restrict__ order,
                                                         const uint32_t *__restrict__ tiles_touched,
                                                         const uint2 *__restrict__ rects,
                                                         const uint32_t *__restrict__ chunk_offsets,
                                                         const GsrHeader *__restrict__ hdr, int gx,
                                                         uint32_t *__restrict__ out_tile,
                                                         uint32_t *__restrict__ out_gidx) {
    __shared__ uint32_t s_w[4];
    const uint32_t V = hdr->V;
    const uint32_t base = blockIdx.x * (uint32_t)GSR_SORT_CHUNK;
    if (base >= V || hdr->overflow) return;
    uint32_t running = chunk_offsets[blockIdx.x];
    const int lane = gsr_lane();
    for (int r = 0; r < GSR_SORT_ITEMS; r++) {
        const uint32_t i = base + (uint32_t)r * GSR_BLOCK + threadIdx.x;
        const bool valid = i < V;
        const uint32_t g = valid ? order[i] : 0u;
        const uint32_t t = valid ? tiles_touched[g] : 0u;
        uint2 rc = make_uint2(0u, 0u);
        if (valid) rc = rects[g];
        uint32_t total;
        const uint32_t incl = gsr_block_incl_scan(t, s_w, total);
        const uint32_t off = running + incl - t;
        running += total;
        const uint32_t minx = rc.x & 0xffffu, miny = rc.x >> 16, maxx = rc.y & 0xffffu;
        const uint32_t width = maxx - minx;
        if (t > 0u && t <= 4u) {
            uint32_t x = minx, y = miny;
            for (uint32_t j = 0; j < t; j++) {
                out_tile[off + j] = y * (uint32_t)gx + x;
                out_gidx[off + j] = g;
                if (++x == maxx) { x = minx; y++; }
            }
        }
        uint64_t big = __ballot(t > 4u);
        while (big) {
            const int src = __ffsll((unsigned long long)big) - 1;
            big &= big - 1ull;
            const uint32_t bt = __shfl(t, src, 64), boff = __shfl(off, src, 64), bg = __shfl(g, src, 64);
            const uint32_t bminx = __shfl(minx, src, 64), bminy = __shfl(miny, src, 64);
            const uint32_t bw = __shfl(width, src, 64);
            for (uint32_t j = (uint32_t)lane; j < bt; j += 64u) {
                const uint32_t yy = j / bw, xx = j - yy * bw;
                out_tile[boff + j] = (bminy + yy) * (uint32_t)gx + (bminx + xx);
                out_gidx[boff + j] = bg;
            }
        }
    }
}

__global__ __launch_bounds__(GSR_BLOCK) void zero_ranges_kernel(uint2 *ranges, int tiles) {
    const int i = blockIdx.x * GSR_BLOCK + (int)threadIdx.x;
    if (i < tiles) ranges[i] = make_uint2(0u, 0u);
}

// ranges[t] = [first, last+1) of tile t in the sorted instance list; untouched tiles stay (0,0)
__global__ __launch_bounds__(GSR_BLOCK) void tile_ranges_kernel(const uint32_t *__restrict__ tiles,
                                                                const GsrHeader *__restrict__ hdr,
                                                                uint2 *__restrict__ ranges) {
    const uint32_t R = hdr->R;
    const uint32_t i = blockIdx.x * (uint32_t)GSR_BLOCK + threadIdx.x;
    if (i >= R) return;
    const uint32_t t = tiles[i];
    if (i == 0u) {
        ranges[t].x = 0u;
    } else {
        const uint32_t p = tiles[i - 1u];
        if (p != t) {
            ranges[p].y = i;
            ranges[t].x = i;
        }
    }
    if (i == R - 1u) ranges[t].y = R;
}


// ---------------------------------------------------------------------------------------------------------
// Counting placement (default path, tile grids up to GSR_MAX_COUNT_TILES).
//
// The Gaussians arrive depth-sorted, and every Gaussian touches DISTINCT tiles.  The final slot of instance
// (g, t) is therefore  tile_start[t] + #{Gaussians earlier in depth order that touch t}.  A workgroup owns 256
// consecutive depth ranks (64 per wave); a wave walks its Gaussians one at a time with its 64 lanes spread over
// that Gaussian's tiles, so one LDS counter per (wave, tile) ranks instances in depth order without any key
// material: ds_add_rtn from one wave retires in issue order and lanes of one instruction never collide.
//   tile_count  : per-workgroup tile histogram  -> table[t][workgroup]
//   row scan    : exclusive scan of each tile's row over the workgroups, row total -> totals[t]   (sort.hip)
//   tile_starts : exclusive scan of totals -> ranges[t] = [start, start+total), R, overflow check
//   tile_place  : recount per wave, turn the counters into running cursors, write the point list
// This replaces emit + a 2-pass radix sort of R (tile, index) pairs + identifyTileRanges.
//
// A workgroup always owns 256 consecutive depth ranks (one table column), split over NW waves.  The placement walk
// is a serial chain per wave (one returning LDS atomic + one scattered store per Gaussian), so the fewer ranks a
// wave owns the shorter the critical path; lanes that hold no Gaussian still help spread tiles.  Measured on
// MI355X: 8 waves x 32 ranks beats 4 x 64 at every grid size (53 -> 37 us at 1200 tiles, 364 -> 300 us at 2500) and
// 16 x 16 buys nothing more.
// ---------------------------------------------------------------------------------------------------------
constexpr int kPlaceWaves = 8;
constexpr int kPlaceGrid = 1536;  // workgroups launched at most (256 CUs x 6); they stride over the 256-rank chunks

// rows of tiles per band: the NW x (band tiles) counters stay within 40 KiB so that four workgroups share a CU
// (one band at 1200 tiles, two at 2500, seven at 1920x1080)
inline int place_band_rows(int gx, int gy) {
    // (measured at 800x800 / 500 k Gaussians: 10 / 20 / 30 / 40 / 64 KiB -> 1.37 / 1.37 / 1.31 / 1.31 / 1.35 ms per
    // training step, one 80 KiB band 1.42 ms; at 640x480 splitting the 38 KiB grid only costs)
    const int max_tiles = 40 * 1024 / (kPlaceWaves * (int)sizeof(uint32_t));
    int rows = max_tiles / gx;
    if (rows < 1) rows = 1;
    return rows < gy ? rows : gy;
}

struct WaveSplats {  // this lane's Gaussian (t == 0: none)
    uint32_t g, t;
    uint2 rc;
};

// Tile rows [row0, row1) are this workgroup's band: grids whose NW x tiles counters would not leave room for several
// workgroups per CU are cut into bands of rows (blockIdx.y), and a Gaussian's rect is clipped to the band here, so
// the walks below only ever see tiles of the band.
template <int NW>
__device__ __forceinline__ WaveSplats load_wave_splats(const uint32_t *__restrict__ order,
                                                       const uint32_t *__restrict__ tiles_touched,
                                                       const uint2 *__restrict__ rects, uint32_t V, uint32_t block_base,
                                                       uint32_t row0, uint32_t row1) {
    constexpr int PER_WAVE = GSR_BLOCK / NW;
    const int lane = gsr_lane();
    const uint32_t rank = block_base + (uint32_t)(gsr_wave() * PER_WAVE + lane);
    WaveSplats s;
    s.g = 0u; s.t = 0u; s.rc = make_uint2(0u, 0u);
    if (lane < PER_WAVE && rank < V) {
        s.g = order[rank];
        const uint2 rc = rects[s.g];
        const uint32_t minx = rc.x & 0xffffu, maxx = rc.y & 0xffffu;
        const uint32_t miny = max(rc.x >> 16, row0), maxy = min(rc.y >> 16, row1);
        if (tiles_touched[s.g] != 0u && maxy > miny) {
            s.t = (maxx - minx) * (maxy - miny);
            s.rc = make_uint2(minx | (miny << 16), maxx | (maxy << 16));
        }
    }
    return s;
}

template <bool PLACE>
__device__ __forceinline__ void walk_wave(const WaveSplats s, int gx, uint32_t tile0, uint32_t *cnt,
                                          uint32_t *__restrict__ out) {
    const int lane = gsr_lane();
    const uint32_t g = s.g, t = s.t;
    const uint2 rc = s.rc;
    const uint32_t minx = rc.x & 0xffffu, miny = rc.x >> 16, maxx = rc.y & 0xffffu, width = maxx - minx;
    uint64_t todo;
    if (!PLACE) {
        // counting does not care about order: a lane walks a short tile list itself (LDS atomics resolve any
        // collisions between lanes), only long lists are spread over the wave
        if (t > 0u && t <= 32u) {
            uint32_t x = minx, tile_row = miny * (uint32_t)gx - tile0;
            for (uint32_t j = 0; j < t; j++) {
                atomicAdd(&cnt[tile_row + x], 1u);
                if (++x == maxx) { x = minx; tile_row += (uint32_t)gx; }
            }
        }
        todo = __ballot(t > 32u);
    } else {
        todo = __ballot(t > 0u);
    }
    // j / width without an integer division: (j + 0.5) * (1 / width) truncates to the exact quotient for every
    // j below 2^21 (the margin 0.5 / width dwarfs the 2^-23 relative error of the product)
    const float inv_w = 1.0f / (float)(width > 0u ? width : 1u);
    while (todo) {
        const int src = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1ull;
        const uint32_t bt = (uint32_t)__builtin_amdgcn_readlane((int)t, src);
        const uint32_t bg = (uint32_t)__builtin_amdgcn_readlane((int)g, src);
        const uint32_t bminx = (uint32_t)__builtin_amdgcn_readlane((int)minx, src);
        const uint32_t bminy = (uint32_t)__builtin_amdgcn_readlane((int)miny, src);
        const uint32_t bw = (uint32_t)__builtin_amdgcn_readlane((int)width, src);
        const float binv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inv_w), src));
        for (uint32_t j = (uint32_t)lane; j < bt; j += 64u) {
            const uint32_t yy = (uint32_t)(((float)j + 0.5f) * binv), xx = j - yy * bw;
            const uint32_t tile = (bminy + yy) * (uint32_t)gx + (bminx + xx) - tile0;  // index inside the band
            if (PLACE) {
                const uint32_t pos = atomicAdd(&cnt[tile], 1u);
                out[pos] = bg;
            } else {
                atomicAdd(&cnt[tile], 1u);
            }
        }
    }
}

template <int NW>
__global__ __launch_bounds__(NW * GSR_WAVE) void tile_count_kernel(const uint32_t *__restrict__ order,
                                                                   const uint32_t *__restrict__ tiles_touched,
                                                                   const uint2 *__restrict__ rects,
                                                                   const GsrHeader *__restrict__ hdr, int gx, int gy,
                                                                   int band_rows, uint32_t *__restrict__ table,
                                                                   int nb_stride) {
    extern __shared__ uint32_t s_cnt[];  // [NW][Tb]
    constexpr int THREADS = NW * GSR_WAVE;
    const uint32_t V = hdr->V;
    const uint32_t row0 = blockIdx.y * (uint32_t)band_rows, row1 = min((uint32_t)gy, row0 + (uint32_t)band_rows);
    const uint32_t tile0 = row0 * (uint32_t)gx;
    const int Tb = (int)((row1 - row0) * (uint32_t)gx);
    // the grid is capped (the launch covers the capacity P, the live count V is usually a fraction of it and dead
    // workgroups are not free to dispatch): a workgroup strides over the 256-rank chunks
    for (uint32_t chunk = blockIdx.x; chunk * (uint32_t)GSR_BLOCK < V; chunk += gridDim.x) {
        const uint32_t base = chunk * (uint32_t)GSR_BLOCK;
        // (gathers fly under the zeroing)
        const WaveSplats mine = load_wave_splats<NW>(order, tiles_touched, rects, V, base, row0, row1);
        for (int i = (int)threadIdx.x; i < NW * Tb; i += THREADS) s_cnt[i] = 0u;
        __syncthreads();
        walk_wave<false>(mine, gx, tile0, s_cnt + gsr_wave() * Tb, nullptr);
        __syncthreads();
        for (int t = (int)threadIdx.x; t < Tb; t += THREADS) {
            uint32_t sum = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) sum += s_cnt[w * Tb + t];
            table[(size_t)(tile0 + (uint32_t)t) * nb_stride + chunk] = sum;
        }
        __syncthreads();  // the counters are zeroed again by the next chunk
    }
}

// one workgroup: totals[T] -> ranges, R; untouched tiles keep (0,0) like the reference's memset
__device__ __forceinline__ void tile_starts_body(const uint32_t *__restrict__ totals, int T,
                                                                GsrHeader *hdr, uint32_t r_capacity,
                                                                uint2 *__restrict__ ranges,
                                                                uint32_t *__restrict__ tile_order,
                                                                uint32_t *__restrict__ cursor_to_zero,
                                                                const uint32_t *__restrict__ quad_work,
                                                                const uint32_t *__restrict__ quad_work_b,
                                                                uint32_t *__restrict__ split_flag,
                                                                uint32_t *__restrict__ split_list,
                                                                uint32_t *__restrict__ split_count, int split_cap,
                                                                uint32_t *__restrict__ quad_order, int cus_per_xcd,
                                                                uint32_t *__restrict__ mirror) {
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_bins[64];
    const bool keyed = tile_order != nullptr && quad_work != nullptr && T <= 32 * GSR_BLOCK;
    // workgroup #2: which quadrants the compositor cuts in two this frame -- the costliest ones of the previous frame
    // on this state, as many as spare waves exist (split_cap), and only those clearly above the average.  A quadrant
    // that WAS split reports its two halves separately; together they did about 1.5x the work of the whole (two cull
    // rectangles, two candidate sweeps).  Depends on nothing of this frame; any choice gives the same image.
    if (blockIdx.x == 2) {
        if (split_flag == nullptr) return;
        const int Q = 4 * T;
        __shared__ uint32_t s_cnt, s_thr;
        auto est = [&](int q) -> uint32_t {
            const uint32_t a = min(quad_work[q], 1u << 24);
            if (split_flag[q] == 0u) return a;  // (last frame's flags: read before they are rewritten below)
            return (uint32_t)(((uint64_t)(a + min(quad_work_b[q], 1u << 24)) * 2u) / 3u);
        };
        uint32_t mx = 0, sum = 0;
        for (int q = (int)threadIdx.x; q < Q; q += GSR_BLOCK) {
            const uint32_t e = est(q);
            mx = max(mx, e);
            sum += e;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
        if (gsr_lane() == 0) s_w[gsr_wave()] = mx;
        if (threadIdx.x < 64) s_bins[threadIdx.x] = 0u;
        __syncthreads();
        mx = max(max(s_w[0], s_w[1]), max(s_w[2], s_w[3]));
        __syncthreads();
        uint32_t total;
        (void)gsr_block_incl_scan(sum, s_w, total);
        const uint32_t floor_cost = (uint32_t)(((uint64_t)total * 115u) / (100u * (uint32_t)max(Q, 1)));  // 1.15 x mean
        const float scale = mx > 0u ? 63.999f / (float)mx : 0.f;
        for (int q = (int)threadIdx.x; q < Q; q += GSR_BLOCK) atomicAdd(&s_bins[(int)((float)est(q) * scale)], 1u);
        __syncthreads();
        if (threadIdx.x == 0) {
            // lowest bucket whose quadrants, with all costlier ones, still fit the spare waves
            uint32_t acc = 0;
            int b = 64;
            while (b > 0 && acc + s_bins[b - 1] <= (uint32_t)max(split_cap, 0)) acc += s_bins[--b];
            s_thr = (uint32_t)b;
            s_cnt = 0u;
        }
        __syncthreads();
        const uint32_t thr = s_thr;
        uint32_t flags[32];  // (Q <= 32 x 256 on this path: T <= 2048)
        int k = 0;
        for (int q = (int)threadIdx.x; q < Q && k < 32; q += GSR_BLOCK, k++) {
            const uint32_t e = est(q);
            flags[k] = (split_cap > 0 && (uint32_t)((float)e * scale) >= thr && e > floor_cost && e > 0u) ? 1u : 0u;
        }
        __syncthreads();  // every estimate above used last frame's flags
        k = 0;
        for (int q = (int)threadIdx.x; q < Q && k < 32; q += GSR_BLOCK, k++) {
            split_flag[q] = flags[k];
            if (flags[k]) split_list[atomicAdd(&s_cnt, 1u)] = (uint32_t)q;
        }
        __syncthreads();
        if (threadIdx.x == 0) *split_count = s_cnt;
        return;
    }
    // two workgroups: #0 turns the totals into ranges, #1 computes the cost order, which depends on nothing of this
    // frame (without keys the order is by list length and has to follow the ranges in #0)
    if (blockIdx.x == 1) {
        if (!keyed) return;
        if (quad_order != nullptr) {
            gsr_quad_order_block(quad_work, 4 * T, quad_order, s_w, cus_per_xcd);
            return;  // (the tile-level order below is what the compositor uses when it has no quadrant order)
        }
        if (T <= 8 * GSR_BLOCK) {
            uint32_t key[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int t = (int)threadIdx.x + i * GSR_BLOCK;
                key[i] = t < T ? gsr_tile_order_key(nullptr, quad_work, t) : 0u;
            }
            gsr_tile_order_block_keys<8>(key, T, tile_order, s_bins, s_w);
        } else {  // (up to 8192 tiles -- 800 x 800 training frames, 1920 x 1080 -- in this workgroup too, instead of
                  //  three sweeps over global memory behind the ranges in workgroup #0)
            uint32_t key[32];
#pragma unroll
            for (int i = 0; i < 32; i++) {
                const int t = (int)threadIdx.x + i * GSR_BLOCK;
                key[i] = t < T ? gsr_tile_order_key(nullptr, quad_work, t) : 0u;
            }
            gsr_tile_order_block_keys<32>(key, T, tile_order, s_bins, s_w);
        }
        return;
    }
    uint32_t grand;
    constexpr int kOwn = 32;  // tiles per thread held in registers: grids up to 8192 tiles in ONE workgroup scan
    if (T <= kOwn * GSR_BLOCK) {
        // a thread owns `per` consecutive tiles: their sum goes through one block scan, the ranges follow from the
        // thread's exclusive base (the loop below costs a scan -- two barriers -- per 256 tiles: 10 of them at 800 x 800)
        const int per = (T + GSR_BLOCK - 1) / GSR_BLOCK, t0 = (int)threadIdx.x * per;
        uint32_t v[kOwn], sum = 0;
#pragma unroll
        for (int k = 0; k < kOwn; k++) {
            v[k] = (k < per && t0 + k < T) ? totals[t0 + k] : 0u;
            sum += v[k];
        }
        uint32_t run = gsr_block_incl_scan(sum, s_w, grand) - sum;
        const bool overflow = grand > r_capacity;
#pragma unroll
        for (int k = 0; k < kOwn; k++) {
            const int t = t0 + k;
            if (k < per && t < T) {
                ranges[t] = (v[k] == 0u || overflow) ? make_uint2(0u, 0u) : make_uint2(run, run + v[k]);
                if (cursor_to_zero) cursor_to_zero[t] = 0u;
            }
            run += v[k];
        }
        if (threadIdx.x == 0) {
            hdr->R_raw = grand;
            hdr->r_capacity = r_capacity;
            gsr_set_overflow(hdr, overflow, mirror);
            hdr->R = overflow ? 0u : grand;
        }
        if (tile_order == nullptr || keyed) return;
        __syncthreads();  // this workgroup's range stores are visible to all of its threads
        gsr_tile_order_block(ranges, T, tile_order, s_bins, s_w, quad_work);
        return;
    }
    uint32_t sum = 0;
    for (int t = (int)threadIdx.x; t < T; t += GSR_BLOCK) sum += totals[t];
    gsr_block_incl_scan(sum, s_w, grand);
    const bool overflow = grand > r_capacity;
    uint32_t carry = 0;
    for (int base = 0; base < T; base += GSR_BLOCK) {
        const int t = base + (int)threadIdx.x;
        const uint32_t v = t < T ? totals[t] : 0u;
        uint32_t total;
        const uint32_t incl = gsr_block_incl_scan(v, s_w, total);
        if (t < T) {
            const uint32_t start = carry + incl - v;
            ranges[t] = (v == 0u || overflow) ? make_uint2(0u, 0u) : make_uint2(start, start + v);
            if (cursor_to_zero) cursor_to_zero[t] = 0u;
        }
        carry += total;
    }
    if (threadIdx.x == 0) {
        hdr->R_raw = grand;
        hdr->r_capacity = r_capacity;
        gsr_set_overflow(hdr, overflow, mirror);
        hdr->R = overflow ? 0u : grand;
    }
    if (tile_order == nullptr || keyed) return;
    __syncthreads();  // this workgroup's range stores are visible to all of its threads
    gsr_tile_order_block(ranges, T, tile_order, s_bins, s_w, quad_work);
}

// grid = (1 .. 3 workgroups, frames): blockIdx.y picks the frame's argument block (gsr_internal.h GsrBatch)
struct TileStartsArgs {
    const uint32_t *totals;
    int T;
    GsrHeader *hdr;
    uint32_t r_capacity;
    uint2 *ranges;
    uint32_t *tile_order, *cursor_to_zero;
    const uint32_t *quad_work, *quad_work_b;
    uint32_t *split_flag, *split_list, *split_count;
    int split_cap;
    uint32_t *quad_order;
    int cus_per_xcd;
    uint32_t *mirror;  // (GsrOutputs.overflow_mirror or nullptr)
};
__global__ __launch_bounds__(GSR_BLOCK) void tile_starts_kernel(const GsrBatch<TileStartsArgs> bt) {
    const TileStartsArgs &a = bt.f[blockIdx.y];
    tile_starts_body(a.totals, a.T, a.hdr, a.r_capacity, a.ranges, a.tile_order, a.cursor_to_zero, a.quad_work,
                     a.quad_work_b, a.split_flag, a.split_list, a.split_count, a.split_cap, a.quad_order, a.cus_per_xcd,
                     a.mirror);
}

template <int NW>
__global__ __launch_bounds__(NW * GSR_WAVE) void tile_place_kernel(const uint32_t *__restrict__ order,
                                                                   const uint32_t *__restrict__ tiles_touched,
                                                                   const uint2 *__restrict__ rects,
                                                                   const GsrHeader *__restrict__ hdr, int gx, int gy,
                                                                   int band_rows, const uint32_t *__restrict__ table,
                                                                   int nb_stride, const uint2 *__restrict__ ranges,
                                                                   uint32_t *__restrict__ point_list) {
    extern __shared__ uint32_t s_cnt[];  // [NW][Tb]: counts, then running cursors
    constexpr int THREADS = NW * GSR_WAVE;
    const uint32_t V = hdr->V;
    if (hdr->overflow) return;
    const uint32_t row0 = blockIdx.y * (uint32_t)band_rows, row1 = min((uint32_t)gy, row0 + (uint32_t)band_rows);
    const uint32_t tile0 = row0 * (uint32_t)gx;
    const int Tb = (int)((row1 - row0) * (uint32_t)gx);
    const int wave = gsr_wave();
    for (uint32_t chunk = blockIdx.x; chunk * (uint32_t)GSR_BLOCK < V; chunk += gridDim.x) {  // (capped grid)
        const uint32_t base = chunk * (uint32_t)GSR_BLOCK;
        const WaveSplats mine = load_wave_splats<NW>(order, tiles_touched, rects, V, base, row0, row1);
        for (int i = (int)threadIdx.x; i < NW * Tb; i += THREADS) s_cnt[i] = 0u;
        __syncthreads();
        walk_wave<false>(mine, gx, tile0, s_cnt + wave * Tb, nullptr);
        __syncthreads();
        for (int t = (int)threadIdx.x; t < Tb; t += THREADS) {
            uint32_t c[NW], any = 0u;
#pragma unroll
            for (int w = 0; w < NW; w++) {
                c[w] = s_cnt[w * Tb + t];
                any |= c[w];
            }
            if (any != 0u) {
                const uint32_t tile = tile0 + (uint32_t)t;
                uint32_t s = ranges[tile].x + table[(size_t)tile * nb_stride + chunk];
#pragma unroll
                for (int w = 0; w < NW; w++) {
                    s_cnt[w * Tb + t] = s;
                    s += c[w];
                }
            }
        }
        __syncthreads();
        walk_wave<true>(mine, gx, tile0, s_cnt + wave * Tb, point_list);
        __syncthreads();  // the cursors are zeroed again by the next chunk
    }
}


// ---------------------------------------------------------------------------------------------------------
// Bin-then-sort placement (default path).
//
// No global depth sort at all: preprocess accumulates per-tile instance totals (LDS histogram per workgroup +
// one global add per touched tile), tile_starts turns them into ranges, bin_scatter drops every instance into
// its tile's segment in ARBITRARY order as the 64-bit key (depth bits << 32 | Gaussian index) -- a workgroup
// reserves a contiguous piece of each touched segment with one returning global atomic and ranks inside it with
// LDS atomics -- and tile_sort orders each segment by that key in LDS.  Sorting the key is exactly the
// reference's order inside a tile (ascending depth bits, ascending index on ties), so the point list is still
// bit-identical, while the 9-launch radix chain and the index-ordered compaction disappear and the work per
// workgroup no longer grows with tiles x Gaussians.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GSR_BLOCK) void bin_scatter_kernel(int P, const uint32_t *__restrict__ tiles_touched,
                                                                const uint2 *__restrict__ rects,
                                                                const float4 *__restrict__ splat,
                                                                const uint2 *__restrict__ ranges,
                                                                uint32_t *__restrict__ tile_cursor,
                                                                const GsrHeader *__restrict__ hdr, int gx, int T,
                                                                uint64_t *__restrict__ keys) {
    extern __shared__ uint32_t s_mem[];  // [T] counts -> ranks, [T] segment bases
    uint32_t *s_cnt = s_mem, *s_base = s_mem + T;
    if (hdr->overflow) return;
    const int i = blockIdx.x * GSR_BLOCK + (int)threadIdx.x;
    const uint32_t t = i < P ? tiles_touched[i] : 0u;
    uint2 rc = make_uint2(0u, 0u);
    uint32_t dbits = 0u;
    if (t != 0u) {
        rc = rects[i];
        dbits = __float_as_uint(splat[3 * (size_t)i].z);
    }
    if (__syncthreads_count(t != 0u) == 0) return;
    for (int k = (int)threadIdx.x; k < T; k += GSR_BLOCK) s_cnt[k] = 0u;
    __syncthreads();
    gsr_for_each_tile(t, rc, gx, 0u, 0u, [s_cnt](uint32_t tile, uint32_t, uint32_t) { atomicAdd(&s_cnt[tile], 1u); });
    __syncthreads();
    // reserve this workgroup's piece of every touched segment: 4 returning atomics in flight per thread
    for (int k0 = (int)threadIdx.x; k0 < T; k0 += 4 * GSR_BLOCK) {
        uint32_t c[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int k = k0 + u * GSR_BLOCK;
            c[u] = k < T ? s_cnt[k] : 0u;
            b[u] = 0u;
        }
        uint32_t *cur = tile_cursor + (size_t)(blockIdx.x % GSR_BIN_SLOTS) * T;  // this workgroup's counter copy
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (c[u] != 0u) b[u] = atomicAdd(&cur[k0 + u * GSR_BLOCK], c[u]);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int k = k0 + u * GSR_BLOCK;
            if (c[u] != 0u) {
                s_base[k] = b[u];  // cursors hold absolute positions in the instance list
                s_cnt[k] = 0u;
            }
        }
    }
    __syncthreads();
    gsr_for_each_tile(t, rc, gx, (uint32_t)i, dbits, [=](uint32_t tile, uint32_t lo, uint32_t hi) {
        const uint32_t pos = s_base[tile] + atomicAdd(&s_cnt[tile], 1u);
        keys[pos] = ((uint64_t)hi << 32) | lo;
    });
}

// one workgroup: slot-replicated tile totals -> ranges, R, overflow, tile order; cursor[s][t] = first slot of
// the piece of tile t's segment that the workgroups of slot s fill
__global__ __launch_bounds__(GSR_BLOCK) void bin_starts_kernel(const uint32_t *__restrict__ accum, int T,
                                                               GsrHeader *hdr, uint32_t r_capacity,
                                                               uint2 *__restrict__ ranges,
                                                               uint32_t *__restrict__ tile_order,
                                                               uint32_t *__restrict__ cursor) {
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_bins[64];
    uint32_t sum = 0;
    for (int t = (int)threadIdx.x; t < T; t += GSR_BLOCK)
#pragma unroll
        for (int s = 0; s < GSR_BIN_SLOTS; s++) sum += accum[(size_t)s * T + t];
    uint32_t grand;
    gsr_block_incl_scan(sum, s_w, grand);
    const bool overflow = grand > r_capacity;
    uint32_t carry = 0;
    for (int base = 0; base < T; base += GSR_BLOCK) {
        const int t = base + (int)threadIdx.x;
        uint32_t part[GSR_BIN_SLOTS], v = 0;
#pragma unroll
        for (int s = 0; s < GSR_BIN_SLOTS; s++) {
            part[s] = t < T ? accum[(size_t)s * T + t] : 0u;
            v += part[s];
        }
        uint32_t total;
        const uint32_t incl = gsr_block_incl_scan(v, s_w, total);
        if (t < T) {
            const uint32_t start = carry + incl - v;
            ranges[t] = (v == 0u || overflow) ? make_uint2(0u, 0u) : make_uint2(start, start + v);
            uint32_t run = start;
#pragma unroll
            for (int s = 0; s < GSR_BIN_SLOTS; s++) {
                cursor[(size_t)s * T + t] = run;
                run += part[s];
            }
        }
        carry += total;
    }
    if (threadIdx.x == 0) {
        hdr->R_raw = grand;
        hdr->r_capacity = r_capacity;
        gsr_set_overflow(hdr, overflow);
        hdr->R = overflow ? 0u : grand;
    }
    __syncthreads();
    gsr_tile_order_block(ranges, T, tile_order, s_bins, s_w);
}

constexpr int kSortLds = 8192;  // keys held in LDS (64 KiB); longer tile lists are sorted in place in global memory

__global__ __launch_bounds__(GSR_BLOCK) void tile_sort_kernel(const uint2 *__restrict__ ranges,
                                                              const uint32_t *__restrict__ tile_order,
                                                              uint64_t *__restrict__ keys,
                                                              uint32_t *__restrict__ point_list) {
    extern __shared__ uint64_t s_keys[];
    const int tile = (int)tile_order[blockIdx.x];  // longest lists first
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    if (n == 0) return;
    int N = 2;
    while (N < n) N <<= 1;
    uint64_t *seg = keys + range.x;
    if (n == 1) {
        if (threadIdx.x == 0) point_list[range.x] = (uint32_t)seg[0];
        return;
    }
    if (N <= kSortLds) {
        for (int i = (int)threadIdx.x; i < n; i += GSR_BLOCK) s_keys[i] = seg[i];
        __syncthreads();
        bitonic_sort_block(s_keys, n, N);
        for (int i = (int)threadIdx.x; i < n; i += GSR_BLOCK) point_list[range.x + i] = (uint32_t)s_keys[i];
    } else {
        __syncthreads();
        bitonic_sort_block(seg, n, N);  // same network on global memory (one workgroup: barriers order the passes)
        for (int i = (int)threadIdx.x; i < n; i += GSR_BLOCK) point_list[range.x + i] = (uint32_t)seg[i];
    }
}

}  // namespace

// default path, part 1: per-tile instance counts -> ranges and R (no instance buffer needed yet)
int gsr_launch_tile_count(const GsrSettings &st, int32_t P, const GeomState &g, const ImageState &img,
                          uint32_t r_capacity, bool debug, hipStream_t stream) {
    const int gx = gsr_div_up(st.image_width, GSR_TILE), gy = gsr_div_up(st.image_height, GSR_TILE);
    const int T = gx * gy;
    const int nb = GeomState::prep_blocks(P);
    const int band_rows = place_band_rows(gx, gy);
    const int bands = gsr_div_up(gy, band_rows);
    const size_t lds = (size_t)kPlaceWaves * band_rows * gx * sizeof(uint32_t);
    hipLaunchKernelGGL(tile_count_kernel<kPlaceWaves>, dim3(nb < kPlaceGrid ? nb : kPlaceGrid, bands),
                       dim3(kPlaceWaves * GSR_WAVE), lds, stream, g.order, g.tiles_touched, g.rects, g.hdr, gx, gy,
                       band_rows, g.tile_table, nb);
    if (int e = gsr_check_launch("tile_count", debug, stream)) return e;
    // one table column per 256 depth ranks: the live row length is ceil(V / 256)
    if (int e = gsr_launch_rowscan(g.tile_table, &g.hdr->V, nb, GSR_BLOCK, T, g.tile_totals, debug, stream)) return e;
    return gsr_launch_tile_starts(st, g, img, r_capacity, false, debug, stream);
}

// per-tile totals -> ranges, R, capacity check, compositing order (shared by the counting placements)
int gsr_launch_tile_starts(int B, const GsrFrame *fr, bool order_done, bool debug, hipStream_t stream) {
    const GsrSettings &st = *fr[0].st_bin;
    const int T = gsr_div_up(st.image_width, GSR_TILE) * gsr_div_up(st.image_height, GSR_TILE);
    const int split_blocks = T <= 2048 ? gsr_render_split_blocks(st, T) : 0;
    GsrBatch<TileStartsArgs> bt{};  // (entries beyond B stay zero: nothing uninitialised travels in the kernarg)
    for (int k = 0; k < B; k++) {
        const GeomState &g = fr[k].g;
        const ImageState &img = fr[k].img;
        TileStartsArgs &a = bt.f[k];
        a.totals = g.tile_totals;
        a.T = T;
        a.hdr = g.hdr;
        a.r_capacity = fr[k].cap32;
        a.ranges = img.ranges;
        a.tile_order = !order_done && gsr_render_wants_tile_order(st, T) ? img.tile_order : (uint32_t *)nullptr;
        a.cursor_to_zero = nullptr;
        a.quad_work = img.quad_work;
        a.quad_work_b = img.quad_work_b;
        a.split_flag = img.split_flag;
        a.split_list = img.split_list;
        a.split_count = img.split_count;
        a.split_cap = split_blocks * (GSR_BLOCK / GSR_WAVE);
        a.quad_order = !order_done && gsr_render_uses_quad_order(st, T) ? img.quad_order : (uint32_t *)nullptr;
        a.cus_per_xcd = gsr_render_cus_per_xcd();
        a.mirror = fr[k].out ? fr[k].out->overflow_mirror : (uint32_t *)nullptr;
    }
    // (order_done: the compositing order was computed earlier in the frame, beside the depth sort's partition pass)
    hipLaunchKernelGGL(tile_starts_kernel, dim3(split_blocks > 0 ? 3 : (order_done ? 1 : 2), B), dim3(GSR_BLOCK), 0, stream,
                       bt);
    return gsr_check_launch("tile_starts", debug, stream);
}
int gsr_launch_tile_starts(const GsrSettings &st, const GeomState &g, const ImageState &img, uint32_t r_capacity,
                           bool order_done, bool debug, hipStream_t stream) {
    GsrFrame f{};
    f.st = &st;
    f.st_bin = &st;
    f.g = g;
    f.img = img;
    f.cap32 = r_capacity;
    return gsr_launch_tile_starts(1, &f, order_done, debug, stream);
}

// default path, part 2: write the point list (tile-major, depth order, index order on ties)
int gsr_launch_tile_place(const GsrSettings &st, int32_t P, const GeomState &g, const BinningState &b,
                          const ImageState &img, bool debug, hipStream_t stream) {
    const int gx = gsr_div_up(st.image_width, GSR_TILE), gy = gsr_div_up(st.image_height, GSR_TILE);
    const int nb = GeomState::prep_blocks(P);
    const int band_rows = place_band_rows(gx, gy);
    const int bands = gsr_div_up(gy, band_rows);
    const size_t lds = (size_t)kPlaceWaves * band_rows * gx * sizeof(uint32_t);
    hipLaunchKernelGGL(tile_place_kernel<kPlaceWaves>, dim3(nb < kPlaceGrid ? nb : kPlaceGrid, bands),
                       dim3(kPlaceWaves * GSR_WAVE), lds, stream, g.order, g.tiles_touched, g.rects, g.hdr, gx, gy,
                       band_rows, g.tile_table, nb, img.ranges, b.gidx[0]);
    return gsr_check_launch("tile_place", debug, stream);
}

// bin-then-sort path, part 1: tile totals (accumulated by preprocess) -> ranges, R, tile order; cursors zeroed
int gsr_launch_bin_starts(const GsrSettings &st, const GeomState &g, const ImageState &img, uint32_t r_capacity,
                          bool debug, hipStream_t stream) {
    const int T = gsr_div_up(st.image_width, GSR_TILE) * gsr_div_up(st.image_height, GSR_TILE);
    hipLaunchKernelGGL(bin_starts_kernel, dim3(1), dim3(GSR_BLOCK), 0, stream, g.tile_accum, T, g.hdr, r_capacity,
                       img.ranges, img.tile_order, g.tile_cursor);
    return gsr_check_launch("bin_starts", debug, stream);
}

// bin-then-sort path, part 2: unordered scatter of the 64-bit keys, then the per-tile sort -> point list
int gsr_launch_bin_scatter_and_sort(const GsrSettings &st, int32_t P, const GeomState &g, const BinningState &b,
                                    const ImageState &img, bool debug, hipStream_t stream) {
    const int gx = gsr_div_up(st.image_width, GSR_TILE), gy = gsr_div_up(st.image_height, GSR_TILE);
    const int T = gx * gy;
    hipLaunchKernelGGL(bin_scatter_kernel, dim3(GeomState::prep_blocks(P)), dim3(GSR_BLOCK),
                       (size_t)2 * T * sizeof(uint32_t), stream, P, g.tiles_touched, g.rects, g.splat, img.ranges,
                       g.tile_cursor, g.hdr, gx, T, b.keys64);
    if (int e = gsr_check_launch("bin_scatter", debug, stream)) return e;
    hipLaunchKernelGGL(tile_sort_kernel, dim3(T), dim3(GSR_BLOCK), (size_t)kSortLds * sizeof(uint64_t), stream,
                       img.ranges, img.tile_order, b.keys64, b.gidx[0]);
    return gsr_check_launch("tile_sort", debug, stream);
}

int gsr_launch_emit_and_tile_sort(const GsrSettings &st, int32_t P, const GeomState &g, const BinningState &b,
                                  const ImageState &img, int64_t r_capacity, bool debug, hipStream_t stream) {
    const int gx = gsr_div_up(st.image_width, GSR_TILE), gy = gsr_div_up(st.image_height, GSR_TILE);
    const int tiles = gx * gy;
    hipLaunchKernelGGL(zero_ranges_kernel, dim3(gsr_div_up(tiles, GSR_BLOCK)), dim3(GSR_BLOCK), 0, stream,
                       img.ranges, tiles);
    if (int e = gsr_check_launch("zero_ranges", debug, stream)) return e;
    hipLaunchKernelGGL(emit_kernel, dim3(GeomState::sort_blocks(P)), dim3(GSR_BLOCK), 0, stream, g.order,
                       g.tiles_touched, g.rects, g.tile_bsum, g.hdr, gx, b.tile[0], b.gidx[0]);
    if (int e = gsr_check_launch("emit", debug, stream)) return e;
    uint32_t *key[2] = {b.tile[0], b.tile[1]};
    uint32_t *val[2] = {b.gidx[0], b.gidx[1]};
    const int bits = BinningState::tile_bits(tiles);
    if (int e = gsr_radix_sort_u32(key, val, &g.hdr->R, r_capacity, bits, GSR_RADIX_BITS, 0, b.sort_table,
                                   b.sort_totals, debug, stream))
        return e;
    const int side = BinningState::tile_passes(tiles) & 1;
    hipLaunchKernelGGL(tile_ranges_kernel, dim3(gsr_div_up(r_capacity > 0 ? r_capacity : 1, GSR_BLOCK)),
                       dim3(GSR_BLOCK), 0, stream, b.tile[side], g.hdr, img.ranges);
    if (int e = gsr_check_launch("tile_ranges", debug, stream)) return e;
    if (side == 1) {
        // the point list always ends in side 0 (what render / backward / state views read)
        const size_t n = (size_t)(r_capacity > 0 ? r_capacity : 1) * sizeof(uint32_t);
        if (hipMemcpyAsync(b.gidx[0], b.gidx[1], n, hipMemcpyDeviceToDevice, stream) != hipSuccess) {
            gsr_set_error("tile sort: result copy failed");
            return GSR_E_HIP;
        }
    }
    return GSR_OK;
}
