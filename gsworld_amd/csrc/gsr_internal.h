// gsr_internal.h -- state layout and wave64 helpers shared by the HIP translation units of libgsr_hip.so.
// gfx950 (MI355X) only: 64-wide wavefronts are assumed everywhere.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/gsr.h"

#define GSR_WAVE 64
#define GSR_BLOCK 256                 // threads per workgroup for every streaming kernel (4 waves)
#define GSR_SORT_ITEMS 8              // keys per thread per radix block
#define GSR_SORT_CHUNK (GSR_BLOCK * GSR_SORT_ITEMS)
#define GSR_RADIX_BITS 8              // digit width of the tile-id sort (large-grid fallback)
#define GSR_RADIX_BINS 256
#define GSR_DEPTH_RADIX_BITS 11       // digit width of the 32-bit depth sort and the 30-bit Morton sort: 3 passes
#define GSR_DEPTH_RADIX_BINS 2048
#define GSR_BIN_SLOTS 16              // replicated per-tile counters of the bin-then-sort path
#ifndef GSR_SS_PER_BUCKET
#define GSR_SS_PER_BUCKET 512          // depth sort: records per bucket the bucket count aims at (B = 256 .. 2048) in a
#endif                                 // frame that samples ...
#ifndef GSR_SS_PER_BUCKET_FULL
#define GSR_SS_PER_BUCKET_FULL 1024    // ... and in one that takes the kept exact quantiles unchecked (depthsort.hip ss_prepare)
#endif
#ifndef GSR_COOP_MAX_FRAMES
#define GSR_COOP_MAX_FRAMES 8          // cooperative quadrants (render.hip): launches of at most this many frames (= all) ...
#endif
#ifndef GSR_COOP_MAX_BLOCKS
#define GSR_COOP_MAX_BLOCKS 64         // ... get up to this many of them (a multiple of 8: as many per XCD) ...
#endif
#ifndef GSR_COOP_FACTOR_X16
#define GSR_COOP_FACTOR_X16 40         // ... whose cost in the previous frame was above 40 / 16 of the mean quadrant's
#endif
#define GSR_BAND_RANGES 64            // band placement (bandplace.hip): depth-rank ranges per tile row
// forward_only frames bin per (2^SX x 2^SY)-tile super-tile.  Measured at config 2 (one frame at a time; default
// per-tile binning 195 us): 2 x 1 tiles (32 x 16 px) 168 us -- 0.58 of the instances, compositor unchanged at 54 us;
// 2 x 2 179 us -- 0.34 of the instances, but the compositor 70 us (a quadrant then walks every nearer splat of its
// vertical neighbour before it reaches its own); 1 x 2 175 us; 4 x 1 176 us.
#ifndef GSR_SUPER_SX
#define GSR_SUPER_SX 1
#endif
#ifndef GSR_SUPER_SY
#define GSR_SUPER_SY 0
#endif
#define GSR_MAX_COUNT_TILES 16384     // counting placement: tile_table is tiles x ceil(P/256) words (bands keep LDS <= 40 KiB)

// Frame header, first 256 bytes of the geometry state.  Lives on the device so that no kernel launch
// depends on a value the host would have to read back.
struct GsrHeader {
    uint32_t V;           // visible Gaussians (radii > 0)
    uint32_t R;           // rendered instances actually binned (0 when overflowed)
    uint32_t overflow;    // R_raw > r_capacity
    uint32_t r_capacity;
    uint32_t R_raw;       // sum of tiles touched, even when it overflowed
    uint32_t tile_queue;  // ticket counter of the compositing kernel's tile queue (zeroed with the header)
    uint32_t ss_magic;    // depth sort: the splitters in the state are the exact quantiles of the last frame ...
    uint32_t ss_buckets;  // ... for this bucket count (both survive from frame to frame; garbage on a fresh state)
    uint32_t ss_bad;      // a bucket of the last frame came out above what exact quantiles of an UNCHANGED scene give
    uint32_t ss_view[16]; // bits of the view matrix the splitters were built under
    uint32_t ss_blind;    // this frame's compaction took the splitters unchecked (static camera)
    uint32_t br_magic;    // band placement: wave_lo_base holds exact equal-cost cuts of a depth order of ...
    uint32_t br_V;        // ... this many visible Gaussians, computed ...
    uint32_t br_age;      // ... this many frames ago
    uint32_t ss_P;        // model size the splitters / cuts belong to (the arrays move with P; a recycled buffer may
    uint32_t br_P;        //   carry an old header over new garbage: both users also check what they read)
    uint32_t ss_trust;    // consecutive frames that classified with the KEPT table and came out balanced (blind needs 2)
    uint32_t ss_prev_fresh;  // the previous frame drew its own splitters (says nothing about the kept table)
    uint32_t ss_fresh;    // this frame's compaction drew new splitters: the partition pass reads ss_splitters_new
    uint32_t ss_B;        // depth buckets of THIS frame (chosen by ss_prepare: every later kernel of the frame reads it)
    uint32_t ss_stride;   // 2: this (blind) frame takes every second entry of a kept table of 2 ss_B quantiles; else 1
    uint32_t coop_quads;  // quadrants the compositor of this frame hands to cooperative workgroups (render.hip)
    uint32_t br_route;    // band placement: how this frame's counting pass came by its cuts (bandplace.hip band_cuts_block)
    uint32_t coop_timeouts;  // cooperative quadrants whose hand-off timed out in the frames on this state (render.hip;
                          //   never cleared by a frame: reported like overflow_frames -- such a quadrant is truncated)
    uint32_t coop_timeout_now;  // ... and in THIS frame (cleared by the frame's first kernel: gsr_frame_stats -> GSR_E_TRUNCATED)
    uint32_t ss_moved;    // this frame's view matrix is not the one the kept splitters were built under (depthsort.hip)
    uint32_t ss_near;     // ... and the frame took them unchecked all the same (a camera that moves a little)
    uint32_t ss_near_fail;  // such frames that came out unbalanced (each doubles the trust the next one has to show)
    uint32_t ss_vfail;    // frames since a check of the kept table against samples failed (0: the last one passed)
    // block cache of preprocess (preprocess.hip prep_block_cached): the previous frame on this state left its camera and pose
    // table in GeomState::pc_slots[pc_parity] and every per-Gaussian record it computed is still in place
    uint32_t pc_magic;    // GSR_PC_MAGIC: ... and they are valid for pc_sig
    uint32_t pc_parity;   // which slot holds them
    uint32_t pc_pending;  // this frame's preprocess has written the other slot (ss_prepare flips)
    uint32_t pc_sig;      // settings + model signature of the frame that left the valid slot
    uint32_t pc_sig_next; // ... of the frame that wrote the pending one
    uint32_t pc_hit;      // some workgroup of this frame's preprocess kept its block (plain stores of 1) ...
    uint32_t pc_hit_last; // ... as ss_prepare found it: did the last frame on this state keep any block? (gsr_debug_sort_state)
    // tile reuse (render.hip): a tile no recomputed Gaussian touches -- now or in the previous frame -- keeps that frame's pixels
    uint32_t td_token;    // ImageState::tile_dirty[t] == td_token: tile t has to be composited in this frame
    uint32_t td_reuse;    // this frame may skip the others (same camera, background and output buffer; the previous frame complete)
    uint32_t td_skipped;  // (diagnostics: some tile of the last frame was skipped)
    uint32_t pad[8];
    uint32_t of_magic;    // overflow_frames below is a count (anything else: a fresh / recycled buffer, count = 0)
    uint32_t overflow_frames;  // frames rendered on this state whose R exceeded the capacity (never cleared by a frame:
                               //   a no-sync rollout learns at its end whether EVERY frame was valid)
};
#define GSR_OF_MAGIC 0x0F10F10Fu
#ifndef GSR_TILE_REUSE
#define GSR_TILE_REUSE 1  // (A/B: 0 = every tile composited in every frame)
#endif
#ifndef GSR_PREP_BLOCK_CACHE
#define GSR_PREP_BLOCK_CACHE 1  // (A/B: 0 = every frame recomputes every block, whatever its caller promises)
#endif
#define GSR_PC_MAGIC 0x50434348u  // 'PCCH'
#define GSR_PC_MAX_PARTS 64
#define GSR_PC_SLOT (40 + 17 * GSR_PC_MAX_PARTS)  // floats of one slot: view 16 | proj 16 | campos 3 | pad 5 | pose table
// The one thread that compared R with the capacity.  Every frame clears hdr->overflow before its first such check, and a
// path with two checks per frame (bin-then-sort) must count the frame once: only the 0 -> 1 edge counts.
// `mirror` (GsrOutputs.overflow_mirror: two words the HOST can read, or nullptr) receives (of_magic, overflow_frames) as
// they stand after this frame's check -- what callers used to fetch with an 8-byte copy behind every frame.
__device__ inline void gsr_set_overflow(GsrHeader *hdr, bool overflow, uint32_t *mirror = nullptr) {
    const bool was = hdr->overflow != 0u;
    hdr->overflow = overflow ? 1u : 0u;
    uint32_t magic = hdr->of_magic, count = hdr->overflow_frames;
    if (overflow && !was) {
        if (magic != GSR_OF_MAGIC) {
            hdr->of_magic = magic = GSR_OF_MAGIC;
            count = 0u;
            hdr->coop_timeouts = 0u;
        }
        hdr->overflow_frames = ++count;
    }
    if (mirror != nullptr) {
        mirror[1] = magic == GSR_OF_MAGIC ? count : 0u;
        mirror[0] = magic;
    }
}
// A cooperative quadrant of the compositor gave up waiting for a hand-off (render.hip render_coop_quadrant): its pixels are
// truncated.  Counted under the same validity word as the overflow count (a header that never held a count reads as 0);
// several workgroups may report at once.
__device__ inline void gsr_note_coop_timeout(GsrHeader *hdr) {
    if (hdr->of_magic != GSR_OF_MAGIC) {
        hdr->overflow_frames = 0u;
        hdr->coop_timeouts = 0u;
        __threadfence();
        hdr->of_magic = GSR_OF_MAGIC;
    }
    atomicAdd(&hdr->coop_timeouts, 1u);
    hdr->coop_timeout_now = 1u;
}
static_assert(sizeof(GsrHeader) == 256, "header is one 256-byte line");

static inline size_t gsr_align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline int gsr_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// sample sort geometry (depthsort.hip): compaction workgroups and bucket capacity for P Gaussians
int gsr_ss_nbc(int32_t P);
int gsr_ss_bmax(int32_t P);
bool gsr_ss_supported(int32_t P);

#ifndef GSR_GRAD_F64
#define GSR_GRAD_F64 1  // (A/B: 0 = the per-Gaussian gradient sums of the backward accumulate in binary32, rounds 1-5)
#endif
#if GSR_GRAD_F64
typedef double GsrGradWord;
#else
typedef float GsrGradWord;
#endif

// ---- geometry state (per Gaussian) -------------------------------------------------------------------
struct GeomState {
    GsrHeader *hdr;
    uint32_t *tile_accum;     // [tiles] per-tile instance totals accumulated by preprocess (bin-then-sort path);
                              //         sits right behind the header so that one memset clears both
    uint32_t *tile_cursor;    // [tiles] segment cursors of the unordered binning
    float4 *splat;            // [3P]  rec0 = (px, py, depth, 1/depth) rec1 = (conic.x, conic.y, conic.z, opacity)
                              //       rec2 = (r, g, b, radius as float)
    float *cov3D;             // [6P]
    uint32_t *clamped;        // [P]   byte c = SH clamp flag of channel c
    uint2 *block_recs;        // [P]   (index, depth bits) of every block's visible Gaussians, compacted to the head of the
                              //       block's own 256 slots (preprocess -> the sample sort's compaction).  An array of its own
                              //       since round 6 (it used to be pair[1], which the sort's partition pass overwrites): a block
                              //       the next frame does not recompute keeps its records
    float *pc_slots;          // [2][GSR_PC_SLOT] camera + pose table of the last two frames' preprocess (GsrHeader::pc_*)
    GsrGradWord *grad_rec;    // [12P] backward only: the compositor's ten per-Gaussian sums as ONE record (mean2D.xy,
                              //       conic xx xy yy, opacity, rgb, 1/depth, 2 unused) -- see backward.hip.  binary64 words
                              //       (round 6): a Gaussian's sums arrive tile by tile, by device-scope atomics in whatever
                              //       order the workgroups retire -- in binary32 that order was worth up to 1.7e-3 of a
                              //       gradient's scale from run to run at configs[4] size
    uint32_t *tiles_touched;  // [P]
    uint2 *rects;             // [P]   x = min.x | min.y<<16, y = max.x | max.y<<16
    uint32_t *block_counts;   // [ceil(P/256)+1]  visible per preprocess block -> exclusive offsets
    uint2 *pair[2];           // [P]   depth-sort ping-pong records (float bits of depth, Gaussian index)
    uint32_t *order;          // [P]   Gaussian indices in depth order (depth bits, index on ties)
    uint32_t *sort_table;     // [256 * nb]  per-block digit histograms (digit-major)
    uint32_t *sort_totals;    // [256]
    uint32_t *tile_bsum;      // [nb+1] tiles touched per 2048 depth-ordered Gaussians -> exclusive offsets
    uint32_t *tile_table;     // [tiles * prep_blocks]  per-workgroup tile histograms (counting placement)
    uint32_t *tile_totals;    // [tiles]
    // sample sort of the depth keys (depthsort.hip)
    uint32_t *vis_key;        // (round 2: depth bits of a visible Gaussian per index; not carved any more)
    uint32_t *block_cand;     // [ceil(P/256)]  depth bits of the first visible Gaussian of every preprocess block
    uint32_t *ss_table;       // [nbc * bmax]   bucket histogram of every compaction workgroup
    uint32_t *ss_splitters;   // [bmax]  last frame's exact quantiles (written by ss_buckets only: never while it is read)
    uint32_t *ss_totals;      // [bmax]  records per bucket (column totals of ss_table, ss_colscan)
    uint32_t *ss_splitters_new; // [bmax] the table a sampling frame draws (written by ss_compact, read by ss_partition)
    uint32_t *ss_bucket_start;// [bmax + 1]
    uint32_t *ss_seg;         // [nbc + 1]      first output slot of every compaction workgroup
    uint32_t *ss_first;       // [nbc + 1]      first preprocess block of every compaction workgroup (ss_prepare)
    uint64_t *ss_dbg;         // [64] cycle stamps of workgroup 0 (builds with -DGSR_SS_TIMING only)
    // band placement (bandplace.hip)
    uint2 *rect_sorted;       // [P]   tile rects in depth order (written by the depth sort)
    uint32_t *band_table;     // [tiles * GSR_BAND_RANGES]  instances per (tile, rank range) -> exclusive offsets
    uint32_t *band_wtable;    // [tile rows * GSR_BAND_RANGES * 4 waves * padded row width]  per-wave column counts
    uint32_t *tile_cum;       // [P]   tiles touched, inclusive running sum in depth order INSIDE each sort bucket
    uint32_t *bucket_tiles;   // [bmax] tiles touched per sort bucket
    uint32_t *wave_lo;        // [GSR_BAND_RANGES * 4 + 1] first depth rank of every placement wave (equal cost shares)
    uint32_t *wave_lo_base;   // [GSR_BAND_RANGES * 4 + 1] the last exactly computed cuts (rescaled while the camera rests)
    uint32_t *band_nseg;      // [tile rows * GSR_BAND_RANGES] column segments (1 / 2 / 4) of the four placement waves of
                              //   every (tile row, rank range), one byte per wave: written by the counting pass

    static int sort_blocks(int32_t P) { return gsr_div_up(P > 0 ? P : 1, GSR_SORT_CHUNK); }
    static int prep_blocks(int32_t P) { return gsr_div_up(P > 0 ? P : 1, GSR_BLOCK); }

    template <typename T>
    static T *take(char *&p, size_t count) {
        T *r = reinterpret_cast<T *>(p);
        p += gsr_align_up(count * sizeof(T));
        return r;
    }
    static bool counting(int tiles) { return tiles <= GSR_MAX_COUNT_TILES; }
    // words of band_wtable for a grid `tiles_x` wide (0: band placement not used for this grid)
    static size_t band_wtable_words(int tiles_x, int tiles) {
        if (tiles_x <= 0 || tiles_x > 256) return 1;
        const int rows = tiles / tiles_x, padded = tiles_x <= 64 ? 64 : (tiles_x <= 128 ? 128 : 256);
        return (size_t)rows * GSR_BAND_RANGES * 4 * padded;
    }
    // lean (inference frames, GsrSettings.forward_only on the default path): the arrays only a backward or a fallback
    // placement reads are not carved -- 62 % of the state at config 2 (157 of 253 MB; a closed loop keeps one state per
    // environment x camera).  A lean state is no input for gsr_backward / gsr_state_view.
    static GeomState carve(char *base, int32_t P, int tiles, size_t *bytes = nullptr, int tiles_x = 0, bool lean = false) {
        GeomState g;
        char *p = base;
        const size_t n = (size_t)(P > 0 ? P : 1), nf = lean ? 0 : n;
        g.hdr = take<GsrHeader>(p, 1);
        // GSR_BIN_SLOTS copies of each per-tile counter: workgroup b uses copy b % GSR_BIN_SLOTS, which divides the
        // same-address contention of the device-scope atomics (they resolve memory-side, ~100 ns apiece) by 16
        g.tile_accum = take<uint32_t>(p, (counting(tiles) ? (size_t)tiles * GSR_BIN_SLOTS : 0) + 1);
        g.tile_cursor = take<uint32_t>(p, (counting(tiles) ? (size_t)tiles * GSR_BIN_SLOTS : 0) + 1);
        g.splat = take<float4>(p, 3 * n);
        g.cov3D = take<float>(p, 6 * nf);
        g.clamped = take<uint32_t>(p, nf);
        g.grad_rec = take<GsrGradWord>(p, 12 * nf);
        g.tiles_touched = take<uint32_t>(p, nf);
        g.rects = take<uint2>(p, n);
        g.block_counts = take<uint32_t>(p, (size_t)prep_blocks(P) + 1);
        g.pair[0] = take<uint2>(p, n);
        g.pair[1] = take<uint2>(p, n);
        g.order = take<uint32_t>(p, n);
        g.sort_table = take<uint32_t>(p, lean ? 0 : (size_t)GSR_DEPTH_RADIX_BINS * sort_blocks(P));
        g.sort_totals = take<uint32_t>(p, GSR_DEPTH_RADIX_BINS);
        g.tile_bsum = take<uint32_t>(p, (size_t)sort_blocks(P) + 1);
        const size_t tt = counting(tiles) ? (size_t)tiles : 0;
        g.tile_table = take<uint32_t>(p, (lean ? 0 : tt * prep_blocks(P)) + 1);
        g.tile_totals = take<uint32_t>(p, tt + 1);
        g.vis_key = take<uint32_t>(p, 0);   // (no longer carved: preprocess leaves block-local records in pair[1] instead)
        g.block_cand = take<uint32_t>(p, (size_t)prep_blocks(P));
        g.ss_table = take<uint32_t>(p, (size_t)gsr_ss_nbc(P) * gsr_ss_bmax(P));
        g.ss_splitters = take<uint32_t>(p, (size_t)gsr_ss_bmax(P));
        g.ss_bucket_start = take<uint32_t>(p, (size_t)gsr_ss_bmax(P) + 1);
        g.ss_seg = take<uint32_t>(p, (size_t)gsr_ss_nbc(P) + 1);
        g.ss_dbg = take<uint64_t>(p, 64);
        g.rect_sorted = take<uint2>(p, n);
        g.band_table = take<uint32_t>(p, (size_t)tiles * GSR_BAND_RANGES + 1);
        g.tile_cum = take<uint32_t>(p, n);
        g.bucket_tiles = take<uint32_t>(p, (size_t)gsr_ss_bmax(P));
        g.wave_lo = take<uint32_t>(p, (size_t)GSR_BAND_RANGES * 4 + 1);
        g.wave_lo_base = take<uint32_t>(p, (size_t)GSR_BAND_RANGES * 4 + 1);
        g.ss_splitters_new = take<uint32_t>(p, (size_t)gsr_ss_bmax(P));
        g.ss_totals = take<uint32_t>(p, (size_t)gsr_ss_bmax(P));
        g.band_nseg = take<uint32_t>(p, (size_t)tiles * GSR_BAND_RANGES);  // (rows <= tiles: sized for the narrowest grid)
        g.ss_first = take<uint32_t>(p, (size_t)gsr_ss_nbc(P) + 1);
        g.block_recs = take<uint2>(p, n);
        g.pc_slots = take<float>(p, 2 * GSR_PC_SLOT);
        // LAST: the only array whose size depends on tiles_x, which the read-only carvers (gsr_backward,
        // gsr_state_view, gsr_debug_ss_stamps) do not pass -- nothing may follow it
        g.band_wtable = take<uint32_t>(p, band_wtable_words(tiles_x, tiles));
        if (bytes) *bytes = (size_t)(p - base);
        return g;
    }
    static size_t required(int32_t P, int tiles, int tiles_x = 0, bool lean = false) {
        size_t bytes = 0;
        carve(nullptr, P, tiles, &bytes, tiles_x, lean);
        return bytes;
    }
};

// ---- binning state (per rendered instance) -------------------------------------------------------------
struct BinningState {
    uint32_t *tile[2];     // [Rcap] tile id per instance, ping-pong
    uint32_t *gidx[2];     // [Rcap] Gaussian index per instance, ping-pong
    uint32_t *sort_table;  // [256 * nb]
    uint32_t *sort_totals; // [256]
    uint64_t *keys64;      // [Rcap] (depth bits << 32 | Gaussian index) per instance, grouped by tile, unsorted

    static int sort_blocks(int64_t rcap) { return gsr_div_up(rcap > 0 ? rcap : 1, GSR_SORT_CHUNK); }
    // gidx[0] -- the point list every path ends in -- comes first, so that a view of the state finds it whatever was
    // carved behind it.  lean: nothing else (the counting placements write the list in place; the keys, tile ids and
    // ping-pong sides only serve the radix / bin-then-sort fallbacks): 4 instead of 24 bytes per instance.
    static BinningState carve(char *base, int64_t rcap, size_t *bytes = nullptr, bool lean = false) {
        BinningState b;
        char *p = base;
        const size_t n = (size_t)(rcap > 0 ? rcap : 1), nf = lean ? 0 : n;
        b.gidx[0] = GeomState::take<uint32_t>(p, n);
        b.keys64 = GeomState::take<uint64_t>(p, nf);
        b.tile[0] = GeomState::take<uint32_t>(p, nf);
        b.tile[1] = GeomState::take<uint32_t>(p, nf);
        b.gidx[1] = GeomState::take<uint32_t>(p, nf);
        b.sort_table = GeomState::take<uint32_t>(p, lean ? 0 : (size_t)GSR_RADIX_BINS * sort_blocks(rcap));
        b.sort_totals = GeomState::take<uint32_t>(p, lean ? 0 : GSR_RADIX_BINS);
        if (bytes) *bytes = (size_t)(p - base);
        return b;
    }
    static size_t required(int64_t rcap, bool lean = false) {
        size_t bytes = 0;
        carve(nullptr, rcap, &bytes, lean);
        return bytes;
    }
    // number of tile-sort passes and therefore which ping-pong side holds the sorted result
    static int tile_bits(int num_tiles) {
        int bits = 1;
        while ((1 << bits) < num_tiles) bits++;
        return bits;
    }
    static int tile_passes(int num_tiles) { return (tile_bits(num_tiles) + GSR_RADIX_BITS - 1) / GSR_RADIX_BITS; }
};

// ---- image state (per pixel / per tile) ----------------------------------------------------------------
struct ImageState {
    uint2 *ranges;       // [tiles]
    float *final_T;      // [W*H]
    uint32_t *n_contrib; // [W*H]
    uint32_t *tile_order; // [tiles] tile ids, longest instance list first (compositing queue order)
    uint32_t *quad_work;  // [4*tiles] cost of each 8x8 quadrant in the compositor's LAST frame on this state (order key of
                          // the next frame; garbage on a fresh state -- any key gives a valid permutation)
    uint32_t *quad_work_b; // [4*tiles] cost of rows 4-7 of a quadrant that was split in the last frame
    uint32_t *split_flag; // [4*tiles] 1: this frame the quadrant is composited as two 8x4 halves
    uint32_t *split_list; // [4*tiles] quadrant ids (4 tile + quad) whose second half an extra wave takes
    uint32_t *split_count;// [1]
    uint32_t *quad_order; // [4*tiles] quadrant ids (4 tile + quad) sorted by descending cost in the last frame
    uint32_t *tile_dirty; // [tiles] == GsrHeader::td_token: a Gaussian whose records this frame's preprocess recomputed touches the
                          // tile now, or touched it in the previous frame (tile reuse: render.hip skips the others)
    static ImageState carve(char *base, int32_t W, int32_t H, size_t *bytes = nullptr) {
        ImageState s;
        char *p = base;
        const size_t tiles = (size_t)gsr_div_up(W, GSR_TILE) * gsr_div_up(H, GSR_TILE);
        s.ranges = GeomState::take<uint2>(p, tiles);
        s.tile_order = GeomState::take<uint32_t>(p, tiles);
        s.quad_work = GeomState::take<uint32_t>(p, 4 * tiles);
        s.quad_work_b = GeomState::take<uint32_t>(p, 4 * tiles);
        s.split_flag = GeomState::take<uint32_t>(p, 4 * tiles);
        s.split_list = GeomState::take<uint32_t>(p, 4 * tiles);
        s.split_count = GeomState::take<uint32_t>(p, 64);
        s.quad_order = GeomState::take<uint32_t>(p, 4 * tiles);
        s.final_T = GeomState::take<float>(p, (size_t)W * H);
        s.n_contrib = GeomState::take<uint32_t>(p, (size_t)W * H);
        s.tile_dirty = GeomState::take<uint32_t>(p, tiles);
        if (bytes) *bytes = (size_t)(p - base);
        return s;
    }
    static size_t required(int32_t W, int32_t H) {
        size_t bytes = 0;
        carve(nullptr, W, H, &bytes);
        return bytes;
    }
};

#if defined(__HIPCC__)
// exp in the canonical float32 order of oracle/gs_oracle.c gso_expf (raw-parameter path):
// 2^n * p(r) with n = rint(x log2e), r = x - n ln2 (Cody-Waite), p = the Cephes expf polynomial.  Bit-identical
// on host and device because every operation is a correctly rounded IEEE one (fma, rint, ldexp).
__device__ __forceinline__ float exp_canonical(float x) {
    if (x > 88.72283905206835f) return __builtin_inff();
    if (x < -103.972084045410f) return 0.0f;
    const float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float y = __builtin_fmaf(p, r * r, r) + 1.0f;
    return __builtin_ldexpf(y, (int)n);
}

// exp(power) of the compositors (forward and backward evaluate the same value).  The hardware transcendental works in
// base 2: t = power * log2(e) is rounded once (relative 2^-24 of |t|, i.e. up to ~3 ulp of the result at |t| ~ 8) before
// v_exp_f32 adds its own ~1 ulp.  -DGSR_EXP_ACCURATE=1 recovers the rounding of t -- e = fma(power, L, -t) + power * L_lo
// is the exact residual, exp2(t + e) = exp2(t) (1 + e ln 2) -- which leaves v_exp_f32's own error only.
#ifndef GSR_EXP_ACCURATE
#define GSR_EXP_ACCURATE 0
#endif
__device__ __forceinline__ float gsr_exp_power(float power) {
    constexpr float L = 1.4426950408889634f;
    const float t = power * L;
#if GSR_EXP_ACCURATE
    constexpr float L_lo = (float)(1.4426950408889634073599246810019 - (double)1.4426950408889634f);
    const float e = __builtin_fmaf(power, L_lo, __builtin_fmaf(power, L, -t));
    const float x = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(x * 0.6931471805599453f, e, x);
#else
    return __builtin_amdgcn_exp2f(t);
#endif
}

// Can instance (centre, conic|opacity) reach alpha >= 1/255 on ANY pixel of the 8x8 quadrant whose first pixel is
// (x0, y0)?  alpha >= 1/255 <=> power >= tau = -ln(255 * opacity), and power(d) = -(A dx^2 + C dy^2)/2 - B dx dy is
// concave in d = centre - pixel, so its maximum over the quadrant's (continuous) rectangle of offsets is at
// d = 0 when the centre is inside, otherwise on an edge facing the centre; on an edge it is a 1-D concave
// parabola whose maximiser is the clamped stationary point.  Evaluating both the dx = clamp(0) line and the
// dy = clamp(0) line covers every case (extra candidates are points of the rectangle, they cannot exceed the
// true maximum).  The answer must never be a false "no": the slack covers the float rounding of this bound and
// of the per-pixel evaluation (a few ulps of the largest term), and a conic that is not positive definite in
// float is never culled.  A culled instance is one every pixel of the quadrant would have skipped (alpha < 1/255),
// so the image, final_T and n_contrib are bit-identical with and without the test.
template <bool HAVE_TAU = false>
__device__ __forceinline__ bool quadrant_may_hit(float cx, float cy, const float4 q, float x0, float y0,
                                                 float yext = 7.0f, float tau_in = 0.0f) {
    const float A = q.x, B = q.y, C = q.z;
    // the same subtractions the corner pixels perform: every pixel's rounded offset lies in [dxl, dxh] x [dyl, dyh]
    const float dxh = cx - x0, dxl = cx - (x0 + 7.0f);
    const float dyh = cy - y0, dyl = cy - (y0 + yext);  // (yext = rows - 1: 7 for a quadrant, 3 for half of one)
    // opacity 0 -> +inf.  HAVE_TAU: computed once per Gaussian by preprocess (inference frames), a hair lower
    const float tau = HAVE_TAU ? tau_in : -0.6931471805599453f * __builtin_amdgcn_logf(255.0f * q.w);
    const float ex = fminf(fmaxf(0.0f, dxl), dxh);
    const float ey = fminf(fmaxf(0.0f, dyl), dyh);
    const float sy = fminf(fmaxf(-(B * ex) * __builtin_amdgcn_rcpf(C), dyl), dyh);
    const float sx = fminf(fmaxf(-(B * ey) * __builtin_amdgcn_rcpf(A), dxl), dxh);
    const float f1 = __builtin_fmaf(-B * ex, sy, -0.5f * __builtin_fmaf(C * sy, sy, (A * ex) * ex));
    const float f2 = __builtin_fmaf(-B * sx, ey, -0.5f * __builtin_fmaf(C * ey, ey, (A * sx) * sx));
    const float mx = fmaxf(fabsf(dxl), fabsf(dxh)), my = fmaxf(fabsf(dyl), fabsf(dyh));
    const float mag = __builtin_fmaf(A * mx, mx, __builtin_fmaf(C * my, my, 2.0f * fabsf(B) * mx * my));
    const float slack = __builtin_fmaf(4e-6f, mag, 1e-4f);
    const bool concave = A > 0.0f && C > 0.0f && A * C > B * B * 1.00001f;
    const bool miss = fmaxf(f1, f2) < tau - slack;  // false on NaN
    // opacity < 1/255: alpha = opacity * exp(power <= 0) <= opacity can never pass (a NaN opacity is not culled:
    // fminf(0.99, NaN) = 0.99 composites, as it does upstream)
    return !(q.w < 1.0f / 255.0f || (concave && miss));
}

// sigmoid of the raw-parameter path (forward and backward use the same value)
__device__ __forceinline__ float sigmoid_canonical(float x) { return 1.0f / (1.0f + exp_canonical(-x)); }
#endif

// ---- frames per launch (gsr_forward_batch) ------------------------------------------------------------------
// Every kernel of the default path takes its arguments as a table of up to GSR_MAX_BATCH per-frame argument blocks in
// the KERNARG segment and picks its frame's block by a grid index (blockIdx.y, or .z where .y is taken): B frames of
// one model are B x the workgroups of every launch instead of B x the launches.  The index is wave-uniform, so the
// block's fields arrive by scalar loads exactly as plain kernel arguments do (s_load from the kernarg base + a
// computed offset; no scratch -- checked in the ISA), and a single frame is the table with one entry.
#define GSR_MAX_BATCH GSR_MAX_FRAMES_PER_LAUNCH
template <typename A>
struct GsrBatch {
    A f[GSR_MAX_BATCH];
    // (the kernarg segment of a launch is 4 KiB: a table that outgrows it fails at run time only -- hipErrorInvalidValue
    //  on the first launch -- so the limit is checked where an argument block gains a field)
    static_assert(sizeof(A) * GSR_MAX_BATCH <= 4096 - 128, "per-frame argument blocks must fit the 4 KiB kernarg segment");
};

// One frame of a (batched) set of launches as api.hip resolved it: the caller's structs and the carved state.
struct GsrFrame {
    const GsrSettings *st;      // the frame's settings
    const GsrSettings *st_bin;  // the same with the SUPER-TILE grid as its image (inference frames), else == st
    const GsrInputs *in;
    const GsrOutputs *out;
    GeomState g;
    ImageState img;
    BinningState b;             // (valid from the placement on)
    uint32_t cap32;             // binning capacity the range kernel checks R against
    bool pc;                    // this frame's preprocess may keep last frame's blocks (preprocess.hip prep_block_cached)
    bool td;                    // ... and its compositor the tiles none of the recomputed Gaussians touches (render.hip)
};

// ---- error plumbing ------------------------------------------------------------------------------------
void gsr_set_error(const char *fmt, ...);
int gsr_check_launch(const char *what, bool debug, hipStream_t stream);

// ---- launchers implemented in the kernel files -----------------------------------------------------------
// (B frames of one model size and image size; every launcher below that takes `B, fr` spans them with one launch)
int gsr_launch_preprocess(int B, GsrFrame *fr, bool count_tiles, bool infer, hipStream_t stream);
// bin-then-sort path (default): unordered binning into tile segments, then a per-tile (depth, index) sort
int gsr_launch_bin_starts(const GsrSettings &st, const GeomState &g, const ImageState &img, uint32_t r_capacity,
                          bool debug, hipStream_t stream);
int gsr_launch_bin_scatter_and_sort(const GsrSettings &st, int32_t P, const GeomState &g, const BinningState &b,
                                    const ImageState &img, bool debug, hipStream_t stream);
int gsr_launch_compact_and_depth_sort(int32_t P, const GeomState &g, bool debug, hipStream_t stream);
// (super_shift: 1 = rect_sorted in 2 x 2 super-tile units, GsrSettings.forward_only)
// (order_early: a second workgroup of the prepare launch deals the num_quads quadrants of the frame's compositor by their
//  cost in the previous frame -> img.quad_order; super_shift: 1 = rect_sorted in super-tile units, GsrSettings.forward_only)
int gsr_launch_sample_depth_sort(int B, const GsrFrame *fr, bool order_early, int num_quads, int super_shift,
                                 int coop_blocks, bool debug, hipStream_t stream);
bool gsr_band_supported(int gx);
int gsr_launch_gather_rects(int32_t P, const GeomState &g, bool debug, hipStream_t stream);
// (merge_starts: the caller wants nothing of tile_starts_kernel but ranges / R / the capacity check; *starts_done = true
//  when the counting launches have produced them -- gsr_launch_tile_starts is then not needed)
int gsr_launch_band_count(int B, const GsrFrame *fr, bool balanced, bool merge_starts, bool *starts_done, bool debug,
                          hipStream_t stream);
int gsr_launch_band_place(int B, const GsrFrame *fr, bool debug, hipStream_t stream);
// placement by chunks of the depth order (chunkplace.hip)
bool gsr_chunk_supported(int gx, int gy);
int gsr_launch_chunk_count(const GsrSettings &st, int32_t P, const GeomState &g, bool debug, hipStream_t stream);
int gsr_launch_chunk_place(const GsrSettings &st, int32_t P, const GeomState &g, const BinningState &b,
                           const ImageState &img, bool debug, hipStream_t stream);
int gsr_launch_tile_starts(int B, const GsrFrame *fr, bool order_done, bool debug, hipStream_t stream);
int gsr_launch_tile_starts(const GsrSettings &st, const GeomState &g, const ImageState &img, uint32_t r_capacity,
                           bool order_done, bool debug, hipStream_t stream);  // (one frame: the fallback placements)
int gsr_launch_tile_offsets(int32_t P, const GeomState &g, uint32_t r_capacity, bool debug, hipStream_t stream);
int gsr_launch_emit_and_tile_sort(const GsrSettings &st, int32_t P, const GeomState &g, const BinningState &b,
                                  const ImageState &img, int64_t r_capacity, bool debug, hipStream_t stream);
bool gsr_render_wants_tile_order(const GsrSettings &st, int num_tiles);
const char *gsr_render_build_flags();  // what render.hip was compiled with (csrc/Makefile may fall back): in gsr_version()
int gsr_render_split_blocks(const GsrSettings &st, int num_tiles);
bool gsr_render_uses_quad_order(const GsrSettings &st, int num_tiles);
int gsr_render_cus_per_xcd();  // CUs of one XCD (the quadrant deal of gsr_quad_order_block)
// (B > 1: the default compositor only -- api.hip batches nothing else)
int gsr_render_coop_blocks(const GsrSettings &st, int num_tiles, int frames);
int gsr_launch_render(int B, const GsrFrame *fr, bool order_ready, bool split_ready, bool super_tiles, int coop_blocks,
                      hipStream_t stream);
int gsr_launch_tile_count(const GsrSettings &st, int32_t P, const GeomState &g, const ImageState &img,
                          uint32_t r_capacity, bool debug, hipStream_t stream);
int gsr_launch_tile_place(const GsrSettings &st, int32_t P, const GeomState &g, const BinningState &b,
                          const ImageState &img, bool debug, hipStream_t stream);
int gsr_launch_rowscan(uint32_t *table, const uint32_t *n_ptr, int nb_stride, int chunk, int rows, uint32_t *totals,
                       bool debug, hipStream_t stream);
int gsr_radix_passes(int bits, int bits_per_pass);
int gsr_radix_sort_u32(uint32_t *key[2], uint32_t *val[2], const uint32_t *n_ptr, int64_t n_max, int bits,
                       int bits_per_pass, int start_side, uint32_t *table, uint32_t *totals, bool debug,
                       hipStream_t stream);

#ifdef __HIPCC__
// ---- wave64 / block primitives ---------------------------------------------------------------------------
__device__ __forceinline__ int gsr_lane() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ int gsr_wave() { return (int)(threadIdx.x >> 6); }
__device__ __forceinline__ uint64_t gsr_lanemask_lt() { return (1ull << gsr_lane()) - 1ull; }

// Inclusive running sum over the 64 lanes of the wave, in DPP moves: four shifted adds inside every row of 16 lanes
// (row_shr 1 / 2 / 4 / 8, out-of-row sources read as 0), then lane 15 of rows 0 and 2 into rows 1 and 3 (row_bcast:15),
// then lane 31 into the upper half (row_bcast:31): six VALU instructions.  (Rounds 1-5: six __shfl_up = ds_bpermute_b32
// round trips through the LDS crossbar, ~100 cycles each in a dependent chain -- every block scan of the frame's
// latency-bound kernels paid them.)
#ifndef GSR_DPP_SCAN
#define GSR_DPP_SCAN 1
#endif
__device__ __forceinline__ uint32_t gsr_wave_incl_scan(uint32_t v) {
#if GSR_DPP_SCAN
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return (uint32_t)x;
#else
    const int lane = gsr_lane();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
#endif
}

// Inclusive scan over the 256 threads of a block.  s_w must hold 4 uint32 in LDS.  Two barriers.
__device__ __forceinline__ uint32_t gsr_block_incl_scan(uint32_t v, uint32_t *s_w, uint32_t &total) {
    const uint32_t incl = gsr_wave_incl_scan(v);
    const int lane = gsr_lane(), wave = gsr_wave();
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    const uint32_t w0 = s_w[0], w1 = s_w[1], w2 = s_w[2], w3 = s_w[3];
    uint32_t add = 0;
    if (wave > 0) add += w0;
    if (wave > 1) add += w1;
    if (wave > 2) add += w2;
    total = w0 + w1 + w2 + w3;
    __syncthreads();
    return incl + add;
}

// One workgroup (256 threads): which compositing workgroup takes which of the image's Q <= 32 x 256 quadrants.
// quad_order[4 b + j], j = 0..3, are the quadrants of workgroup b (render.hip: one quadrant per wave, the four waves of
// a workgroup on the four SIMDs of one CU).  Three things are balanced at once, from the quadrants' costs in the
// previous frame on this state:
//   * a tile's four quadrants read the same instance list, so they stay on ONE XCD (one L2): tile t belongs to XCD
//     t mod 8, and workgroup b runs on XCD b mod 8 (round-robin dispatch) -- dealing the quadrants over the whole chip
//     made every L2 fetch the lists for itself (HBM-side traffic 79 -> 153 MB per frame);
//   * inside an XCD the quadrants are sorted by cost (256-bucket counting sort) and a workgroup takes four consecutive
//     ones, i.e. four of (nearly) equal cost: the SIMDs of a CU carry the same load;
//   * workgroups k, k + cus, k + 2 cus ... of an XCD share a CU (observed placement), so the sorted groups of four go
//     to the CUs in a snake: every CU gets one group of each cost class, the costliest of one with the cheapest of the
//     next.
// Depends on nothing of the current frame, so it runs wherever a spare workgroup costs nothing.  Any permutation gives
// the same image: a fresh state (garbage costs) only balances badly.
#define GSR_XCDS 8
__device__ __forceinline__ void gsr_quad_order_block(const uint32_t *__restrict__ quad_work, int Q,
                                                     uint32_t *__restrict__ quad_order, uint32_t *s_w /*[4]*/,
                                                     int cus_per_xcd) {
    __shared__ uint32_t s_qb[GSR_XCDS * 256];
    __shared__ uint32_t s_xbase[GSR_XCDS];
    const int T = Q >> 2;
    uint32_t c[32], qmx = 0;
#pragma unroll
    for (int k = 0; k < 32; k++) {
        const int q = (int)threadIdx.x + k * GSR_BLOCK;
        c[k] = q < Q ? min(quad_work[q], (1u << 24) - 1u) : 0u;
        qmx = max(qmx, c[k]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) qmx = max(qmx, (uint32_t)__shfl_xor((int)qmx, o, 64));
    if ((threadIdx.x & 63u) == 0u) s_w[threadIdx.x >> 6] = qmx;
#pragma unroll
    for (int k = 0; k < GSR_XCDS; k++) s_qb[k * GSR_BLOCK + (int)threadIdx.x] = 0u;
    __syncthreads();
    qmx = max(max(s_w[0], s_w[1]), max(s_w[2], s_w[3]));
    // bucket = 255 - floor(cost * 256 / (max + 1)): cost < 2^24, so the product fits 32 bits after the shift
    const int sh = qmx >= (1u << 16) ? 8 : 0;  // (keeps cost * 256 below 2^32 and the divisor non-zero)
    const uint32_t div = (qmx >> sh) + 1u;
    const float inv = 256.0f / (float)div;
#pragma unroll
    for (int k = 0; k < 32; k++) {
        const int q = (int)threadIdx.x + k * GSR_BLOCK;
        const uint32_t xcd = (uint32_t)(q >> 2) % GSR_XCDS;
        c[k] = xcd * 256u + 255u - min(255u, (uint32_t)((float)(c[k] >> sh) * inv));  // bin = (XCD, cost class)
        if (q < Q) atomicAdd(&s_qb[c[k]], 1u);
    }
    __syncthreads();
    {  // exclusive running sum over the 2048 bins: thread t owns bins 8 t .. 8 t + 7 (XCD t / 32)
        uint32_t v[GSR_XCDS], sum = 0u;
#pragma unroll
        for (int k = 0; k < GSR_XCDS; k++) {
            v[k] = s_qb[GSR_XCDS * (int)threadIdx.x + k];
            sum += v[k];
        }
        uint32_t tot;
        uint32_t run = gsr_block_incl_scan(sum, s_w, tot) - sum;
        if ((threadIdx.x & 31u) == 0u) s_xbase[threadIdx.x >> 5] = run;  // first slot of the XCD's list
#pragma unroll
        for (int k = 0; k < GSR_XCDS; k++) {
            s_qb[GSR_XCDS * (int)threadIdx.x + k] = run;
            run += v[k];
        }
    }
    __syncthreads();
    const uint32_t cus = (uint32_t)max(cus_per_xcd, 1);
#pragma unroll
    for (int k = 0; k < 32; k++) {
        const int q = (int)threadIdx.x + k * GSR_BLOCK;
        if (q < Q) {
            const uint32_t xcd = c[k] >> 8;
            const uint32_t p = atomicAdd(&s_qb[c[k]], 1u) - s_xbase[xcd];        // position in the XCD's sorted list
            const uint32_t nwg = ((uint32_t)T - 1u - xcd) / GSR_XCDS + 1u;         // workgroups (= tiles) of this XCD
            const uint32_t slot = p >> 2, round = slot / cus, idx = slot - round * cus;
            const uint32_t size = min(cus, nwg - round * cus);
            const uint32_t cu = (round & 1u) ? size - 1u - idx : idx;
            const uint32_t b = GSR_XCDS * (round * cus + cu) + xcd;
            quad_order[4u * b + (p & 3u)] = (uint32_t)q;
        }
    }
    __syncthreads();
}
// Visits every tile of this lane's rect (t = tiles touched, rc = packed rect; t == 0 for lanes without a
// Gaussian).  Lists of up to 32 tiles are walked by the owning lane; longer ones are spread over the whole wave,
// 64 tiles per step, so that one large splat does not serialise its wave.  f(tile, lo, hi) runs once per
// (Gaussian, tile) with the OWNER's 64-bit payload.  All 64 lanes must call this together.
template <typename F>
__device__ __forceinline__ void gsr_for_each_tile(uint32_t t, uint2 rc, int gx, uint32_t lo, uint32_t hi, F f) {
    const int lane = gsr_lane();
    const uint32_t minx = rc.x & 0xffffu, miny = rc.x >> 16, maxx = rc.y & 0xffffu, width = maxx - minx;
    if (t > 0u && t <= 32u) {
        uint32_t x = minx, row = miny * (uint32_t)gx;
        for (uint32_t j = 0; j < t; j++) {
            f(row + x, lo, hi);
            if (++x == maxx) { x = minx; row += (uint32_t)gx; }
        }
    }
    uint64_t todo = __builtin_amdgcn_ballot_w64(t > 32u);
    // j / width without an integer division: (j + 0.5) * (1 / width) truncates to the exact quotient (j < 2^21)
    const float inv_w = 1.0f / (float)(width > 0u ? width : 1u);
    while (todo) {
        const int src = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1ull;
        const uint32_t bt = (uint32_t)__builtin_amdgcn_readlane((int)t, src);
        const uint32_t bminx = (uint32_t)__builtin_amdgcn_readlane((int)minx, src);
        const uint32_t bminy = (uint32_t)__builtin_amdgcn_readlane((int)miny, src);
        const uint32_t bw = (uint32_t)__builtin_amdgcn_readlane((int)width, src);
        const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)lo, src);
        const uint32_t bhi = (uint32_t)__builtin_amdgcn_readlane((int)hi, src);
        const float binv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inv_w), src));
        for (uint32_t j = (uint32_t)lane; j < bt; j += 64u) {
            const uint32_t yy = (uint32_t)(((float)j + 0.5f) * binv), xx = j - yy * bw;
            f((bminy + yy) * (uint32_t)gx + (bminx + xx), blo, bhi);
        }
    }
}

// Longest-first tile order for the compositing queue: a 64-bucket counting sort of the tile list lengths, run by
// ONE workgroup (callers: tile_starts_kernel on the counting path, tile_order_kernel on the radix fallback).
// s_bins: 64 uint32 in LDS, s_red: 4 uint32 in LDS.  The ranges must be visible to the whole workgroup.
// With `work` (4 words per tile: the cost of its quadrants in the previous frame) the key is that cost instead of the
// list length.
__device__ __forceinline__ uint32_t gsr_tile_order_key(const uint2 *ranges, const uint32_t *work, int t) {
    if (work != nullptr) {
        const uint4 w = *reinterpret_cast<const uint4 *>(work + 4 * t);
        return (w.x >> 2) + (w.y >> 2) + (w.z >> 2) + (w.w >> 2);  // (no wrap-around on a fresh state's garbage)
    }
    return ranges[t].y - ranges[t].x;
}
__device__ __forceinline__ void gsr_tile_order_block(const uint2 *ranges, int num_tiles, uint32_t *order,
                                                     uint32_t *s_bins, uint32_t *s_red, const uint32_t *work = nullptr) {
    uint32_t mx = 0;
    for (int t = (int)threadIdx.x; t < num_tiles; t += GSR_BLOCK) mx = max(mx, gsr_tile_order_key(ranges, work, t));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
    if (gsr_lane() == 0) s_red[gsr_wave()] = mx;
    if (threadIdx.x < 64) s_bins[threadIdx.x] = 0u;
    __syncthreads();
    mx = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]));
    const float scale = mx > 0u ? 63.999f / (float)mx : 0.f;
    for (int t = (int)threadIdx.x; t < num_tiles; t += GSR_BLOCK)  // bucket 0 = longest lists
        atomicAdd(&s_bins[63 - (int)((float)gsr_tile_order_key(ranges, work, t) * scale)], 1u);
    __syncthreads();
    if (threadIdx.x < 64) {  // exclusive scan of the 64 buckets by the first wave
        const uint32_t c = s_bins[threadIdx.x];
        uint32_t incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, o, 64);
            if ((int)threadIdx.x >= o) incl += v;
        }
        s_bins[threadIdx.x] = incl - c;
    }
    __syncthreads();
    for (int t = (int)threadIdx.x; t < num_tiles; t += GSR_BLOCK) {
        const uint32_t pos = atomicAdd(&s_bins[63 - (int)((float)gsr_tile_order_key(ranges, work, t) * scale)], 1u);
        order[pos] = (uint32_t)t;
    }
}
// The same for up to NK x GSR_BLOCK tiles with the keys already in registers (key[i] belongs to tile
// threadIdx.x + i * GSR_BLOCK): the caller loads them at the top of its kernel, so no global read sits between the passes.
template <int NK>
__device__ __forceinline__ void gsr_tile_order_block_keys(const uint32_t (&key)[NK], int num_tiles, uint32_t *order,
                                                          uint32_t *s_bins, uint32_t *s_red) {
    uint32_t mx = 0;
#pragma unroll
    for (int i = 0; i < NK; i++)
        if ((int)threadIdx.x + i * GSR_BLOCK < num_tiles) mx = max(mx, key[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
    if (gsr_lane() == 0) s_red[gsr_wave()] = mx;
    if (threadIdx.x < 64) s_bins[threadIdx.x] = 0u;
    __syncthreads();
    mx = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]));
    const float scale = mx > 0u ? 63.999f / (float)mx : 0.f;
#pragma unroll
    for (int i = 0; i < NK; i++)
        if ((int)threadIdx.x + i * GSR_BLOCK < num_tiles) atomicAdd(&s_bins[63 - (int)((float)key[i] * scale)], 1u);
    __syncthreads();
    if (threadIdx.x < 64) {
        const uint32_t c = s_bins[threadIdx.x];
        uint32_t incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, o, 64);
            if ((int)threadIdx.x >= o) incl += v;
        }
        s_bins[threadIdx.x] = incl - c;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NK; i++) {
        const int t = (int)threadIdx.x + i * GSR_BLOCK;
        if (t < num_tiles) order[atomicAdd(&s_bins[63 - (int)((float)key[i] * scale)], 1u)] = (uint32_t)t;
    }
}
// Ascending-only bitonic network over `n` keys padded (virtually) to N = 2^k with +inf: every comparator puts the
// minimum at the lower index, so padding slots never move and need not exist.  256 threads, barrier per stage.
// (pay: optional array of the same length whose elements travel with their keys)
template <typename Ptr, typename Pay = uint32_t *>
__device__ __forceinline__ void bitonic_sort_block(Ptr a, int n, int N, Pay pay = nullptr) {
    // all sizes are powers of two: index arithmetic with shifts and masks only
    for (int lk = 1; (1 << lk) <= N; lk++) {
        const int k = 1 << lk, hk = k >> 1;
        // flip step: i <-> mirror position inside each block of k
        for (int p = (int)threadIdx.x; p < (N >> 1); p += GSR_BLOCK) {
            const int off = p & (hk - 1);
            const int blk0 = (p >> (lk - 1)) << lk;
            const int i = blk0 + off, j = blk0 + k - 1 - off;
            if (j < n) {
                const uint64_t x = a[i], y = a[j];
                if (x > y) {
                    a[i] = y; a[j] = x;
                    if (pay != nullptr) { const auto t = pay[i]; pay[i] = pay[j]; pay[j] = t; }
                }
            }
        }
        __syncthreads();
        for (int ld = lk - 2; ld >= 0; ld--) {
            const int d = 1 << ld;
            for (int p = (int)threadIdx.x; p < (N >> 1); p += GSR_BLOCK) {
                const int i = ((p >> ld) << (ld + 1)) | (p & (d - 1)), j = i + d;
                if (j < n) {
                    const uint64_t x = a[i], y = a[j];
                    if (x > y) {
                        a[i] = y; a[j] = x;
                        if (pay != nullptr) { const auto t = pay[i]; pay[i] = pay[j]; pay[j] = t; }
                    }
                }
            }
            __syncthreads();
        }
    }
}
#endif  // __HIPCC__
