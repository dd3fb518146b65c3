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

}  // namespace

int gsr_launch_emit_and_tile_sort(const GsrSettings &st, int32_t P, const GeomState &g, const BinningState &b,
                                  const ImageState &img, int64_t r_capacity, bool debug, hipStream_t stream) {
    const int gx = gsr_div_up(st.image_width, GSR_TILE), gy = gsr_div_up(st.image_height, GSR_TILE);
    const int tiles = gx * gy;
    hipLaunchKernelGGL(zero_ranges_kernel, dim3(gsr_div_up(tiles, GSR_BLOCK)), dim3(GSR_BLOCK), 0, stream,
                       img.ranges, tiles);
    if (int e = gsr_check_launch("zero_ranges", debug, stream)) return e;
    hipLaunchKernelGGL(emit_kernel, dim3(GeomState::sort_blocks(P)), dim3(GSR_BLOCK), 0, stream, g.idx[0],
                       g.tiles_touched, g.rects, g.tile_bsum, g.hdr, gx, b.tile[0], b.gidx[0]);
    if (int e = gsr_check_launch("emit", debug, stream)) return e;
    uint32_t *key[2] = {b.tile[0], b.tile[1]};
    uint32_t *val[2] = {b.gidx[0], b.gidx[1]};
    const int bits = BinningState::tile_bits(tiles);
    if (int e = gsr_radix_sort_u32(key, val, &g.hdr->R, r_capacity, bits, b.sort_table, b.sort_totals, debug, stream))
        return e;
    const int side = BinningState::tile_passes(tiles) & 1;
    hipLaunchKernelGGL(tile_ranges_kernel, dim3(gsr_div_up(r_capacity > 0 ? r_capacity : 1, GSR_BLOCK)),
                       dim3(GSR_BLOCK), 0, stream, b.tile[side], g.hdr, img.ranges);
    return gsr_check_launch("tile_ranges", debug, stream);
}
