// sort.hip -- index-ordered compaction, device scans and the stable LSD radix sort used by the binning stage
// (upstream rasterizer_impl.cu: cub::DeviceScan::InclusiveSum + cub::DeviceRadixSort::SortPairs, SURVEY.md 8a
// rows A5/A6).  Hand-written for wave64: digit matching by ballot, ranks by popcount of the lanes below.
//
// Every kernel takes its element count from DEVICE memory (the frame header) and is launched over the
// capacity, so the frame never waits for the host: blocks past the live count exit at once.
#include "gsr_internal.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// Exclusive scan (in place) of a small array by ONE workgroup; writes the grand total to *total_out.
// mode 0: plain.  mode 1: the total is num_rendered -- record R_raw, clamp against the binning capacity.
// `n_items_ptr`/`chunk` (optional): the live length is ceil(*n_items_ptr / chunk) instead of n_static.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GSR_BLOCK) void scan_small_kernel(uint32_t *data, int n_static,
                                                               const uint32_t *n_items_ptr, int chunk,
                                                               GsrHeader *hdr, int mode, uint32_t r_capacity) {
    __shared__ uint32_t s_w[4];
    int n = n_static;
    if (n_items_ptr) n = (int)((*n_items_ptr + (uint32_t)chunk - 1u) / (uint32_t)chunk);
    // each thread owns a contiguous slice: one pass to sum it, ONE block scan of the 256 slice sums, one pass to
    // write the exclusive offsets (2 barriers in total instead of 2 per 256 elements)
    const int per = (n + GSR_BLOCK - 1) / GSR_BLOCK;
    const int lo = min(n, (int)threadIdx.x * per), hi = min(n, lo + per);
    uint32_t mine = 0;
    for (int i = lo; i < hi; i++) mine += data[i];
    uint32_t carry;
    uint32_t run = gsr_block_incl_scan(mine, s_w, carry) - mine;
    for (int i = lo; i < hi; i++) {
        const uint32_t v = data[i];
        data[i] = run;
        run += v;
    }
    if (threadIdx.x == 0) {
        data[n] = carry;  // arrays carry one spare slot: exclusive offsets have n+1 entries
        if (mode == 0) {
            // first kernel of the frame that touches the header: reset it here (saves a memset launch)
            hdr->V = carry;
            hdr->R = 0u;
            hdr->overflow = 0u;
            hdr->r_capacity = 0u;
            hdr->R_raw = 0u;
            hdr->tile_queue = 0u;
        } else {
            hdr->R_raw = carry;
            hdr->r_capacity = r_capacity;
            if (carry > r_capacity) {
                hdr->overflow = 1u;
                hdr->R = 0u;
            } else {
                hdr->overflow = 0u;
                hdr->R = carry;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Compaction in INDEX order: visible Gaussian i of preprocess block b lands at block_offsets[b] + rank.
// Index order matters: the stable depth sort that follows breaks depth ties by ascending Gaussian index,
// which is what the reference's stable (tile|depth) key sort over index-ordered emission produces.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GSR_BLOCK) void compact_kernel(int P, const uint32_t *__restrict__ tiles_touched,
                                                            const float4 *__restrict__ splat,
                                                            const uint32_t *__restrict__ block_offsets,
                                                            uint32_t *__restrict__ keys, uint32_t *__restrict__ idx) {
    __shared__ uint32_t s_w[4];
    const int i = blockIdx.x * GSR_BLOCK + (int)threadIdx.x;
    const bool vis = i < P && tiles_touched[i] != 0u;
    uint32_t total;
    const uint32_t incl = gsr_block_incl_scan(vis ? 1u : 0u, s_w, total);
    if (vis) {
        const uint32_t pos = block_offsets[blockIdx.x] + incl - 1u;
        keys[pos] = __float_as_uint(splat[3 * (size_t)i].z);  // depth > 0: float bits are order-preserving
        idx[pos] = (uint32_t)i;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Radix sort, one pass = histogram -> per-digit row scan -> stable scatter.
// table[d * nb_stride + b] = number of keys with digit d in block b (block = GSR_SORT_CHUNK consecutive keys).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GSR_BLOCK) void radix_hist_kernel(const uint32_t *__restrict__ keys,
                                                               const uint32_t *__restrict__ n_ptr,
                                                               uint32_t *__restrict__ table, int nb_stride, int shift,
                                                               uint32_t mask) {
    const uint32_t n = *n_ptr;
    const uint32_t base = blockIdx.x * (uint32_t)GSR_SORT_CHUNK;
    if (base >= n) return;
    __shared__ uint32_t s_h[GSR_RADIX_BINS];
    s_h[threadIdx.x] = 0u;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < GSR_SORT_ITEMS; r++) {
        const uint32_t i = base + (uint32_t)r * GSR_BLOCK + threadIdx.x;
        if (i < n) atomicAdd(&s_h[(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    table[(size_t)threadIdx.x * nb_stride + blockIdx.x] = s_h[threadIdx.x];
}

// one workgroup per digit: exclusive scan of its row over the live blocks, row total -> totals[d]
__global__ __launch_bounds__(GSR_BLOCK) void radix_rowscan_kernel(uint32_t *__restrict__ table,
                                                                  const uint32_t *__restrict__ n_ptr, int nb_stride,
                                                                  uint32_t *__restrict__ totals, int chunk) {
    __shared__ uint32_t s_w[4];
    const uint32_t n = *n_ptr;
    const int nb = (int)((n + (uint32_t)chunk - 1u) / (uint32_t)chunk);
    uint32_t *row = table + (size_t)blockIdx.x * nb_stride;
    uint32_t carry = 0;
    for (int base = 0; base < nb; base += GSR_BLOCK) {
        const int i = base + (int)threadIdx.x;
        const uint32_t v = i < nb ? row[i] : 0u;
        uint32_t total;
        const uint32_t incl = gsr_block_incl_scan(v, s_w, total);
        if (i < nb) row[i] = carry + incl - v;
        carry += total;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

__global__ __launch_bounds__(GSR_BLOCK) void radix_scatter_kernel(
    const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, uint32_t *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out, const uint32_t *__restrict__ n_ptr, const uint32_t *__restrict__ table,
    const uint32_t *__restrict__ totals, int nb_stride, int shift, uint32_t mask, int nbits) {
    const uint32_t n = *n_ptr;
    const uint32_t base = blockIdx.x * (uint32_t)GSR_SORT_CHUNK;
    if (base >= n) return;
    __shared__ uint32_t s_base[GSR_RADIX_BINS];
    __shared__ uint32_t s_wcnt[4][GSR_RADIX_BINS];
    __shared__ uint32_t s_w[4];
    const int tid = (int)threadIdx.x, wave = gsr_wave();
    const uint64_t lt = gsr_lanemask_lt();
    {
        // global start of digit `tid` for this block = (keys with a smaller digit) + (same digit, earlier blocks)
        const uint32_t tot = totals[tid];
        uint32_t all;
        const uint32_t incl = gsr_block_incl_scan(tot, s_w, all);
        s_base[tid] = incl - tot + table[(size_t)tid * nb_stride + blockIdx.x];
        s_wcnt[0][tid] = 0u; s_wcnt[1][tid] = 0u; s_wcnt[2][tid] = 0u; s_wcnt[3][tid] = 0u;
    }
    __syncthreads();
    for (int r = 0; r < GSR_SORT_ITEMS; r++) {
        const uint32_t i = base + (uint32_t)r * GSR_BLOCK + (uint32_t)tid;
        const bool valid = i < n;
        const uint32_t key = valid ? keys_in[i] : 0u;
        const uint32_t val = valid ? vals_in[i] : 0u;
        const uint32_t d = (key >> shift) & mask;
        // lanes of this wave holding the same digit (wave64 "match any" from nbits ballots)
        uint64_t same = __ballot(valid);
        for (int b = 0; b < nbits; b++) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            same &= bit ? bal : ~bal;
        }
        const uint32_t rank = (uint32_t)__popcll(same & lt);
        if (valid && rank == 0u) s_wcnt[wave][d] = (uint32_t)__popcll(same);
        __syncthreads();
        if (valid) {
            uint32_t pos = s_base[d] + rank;
            if (wave > 0) pos += s_wcnt[0][d];
            if (wave > 1) pos += s_wcnt[1][d];
            if (wave > 2) pos += s_wcnt[2][d];
            keys_out[pos] = key;
            vals_out[pos] = val;
        }
        __syncthreads();
        s_base[tid] += s_wcnt[0][tid] + s_wcnt[1][tid] + s_wcnt[2][tid] + s_wcnt[3][tid];
        s_wcnt[0][tid] = 0u; s_wcnt[1][tid] = 0u; s_wcnt[2][tid] = 0u; s_wcnt[3][tid] = 0u;
        __syncthreads();
    }
}

// tiles touched by each chunk of GSR_SORT_CHUNK depth-ordered Gaussians
__global__ __launch_bounds__(GSR_BLOCK) void tile_blocksum_kernel(const uint32_t *__restrict__ order,
                                                                  const uint32_t *__restrict__ tiles_touched,
                                                                  const GsrHeader *__restrict__ hdr,
                                                                  uint32_t *__restrict__ bsum) {
    __shared__ uint32_t s_w[4];
    const uint32_t V = hdr->V;
    const uint32_t base = blockIdx.x * (uint32_t)GSR_SORT_CHUNK;
    if (base >= V) return;
    uint32_t acc = 0;
#pragma unroll
    for (int r = 0; r < GSR_SORT_ITEMS; r++) {
        const uint32_t i = base + (uint32_t)r * GSR_BLOCK + threadIdx.x;
        if (i < V) acc += tiles_touched[order[i]];
    }
    uint32_t total;
    gsr_block_incl_scan(acc, s_w, total);
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

}  // namespace

// Exclusive scan of each of `rows` table rows over its live entries (ceil(*n_ptr / chunk)); row totals -> totals.
int gsr_launch_rowscan(uint32_t *table, const uint32_t *n_ptr, int nb_stride, int chunk, int rows, uint32_t *totals,
                       bool debug, hipStream_t stream) {
    hipLaunchKernelGGL(radix_rowscan_kernel, dim3(rows), dim3(GSR_BLOCK), 0, stream, table, n_ptr, nb_stride, totals,
                       chunk);
    return gsr_check_launch("rowscan", debug, stream);
}

// Stable LSD radix sort of (key, val) pairs on the low `bits` bits.  Result lands in key[passes & 1].
int gsr_radix_sort_u32(uint32_t *key[2], uint32_t *val[2], const uint32_t *n_ptr, int64_t n_max, int bits,
                       uint32_t *table, uint32_t *totals, bool debug, hipStream_t stream) {
    const int nb = gsr_div_up(n_max > 0 ? n_max : 1, GSR_SORT_CHUNK);
    int src = 0;
    for (int shift = 0; shift < bits; shift += GSR_RADIX_BITS) {
        const int nbits = (bits - shift) < GSR_RADIX_BITS ? (bits - shift) : GSR_RADIX_BITS;
        const uint32_t mask = (1u << nbits) - 1u;
        hipLaunchKernelGGL(radix_hist_kernel, dim3(nb), dim3(GSR_BLOCK), 0, stream, key[src], n_ptr, table, nb, shift,
                           mask);
        if (int e = gsr_check_launch("radix_hist", debug, stream)) return e;
        hipLaunchKernelGGL(radix_rowscan_kernel, dim3(GSR_RADIX_BINS), dim3(GSR_BLOCK), 0, stream, table, n_ptr, nb,
                           totals, GSR_SORT_CHUNK);
        if (int e = gsr_check_launch("radix_rowscan", debug, stream)) return e;
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(nb), dim3(GSR_BLOCK), 0, stream, key[src], val[src],
                           key[src ^ 1], val[src ^ 1], n_ptr, table, totals, nb, shift, mask, nbits);
        if (int e = gsr_check_launch("radix_scatter", debug, stream)) return e;
        src ^= 1;
    }
    return GSR_OK;
}

// block_counts -> exclusive offsets + V; index-ordered compaction; depth sort (32-bit float keys).
// Leaves the depth order in g.idx[0] (4 passes, even).
int gsr_launch_compact_and_depth_sort(int32_t P, const GeomState &g, bool debug, hipStream_t stream) {
    const int nb1 = GeomState::prep_blocks(P);
    hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(GSR_BLOCK), 0, stream, g.block_counts, nb1,
                       (const uint32_t *)nullptr, 1, g.hdr, 0, 0u);
    if (int e = gsr_check_launch("scan_block_counts", debug, stream)) return e;
    hipLaunchKernelGGL(compact_kernel, dim3(nb1), dim3(GSR_BLOCK), 0, stream, P, g.tiles_touched, g.splat,
                       g.block_counts, g.key[0], g.idx[0]);
    if (int e = gsr_check_launch("compact", debug, stream)) return e;
    uint32_t *key[2] = {g.key[0], g.key[1]};
    uint32_t *val[2] = {g.idx[0], g.idx[1]};
    return gsr_radix_sort_u32(key, val, &g.hdr->V, P, 32, g.sort_table, g.sort_totals, debug, stream);
}

// per-chunk tile counts in depth order -> exclusive chunk offsets, R (clamped against the capacity)
int gsr_launch_tile_offsets(int32_t P, const GeomState &g, uint32_t r_capacity, bool debug, hipStream_t stream) {
    const int nb = GeomState::sort_blocks(P);
    hipLaunchKernelGGL(tile_blocksum_kernel, dim3(nb), dim3(GSR_BLOCK), 0, stream, g.idx[0], g.tiles_touched, g.hdr,
                       g.tile_bsum);
    if (int e = gsr_check_launch("tile_blocksum", debug, stream)) return e;
    hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(GSR_BLOCK), 0, stream, g.tile_bsum, 0, &g.hdr->V,
                       GSR_SORT_CHUNK, g.hdr, 1, r_capacity);
    return gsr_check_launch("scan_tile_bsum", debug, stream);
}
