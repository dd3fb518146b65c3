// sort.hip -- index-ordered compaction, device scans and the stable LSD radix sort used by the binning stage
// (upstream rasterizer_impl.cu: cub::DeviceScan::InclusiveSum + cub::DeviceRadixSort::SortPairs, SURVEY.md 8a
// rows A5/A6).  Hand-written for wave64: digit matching by ballot, ranks by popcount of the lanes below.
//
// Every kernel takes its element count from DEVICE memory (the frame header) and is launched over the
// capacity, so the frame never waits for the host: blocks past the live count exit at once.
#include "gsr_internal.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// Exclusive scan (in place) of a small array by ONE workgroup; writes the grand total to data[n].
// mode 0: plain scan.  mode 1: the total is num_rendered -- record R_raw, clamp against the binning capacity.
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
        if (mode == 1) {
            hdr->R_raw = carry;
            hdr->r_capacity = r_capacity;
            gsr_set_overflow(hdr, carry > r_capacity);
            hdr->R = carry > r_capacity ? 0u : carry;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Compaction in INDEX order: visible Gaussian i of preprocess block b lands at block_offsets[b] + rank.
// Index order matters: the stable depth sort that follows breaks depth ties by ascending Gaussian index,
// which is what the reference's stable (tile|depth) key sort over index-ordered emission produces.
//
// The block's offset is the sum of the visible counts preprocess left for the blocks before it.  Every block adds
// those up itself (<= nb/256 coalesced loads per thread, L2-resident) instead of waiting for a single-workgroup
// scan kernel in between (14 us of pure latency at 5738 blocks).  The last block knows the grand total: it writes
// V and resets the rest of the frame header (first kernel of the frame that touches it).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GSR_BLOCK) void compact_kernel(int P, const uint32_t *__restrict__ tiles_touched,
                                                            const float4 *__restrict__ splat,
                                                            const uint32_t *__restrict__ block_counts,
                                                            uint2 *__restrict__ pairs,
                                                            GsrHeader *__restrict__ hdr) {
    __shared__ uint32_t s_w[4];
    uint32_t part = 0;
    for (int j = (int)threadIdx.x; j < (int)blockIdx.x; j += GSR_BLOCK) part += block_counts[j];
    uint32_t offset;
    (void)gsr_block_incl_scan(part, s_w, offset);
    const int i = blockIdx.x * GSR_BLOCK + (int)threadIdx.x;
    const bool vis = i < P && tiles_touched[i] != 0u;
    uint32_t total;
    const uint32_t incl = gsr_block_incl_scan(vis ? 1u : 0u, s_w, total);
    if (vis) {
        const uint32_t pos = offset + incl - 1u;
        // depth > 0: float bits are order-preserving
        pairs[pos] = make_uint2(__float_as_uint(splat[3 * (size_t)i].z), (uint32_t)i);
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        hdr->V = offset + total;
        hdr->R = 0u;
        hdr->overflow = 0u;
        hdr->coop_timeout_now = 0u;
        hdr->r_capacity = 0u;
        hdr->R_raw = 0u;
        hdr->tile_queue = 0u;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Radix sort, one pass = histogram -> per-digit row scan -> stable scatter; BITS = digit width (8 or 11).
// A workgroup owns GSR_SORT_CHUNK consecutive keys; wave w owns the contiguous quarter [w*512, w*512+512).
// table[d * nb_stride + b] = number of keys with digit d in workgroup b.
// ---------------------------------------------------------------------------------------------------------
// PAIRED: `keys` is an array of (key, value) uint2 records (depth sort: one 8-byte scattered store per element and pass
// instead of two 4-byte ones -- scattered stores are what a scatter workgroup spends its time on).
template <int BITS, bool PAIRED>
__global__ __launch_bounds__(GSR_BLOCK) void radix_hist_kernel(const uint32_t *__restrict__ keys,
                                                               const uint32_t *__restrict__ n_ptr,
                                                               uint32_t *__restrict__ table, int nb_stride, int shift,
                                                               uint32_t mask) {
    constexpr int BINS = 1 << BITS;
    const uint32_t n = *n_ptr;
    const uint32_t base = blockIdx.x * (uint32_t)GSR_SORT_CHUNK;
    if (base >= n) return;
    __shared__ uint32_t s_h[BINS];
    for (int i = (int)threadIdx.x; i < BINS; i += GSR_BLOCK) s_h[i] = 0u;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < GSR_SORT_ITEMS; r++) {
        const uint32_t i = base + (uint32_t)r * GSR_BLOCK + threadIdx.x;
        if (i < n) atomicAdd(&s_h[(keys[PAIRED ? 2u * i : i] >> shift) & mask], 1u);
    }
    __syncthreads();
    for (int d = (int)threadIdx.x; d < BINS; d += GSR_BLOCK) table[(size_t)d * nb_stride + blockIdx.x] = s_h[d];
}

// one workgroup per row: exclusive scan of the row over its live entries, row total -> totals[row]
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

// Stable scatter.  Each wave first counts the digits of ITS quarter (LDS atomics), the counts become per-wave
// running cursors (global digit start + earlier workgroups + earlier waves), and the ranking loop then runs
// without any workgroup barrier: lanes holding the same digit find each other with BITS ballots, the rank inside
// the group is a popcount of the lanes below, the group's first lane advances the cursor.  LDS operations of one
// wave retire in order, so round r+1 sees the cursors round r left behind.
// PAIRED: keys_in / keys_out are uint2 (key, value) records (vals_in unused); LAST (paired only): the final pass
// writes only the values, to vals_out.
template <int BITS, bool PAIRED, bool LAST>
__global__ __launch_bounds__(GSR_BLOCK) void radix_scatter_kernel(
    const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, uint32_t *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out, const uint32_t *__restrict__ n_ptr, const uint32_t *__restrict__ table,
    const uint32_t *__restrict__ totals, int nb_stride, int shift, uint32_t mask, int nbits) {
    constexpr int BINS = 1 << BITS;
    constexpr int PER = BINS / GSR_BLOCK;        // bins owned by one thread when sweeping the digit space
    constexpr int WAVE_ITEMS = GSR_SORT_CHUNK / 4;  // 512 keys per wave, 8 rounds of 64
    const uint32_t n = *n_ptr;
    const uint32_t base = blockIdx.x * (uint32_t)GSR_SORT_CHUNK;
    if (base >= n) return;
    __shared__ uint32_t s_cur[4][BINS];
    __shared__ uint32_t s_w[4];
    const int tid = (int)threadIdx.x, wave = gsr_wave(), lane = gsr_lane();
    const uint64_t lt = gsr_lanemask_lt();
    for (int i = tid; i < 4 * BINS; i += GSR_BLOCK) (&s_cur[0][0])[i] = 0u;
    __syncthreads();
    const uint32_t wbase = base + (uint32_t)wave * WAVE_ITEMS;
    // the wave's keys and values stay in registers for the whole kernel: every load is issued up front and the
    // ranking rounds below never wait for memory
    constexpr int ROUNDS = WAVE_ITEMS / 64;
    uint32_t rkey[ROUNDS], rval[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const uint32_t i = wbase + (uint32_t)r * 64u + (uint32_t)lane;
        if (PAIRED) {
            const uint2 kv = i < n ? reinterpret_cast<const uint2 *>(keys_in)[i] : make_uint2(0u, 0u);
            rkey[r] = kv.x;
            rval[r] = kv.y;
        } else {
            rkey[r] = i < n ? keys_in[i] : 0u;
            rval[r] = i < n ? vals_in[i] : 0u;
        }
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const uint32_t i = wbase + (uint32_t)r * 64u + (uint32_t)lane;
        if (i < n) atomicAdd(&s_cur[wave][(rkey[r] >> shift) & mask], 1u);
    }
    __syncthreads();
    {
        // start of digit d for this workgroup = (keys with a smaller digit) + (digit d in earlier workgroups);
        // thread `tid` owns the PER consecutive digits tid*PER .. tid*PER+PER-1
        uint32_t t[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            t[k] = totals[tid * PER + k];
            sum += t[k];
        }
        uint32_t all;
        uint32_t run = gsr_block_incl_scan(sum, s_w, all) - sum;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int d = tid * PER + k;
            const uint32_t c0 = s_cur[0][d], c1 = s_cur[1][d], c2 = s_cur[2][d];
            const uint32_t start = run + table[(size_t)d * nb_stride + blockIdx.x];
            s_cur[0][d] = start;
            s_cur[1][d] = start + c0;
            s_cur[2][d] = start + c0 + c1;
            s_cur[3][d] = start + c0 + c1 + c2;
            run += t[k];
        }
    }
    __syncthreads();
    uint32_t *cur = s_cur[wave];
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const uint32_t i = wbase + (uint32_t)r * 64u + (uint32_t)lane;
        const bool valid = i < n;
        const uint32_t key = rkey[r], val = rval[r];
        const uint32_t d = (key >> shift) & mask;
        uint64_t same = __builtin_amdgcn_ballot_w64(valid);
        for (int b = 0; b < nbits; b++) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __builtin_amdgcn_ballot_w64(bit);
            same &= bit ? bal : ~bal;
        }
        const uint32_t rank = (uint32_t)__popcll(same & lt);
        if (valid) {
            const uint32_t pos = cur[d] + rank;
            if (PAIRED && !LAST) {
                reinterpret_cast<uint2 *>(keys_out)[pos] = make_uint2(key, val);
            } else if (PAIRED) {
                vals_out[pos] = val;
            } else {
                keys_out[pos] = key;
                vals_out[pos] = val;
            }
        }
        __builtin_amdgcn_wave_barrier();  // every lane has read its cursor before the group leader moves it
        if (valid && rank == 0u) cur[d] += (uint32_t)__popcll(same);
        __builtin_amdgcn_wave_barrier();
    }
}

// tiles touched by each chunk of GSR_SORT_CHUNK depth-ordered Gaussians (large-grid fallback path)
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

template <int BITS>
int radix_pass(uint32_t *kin, uint32_t *vin, uint32_t *kout, uint32_t *vout, const uint32_t *n_ptr, int nb, int shift,
               int nbits, uint32_t *table, uint32_t *totals, bool debug, hipStream_t stream) {
    const uint32_t mask = (1u << nbits) - 1u;
    hipLaunchKernelGGL((radix_hist_kernel<BITS, false>), dim3(nb), dim3(GSR_BLOCK), 0, stream, kin, n_ptr, table, nb, shift,
                       mask);
    if (int e = gsr_check_launch("radix_hist", debug, stream)) return e;
    hipLaunchKernelGGL(radix_rowscan_kernel, dim3(1 << BITS), dim3(GSR_BLOCK), 0, stream, table, n_ptr, nb, totals,
                       GSR_SORT_CHUNK);
    if (int e = gsr_check_launch("radix_rowscan", debug, stream)) return e;
    hipLaunchKernelGGL((radix_scatter_kernel<BITS, false, false>), dim3(nb), dim3(GSR_BLOCK), 0, stream, kin, vin, kout, vout, n_ptr,
                       table, totals, nb, shift, mask, nbits);
    return gsr_check_launch("radix_scatter", debug, stream);
}

// one pass over (key, value) records; `last`: write only the values to `order`
template <int BITS>
int radix_pass_paired(const uint2 *in, uint2 *out, uint32_t *order, bool last, const uint32_t *n_ptr, int nb, int shift,
                      int nbits, uint32_t *table, uint32_t *totals, bool debug, hipStream_t stream) {
    const uint32_t mask = (1u << nbits) - 1u;
    const uint32_t *kin = reinterpret_cast<const uint32_t *>(in);
    uint32_t *kout = reinterpret_cast<uint32_t *>(out);
    hipLaunchKernelGGL((radix_hist_kernel<BITS, true>), dim3(nb), dim3(GSR_BLOCK), 0, stream, kin, n_ptr, table, nb,
                       shift, mask);
    if (int e = gsr_check_launch("radix_hist", debug, stream)) return e;
    hipLaunchKernelGGL(radix_rowscan_kernel, dim3(1 << BITS), dim3(GSR_BLOCK), 0, stream, table, n_ptr, nb, totals,
                       GSR_SORT_CHUNK);
    if (int e = gsr_check_launch("radix_rowscan", debug, stream)) return e;
    if (last)
        hipLaunchKernelGGL((radix_scatter_kernel<BITS, true, true>), dim3(nb), dim3(GSR_BLOCK), 0, stream, kin,
                           (const uint32_t *)nullptr, kout, order, n_ptr, table, totals, nb, shift, mask, nbits);
    else
        hipLaunchKernelGGL((radix_scatter_kernel<BITS, true, false>), dim3(nb), dim3(GSR_BLOCK), 0, stream, kin,
                           (const uint32_t *)nullptr, kout, order, n_ptr, table, totals, nb, shift, mask, nbits);
    return gsr_check_launch("radix_scatter", debug, stream);
}

}  // namespace

// Exclusive scan of each of `rows` table rows over its live entries (ceil(*n_ptr / chunk)); row totals -> totals.
int gsr_launch_rowscan(uint32_t *table, const uint32_t *n_ptr, int nb_stride, int chunk, int rows, uint32_t *totals,
                       bool debug, hipStream_t stream) {
    hipLaunchKernelGGL(radix_rowscan_kernel, dim3(rows), dim3(GSR_BLOCK), 0, stream, table, n_ptr, nb_stride, totals,
                       chunk);
    return gsr_check_launch("rowscan", debug, stream);
}

int gsr_radix_passes(int bits, int bits_per_pass) { return (bits + bits_per_pass - 1) / bits_per_pass; }

// Stable LSD radix sort of (key, val) pairs on the low `bits` bits, `bits_per_pass` (8 or 11) bits at a time.
// Input in side `start_side`; the result lands in side start_side ^ (passes & 1).  The table needs
// (1 << bits_per_pass) * ceil(n_max / GSR_SORT_CHUNK) entries, totals (1 << bits_per_pass).
int gsr_radix_sort_u32(uint32_t *key[2], uint32_t *val[2], const uint32_t *n_ptr, int64_t n_max, int bits,
                       int bits_per_pass, int start_side, uint32_t *table, uint32_t *totals, bool debug,
                       hipStream_t stream) {
    if (bits_per_pass != 8 && bits_per_pass != 11) {
        gsr_set_error("radix sort: bits_per_pass must be 8 or 11");
        return GSR_E_INVALID;
    }
    const int nb = gsr_div_up(n_max > 0 ? n_max : 1, GSR_SORT_CHUNK);
    int src = start_side;
    for (int shift = 0; shift < bits; shift += bits_per_pass) {
        const int nbits = (bits - shift) < bits_per_pass ? (bits - shift) : bits_per_pass;
        int e;
        if (bits_per_pass == 8)
            e = radix_pass<8>(key[src], val[src], key[src ^ 1], val[src ^ 1], n_ptr, nb, shift, nbits, table, totals,
                              debug, stream);
        else
            e = radix_pass<11>(key[src], val[src], key[src ^ 1], val[src ^ 1], n_ptr, nb, shift, nbits, table, totals,
                               debug, stream);
        if (e) return e;
        src ^= 1;
    }
    return GSR_OK;
}

// block_counts -> exclusive offsets + V; index-ordered compaction; depth sort (32-bit float keys, 3 passes of
// 11 bits) on (key, index) records.  The sorted depth order ends in g.order.
int gsr_launch_compact_and_depth_sort(int32_t P, const GeomState &g, bool debug, hipStream_t stream) {
    const int nb1 = GeomState::prep_blocks(P);
    hipLaunchKernelGGL(compact_kernel, dim3(nb1), dim3(GSR_BLOCK), 0, stream, P, g.tiles_touched, g.splat,
                       g.block_counts, g.pair[0], g.hdr);
    if (int e = gsr_check_launch("compact", debug, stream)) return e;
    // (key, index) records ping-pong between pair[0] and pair[1]; the last pass leaves the bare indices in g.order
    const int nb = GeomState::sort_blocks(P);
    const int passes = gsr_radix_passes(32, GSR_DEPTH_RADIX_BITS);
    for (int k = 0; k < passes; k++) {
        const int shift = k * GSR_DEPTH_RADIX_BITS;
        const int nbits = (32 - shift) < GSR_DEPTH_RADIX_BITS ? (32 - shift) : GSR_DEPTH_RADIX_BITS;
        if (int e = radix_pass_paired<GSR_DEPTH_RADIX_BITS>(g.pair[k & 1], g.pair[(k & 1) ^ 1], g.order, k == passes - 1,
                                                            &g.hdr->V, nb, shift, nbits, g.sort_table, g.sort_totals,
                                                            debug, stream))
            return e;
    }
    return GSR_OK;
}

// per-chunk tile counts in depth order -> exclusive chunk offsets, R (clamped against the capacity)
int gsr_launch_tile_offsets(int32_t P, const GeomState &g, uint32_t r_capacity, bool debug, hipStream_t stream) {
    const int nb = GeomState::sort_blocks(P);
    hipLaunchKernelGGL(tile_blocksum_kernel, dim3(nb), dim3(GSR_BLOCK), 0, stream, g.order, g.tiles_touched, g.hdr,
                       g.tile_bsum);
    if (int e = gsr_check_launch("tile_blocksum", debug, stream)) return e;
    hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(GSR_BLOCK), 0, stream, g.tile_bsum, 0, &g.hdr->V,
                       GSR_SORT_CHUNK, g.hdr, 1, r_capacity);
    return gsr_check_launch("scan_tile_bsum", debug, stream);
}
