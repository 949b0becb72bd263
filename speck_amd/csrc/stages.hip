// stages.hip -- the integer-only stages of the pipeline for gfx950:
//   * lightweight analysis      (role of readOperations, reference include/common.cuh:321-459)
//   * row -> kernel-class binning (role of the reference's load balancer,
//     include/GPU/spECK_HashLoadBalancer.cuh:265-347 + scan_largearray_kernel.cuh:182-281;
//     done here as per-block histogram -> single-block scan -> ORDERED scatter with wave64
//     ballots: deterministic, rows stay ascending inside every class)
//   * exclusive scan of the per-row counts into C.row_offsets
//     (role of cub::DeviceScan::ExclusiveSum, reference source/GPU/Multiply.cu:570)
// No kernel here issues a global atomic: same-cache-line device atomics cost ~12 ns each on
// MI355X and serialise, so every block leaves a BlockPartial behind (plain stores) and one
// single-block kernel folds them.
#include "device_common.hpp"
#include "launch.hpp"

namespace speck {

static inline u32 cdiv(u64 a, u64 b) { return (u32)((a + b - 1) / b); }

// rows per block for the analysis / symbolic-scatter pair: contiguous, multiple of 256
static inline void row_chunking(u32 m, u32* rows_per_block, u32* blocks)
{
    u32 r = cdiv(m ? m : 1, 1024);
    r = (r + 255u) & ~255u;
    *rows_per_block = r;
    *blocks = cdiv(m ? m : 1, r);
}

// --------------------------------------------------------------------------------
// Analysis: one lane group (2^group_shift lanes) per row of A; block b owns the contiguous
// rows [b*R, (b+1)*R).
// HBM traffic (algorithmic): 4(m+1) + 4 nnzA + 8 nnzA [B.rowptr pair] + 8 nnzA
// [first/last col of the B row] + 13 m written.
// --------------------------------------------------------------------------------
template <int THREADS>
__global__ __launch_bounds__(THREADS) void analysis_kernel(
    const u32* __restrict__ a_ro, const u32* __restrict__ a_col, const u32* __restrict__ b_ro,
    const u32* __restrict__ b_col, u32 m, u32 rows_per_block, u32 group_shift,
    u32* __restrict__ row_ops, u32* __restrict__ row_max_ops, u32* __restrict__ row_col_min,
    u32* __restrict__ row_col_max, u8* __restrict__ sym_cls, u32* __restrict__ counts,
    BlockPartial* __restrict__ partials, ClassifyParams cp)
{
    constexpr int NW = THREADS / 64;
    __shared__ u64 s_products[NW];
    __shared__ u32 s_max[NW];
    __shared__ u32 s_hist[NW][kMaxClasses];
    __shared__ u64 s_bytes[kMaxClasses];
    if (threadIdx.x < kMaxClasses) s_bytes[threadIdx.x] = 0;
    __syncthreads();

    const u32 G = 1u << group_shift;
    const u32 rows_per_iter = THREADS >> group_shift;
    const u32 gl = threadIdx.x & (G - 1);
    const u32 gsub = threadIdx.x >> group_shift;
    const u32 row_begin = blockIdx.x * rows_per_block;
    const u32 row_end = min(m, row_begin + rows_per_block);
    u64 my_products = 0;
    u32 my_max = 0;
    u32 hist[SYM_CLASSES];
#pragma unroll
    for (int c = 0; c < SYM_CLASSES; ++c) hist[c] = 0;

    for (u32 row0 = row_begin; row0 < row_end; row0 += rows_per_iter) {
        const u32 row = row0 + gsub;
        u64 ops = 0;
        u32 mx = 0, cmin = 0xFFFFFFFFu, cmax = 0, len_a = 0;
        if (row < row_end) {
            const u32 a0 = a_ro[row], a1 = a_ro[row + 1];
            len_a = a1 - a0;
            for (u32 ia = a0 + gl; ia < a1; ia += G) {
                const u32 k = a_col[ia];
                const u32 bs = b_ro[k], be = b_ro[k + 1];
                const u32 len = be - bs;
                ops += len;
                mx = max(mx, len);
                if (len) {
                    cmin = min(cmin, b_col[bs]);
                    cmax = max(cmax, b_col[be - 1]);
                }
            }
        }
        for (u32 off = G >> 1; off > 0; off >>= 1) {
            ops += __shfl_xor(ops, off, 64);
            mx = max(mx, (u32)__shfl_xor((int)mx, off, 64));
            cmin = min(cmin, (u32)__shfl_xor((int)cmin, off, 64));
            cmax = max(cmax, (u32)__shfl_xor((int)cmax, off, 64));
        }
        u8 cls = SYM_NONE;
        if (gl == 0 && row < row_end) {
            const u32 ops32 = ops > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)ops;
            if (row_ops) row_ops[row] = ops32;
            if (row_max_ops) row_max_ops[row] = mx;
            if (row_col_min) row_col_min[row] = cmin;
            if (row_col_max) row_col_max[row] = cmax;
            my_products += ops;
            my_max = max(my_max, ops32);
            if (sym_cls) {
                cls = classify_symbolic(len_a, ops32, cmin, cmax, cp);
                sym_cls[row] = cls;
                if (cls == SYM_NONE) {
                    // empty row, or a single A entry: the C row is a scaled copy of one B row
                    counts[row] = ops32;
                } else if (cp.want_bytes) {
                    atomicAdd(&s_bytes[cls], symbolic_row_bytes(len_a, ops32));
                }
            }
        }
        if (sym_cls) {
#pragma unroll
            for (int c = 0; c < SYM_CLASSES; ++c) hist[c] += __popcll(__ballot(cls == c));
        }
    }
    my_products = wave_reduce_add(my_products);
    my_max = wave_reduce_max(my_max);
    const u32 wid = threadIdx.x >> 6;
    if (lane_id() == 0) {
        s_products[wid] = my_products;
        s_max[wid] = my_max;
#pragma unroll
        for (int c = 0; c < SYM_CLASSES; ++c) s_hist[wid][c] = hist[c];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 p = 0;
        u32 mxv = 0;
        for (int w = 0; w < NW; ++w) {
            p += s_products[w];
            mxv = max(mxv, s_max[w]);
        }
        partials[blockIdx.x].products = p;
        partials[blockIdx.x].max_val = mxv;
    }
    if (threadIdx.x < kMaxClasses) {
        u32 h = 0;
        if (threadIdx.x < SYM_CLASSES)
            for (int w = 0; w < NW; ++w) h += s_hist[w][threadIdx.x];
        partials[blockIdx.x].count[threadIdx.x] = h;
        partials[blockIdx.x].bytes[threadIdx.x] = s_bytes[threadIdx.x];
    }
}

// --------------------------------------------------------------------------------
// Fold the block partials: totals, per-class offsets and, per block, the base of each class
// inside bin_rows.  One workgroup, one wave per class (wave-level scans, no barriers inside).
// --------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void stats_kernel(const BlockPartial* __restrict__ parts, u32 nb,
                                                     int numeric, DeviceStats* __restrict__ st,
                                                     u32* __restrict__ blk_base)
{
    __shared__ u32 s_total[kMaxClasses];
    __shared__ u64 s_bytes[kMaxClasses];
    const u32 lane = lane_id(), w = threadIdx.x >> 6;
    if (w < kMaxClasses) {
        u32 carry = 0;
        u64 bytes = 0;
        for (u32 base = 0; base < nb; base += 64) {
            const u32 i = base + lane;
            const u32 v = i < nb ? parts[i].count[w] : 0;
            if (i < nb) bytes += parts[i].bytes[w];
            const u32 incl = wave_inclusive_scan(v);
            if (i < nb) blk_base[size_t(i) * kMaxClasses + w] = carry + incl - v;
            carry += (u32)__shfl((int)incl, 63, 64);
        }
        bytes = wave_reduce_add(bytes);
        if (lane == 0) {
            s_total[w] = carry;
            s_bytes[w] = bytes;
        }
    } else if (w == kMaxClasses) {
        u64 p = 0;
        for (u32 i = lane; i < nb; i += 64) p += parts[i].products;
        p = wave_reduce_add(p);
        if (lane == 0 && !numeric) st->sum_products = p;
    } else if (w == kMaxClasses + 1) {
        u32 mx = 0;
        for (u32 i = lane; i < nb; i += 64) mx = max(mx, parts[i].max_val);
        mx = wave_reduce_max(mx);
        if (lane == 0) {
            if (numeric) st->max_row_nnz_c = mx; else st->max_row_ops = mx;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        BinTable& t = numeric ? st->num : st->sym;
        u32 run = 0;
        for (int c = 0; c < kMaxClasses; ++c) {
            t.count[c] = s_total[c];
            t.offset[c] = run;
            t.bytes[c] = s_bytes[c];
            run += s_total[c];
        }
        t.offset[kMaxClasses] = run;
    }
}

// Ordered scatter of row ids into bin_rows: block b re-reads the classes of its rows and
// places each row at class_offset + block_base + rank (rank from ballots, ascending rows).
template <int THREADS>
__global__ __launch_bounds__(THREADS) void bin_scatter_kernel(const u8* __restrict__ cls, u32 m,
                                                              u32 rows_per_block,
                                                              const DeviceStats* __restrict__ st,
                                                              int numeric,
                                                              const u32* __restrict__ blk_base,
                                                              u32* __restrict__ bin_rows)
{
    constexpr int NW = THREADS / 64;
    __shared__ u32 s_wcnt[kMaxClasses][NW];
    __shared__ u32 s_run[kMaxClasses];
    const BinTable& t = numeric ? st->num : st->sym;
    const u32 lane = lane_id(), wid = threadIdx.x >> 6;
    if (threadIdx.x < kMaxClasses)
        s_run[threadIdx.x] = t.offset[threadIdx.x] + blk_base[size_t(blockIdx.x) * kMaxClasses + threadIdx.x];
    __syncthreads();
    const u32 row_begin = blockIdx.x * rows_per_block;
    const u32 row_end = min(m, row_begin + rows_per_block);
    for (u32 row0 = row_begin; row0 < row_end; row0 += THREADS) {
        const u32 row = row0 + threadIdx.x;
        const u32 c = row < row_end ? cls[row] : 0xFFu;
        u32 my_rank = 0;
#pragma unroll
        for (u32 b = 0; b < kMaxClasses; ++b) {
            const u64 mask = __ballot(c == b);
            if (lane == 0) s_wcnt[b][wid] = __popcll(mask);
            if (c == b) my_rank = __popcll(mask & lanemask_lt());
        }
        __syncthreads();
        u32 pos = 0;
        if (c < kMaxClasses) {
            pos = s_run[c] + my_rank;
            for (u32 w = 0; w < wid; ++w) pos += s_wcnt[c][w];
        }
        __syncthreads();
        if (threadIdx.x < kMaxClasses) {
            u32 add = 0;
            for (int w = 0; w < NW; ++w) add += s_wcnt[threadIdx.x][w];
            s_run[threadIdx.x] += add;
        }
        if (c < kMaxClasses) bin_rows[pos] = row;
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------
// Exclusive scan of counts[0..m) in place -> row_offsets[0..m]; three kernels (tile reduce,
// scan of tile sums, apply), tile = 2048 rows.  The apply kernel also classifies every row for
// the numeric phase (the first place where the exact nnz of a C row is known).
// Traffic: 8(m+1) B for the scan itself + 4m re-read of the tile.
// --------------------------------------------------------------------------------
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;

__global__ __launch_bounds__(kScanThreads) void scan_reduce_kernel(const u32* __restrict__ counts,
                                                                   u32 m, u64* __restrict__ tile_sums)
{
    __shared__ u64 s_sum[kScanThreads / 64];
    const u64 base = u64(blockIdx.x) * kScanTile;
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        const u64 idx = base + u64(i) * kScanThreads + threadIdx.x;
        if (idx < m) acc += counts[idx];
    }
    acc = wave_reduce_add(acc);
    if (lane_id() == 0) s_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 s = 0;
        for (int w = 0; w < kScanThreads / 64; ++w) s += s_sum[w];
        tile_sums[blockIdx.x] = s;
    }
}

// single workgroup: exclusive scan over the tile sums (sequential chunks of 1024)
__global__ __launch_bounds__(1024) void scan_tiles_kernel(u64* __restrict__ tile_sums, u32 tiles,
                                                          DeviceStats* __restrict__ st)
{
    __shared__ u64 s_wave[17];
    __shared__ u64 s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const u32 lane = lane_id(), wid = threadIdx.x >> 6;
    for (u32 base = 0; base < tiles; base += 1024) {
        const u32 idx = base + threadIdx.x;
        const u64 v = idx < tiles ? tile_sums[idx] : 0;
        u64 incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u64 t = __shfl_up(incl, off, 64);
            if (lane >= (u32)off) incl += t;
        }
        if (lane == 63) s_wave[wid] = incl;
        __syncthreads();
        if (threadIdx.x == 0) {
            u64 run = s_carry;
            for (int w = 0; w < 16; ++w) {
                const u64 t = s_wave[w];
                s_wave[w] = run;
                run += t;
            }
            s_wave[16] = run;
        }
        __syncthreads();
        if (idx < tiles) tile_sums[idx] = s_wave[wid] + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) s_carry = s_wave[16];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        st->nnz_c = s_carry;
        if (s_carry > 0xFFFFFFFFull) st->nnz_overflow = 1;
    }
}

__global__ __launch_bounds__(kScanThreads) void scan_apply_kernel(
    u32* __restrict__ counts_inout, u32 m, const u64* __restrict__ tile_sums,
    const u32* __restrict__ a_ro, const u32* __restrict__ row_ops,
    const u32* __restrict__ row_col_min, const u32* __restrict__ row_col_max,
    u8* __restrict__ num_cls, BlockPartial* __restrict__ partials,
    const DeviceStats* __restrict__ st, ClassifyParams cp, u32 vsize)
{
    constexpr int NW = kScanThreads / 64;
    __shared__ u32 s_scan[NW + 1];
    __shared__ u32 s_hist[NW][kMaxClasses];
    __shared__ u64 s_bytes[kMaxClasses];
    __shared__ u32 s_max[NW];
    if (threadIdx.x < kMaxClasses) s_bytes[threadIdx.x] = 0;
    __syncthreads();

    // thread t owns ITEMS consecutive rows of the tile (blocked arrangement)
    const u64 base = u64(blockIdx.x) * kScanTile + u64(threadIdx.x) * kScanItems;
    u32 c[kScanItems];
    u32 tsum = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        c[i] = (base + i) < m ? counts_inout[base + i] : 0;
        tsum += c[i];
    }
    u32 total;
    u32 excl = block_exclusive_scan<kScanThreads>(tsum, s_scan, &total);
    u32 run = (u32)tile_sums[blockIdx.x] + excl;
    u32 my_max = 0;
    // per-thread class counters, 12 bits per class (<= 8 per thread, <= 512 per wave)
    u64 packed_lo = 0, packed_hi = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        const u64 row = base + i;
        if (row < m) {
            counts_inout[row] = run;
            run += c[i];
            my_max = max(my_max, c[i]);
            if (num_cls) {
                const u32 len_a = a_ro[row + 1] - a_ro[row];
                const u8 cls = classify_numeric(len_a, c[i], row_col_min[row], row_col_max[row], cp);
                num_cls[row] = cls;
                if (cls != NUM_NONE) {
                    if (cls < 5) packed_lo += 1ull << (12 * cls); else packed_hi += 1ull << (12 * (cls - 5));
                    if (cp.want_bytes)
                        atomicAdd(&s_bytes[cls], numeric_row_bytes(len_a, row_ops[row], c[i], vsize));
                }
            }
        }
    }
    packed_lo = wave_reduce_add(packed_lo);
    packed_hi = wave_reduce_add(packed_hi);
    my_max = wave_reduce_max(my_max);
    const u32 wid = threadIdx.x >> 6;
    if (lane_id() == 0) {
        s_max[wid] = my_max;
#pragma unroll
        for (int k = 0; k < kMaxClasses; ++k) {
            const u64 src = k < 5 ? packed_lo >> (12 * k) : packed_hi >> (12 * (k - 5));
            s_hist[wid][k] = k < 10 ? (u32)(src & 0xFFFu) : 0u;
        }
    }
    __syncthreads();
    if (partials && threadIdx.x < kMaxClasses) {
        u32 h = 0;
        for (int w = 0; w < NW; ++w) h += s_hist[w][threadIdx.x];
        partials[blockIdx.x].count[threadIdx.x] = h;
        partials[blockIdx.x].bytes[threadIdx.x] = s_bytes[threadIdx.x];
    }
    if (threadIdx.x == 0) {
        if (partials) {
            u32 mxv = 0;
            for (int w = 0; w < NW; ++w) mxv = max(mxv, s_max[w]);
            partials[blockIdx.x].max_val = mxv;
            partials[blockIdx.x].products = 0;
        }
        if (blockIdx.x == gridDim.x - 1) counts_inout[m] = (u32)(st->nnz_c);
    }
}

// --------------------------------------------------------------------------------
// host launchers
// --------------------------------------------------------------------------------
u32 analysis_blocks(u32 m)
{
    u32 r, b;
    row_chunking(m, &r, &b);
    return b;
}
u32 scan_tiles(u32 m) { return cdiv(m ? m : 1, kScanTile); }
size_t scan_scratch_bytes(u32 m) { return size_t(scan_tiles(m)) * sizeof(u64); }

void launch_analysis(hipStream_t s, const u32* a_ro, const u32* a_col, const u32* b_ro,
                     const u32* b_col, u32 m, u64 nnz_a, u32* row_ops, u32* row_max_ops,
                     u32* row_col_min, u32* row_col_max, u8* sym_cls, u32* counts,
                     BlockPartial* partials, u32* blk_base, u32* bin_rows, DeviceStats* st,
                     const ClassifyParams& cp)
{
    constexpr int THREADS = 256;
    // lanes per row ~ average row length of A, rounded up to a power of two
    const u64 avg = m ? (nnz_a + m - 1) / m : 1;
    u32 shift = 0;
    while ((1ull << shift) < avg && shift < 6) ++shift;
    u32 rows_per_block, blocks;
    row_chunking(m, &rows_per_block, &blocks);
    hipLaunchKernelGGL(analysis_kernel<THREADS>, dim3(blocks), dim3(THREADS), 0, s, a_ro, a_col, b_ro,
                       b_col, m, rows_per_block, shift, row_ops, row_max_ops, row_col_min, row_col_max,
                       sym_cls, counts, partials, cp);
    hipLaunchKernelGGL(stats_kernel, dim3(1), dim3(1024), 0, s, partials, blocks, 0, st, blk_base);
    if (sym_cls)
        hipLaunchKernelGGL(bin_scatter_kernel<THREADS>, dim3(blocks), dim3(THREADS), 0, s, sym_cls, m,
                           rows_per_block, st, 0, blk_base, bin_rows);
}

void launch_scan(hipStream_t s, u32* counts_inout, u32 m, u64* tile_sums, const u32* a_ro,
                 const u32* row_ops, const u32* row_col_min, const u32* row_col_max, u8* num_cls,
                 BlockPartial* partials, u32* blk_base, u32* bin_rows, DeviceStats* st,
                 const ClassifyParams& cp, u32 vsize)
{
    const u32 tiles = scan_tiles(m);
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(tiles), dim3(kScanThreads), 0, s, counts_inout, m,
                       tile_sums);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(1024), 0, s, tile_sums, tiles, st);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(tiles), dim3(kScanThreads), 0, s, counts_inout, m,
                       tile_sums, a_ro, row_ops, row_col_min, row_col_max, num_cls,
                       num_cls ? partials : nullptr, st, cp, vsize);
    if (num_cls) {
        hipLaunchKernelGGL(stats_kernel, dim3(1), dim3(1024), 0, s, partials, tiles, 1, st, blk_base);
        hipLaunchKernelGGL(bin_scatter_kernel<kScanThreads>, dim3(tiles), dim3(kScanThreads), 0, s, num_cls,
                           m, (u32)kScanTile, st, 1, blk_base, bin_rows);
    }
}

}  // namespace speck
