// stages.hip -- the integer-only stages of the pipeline for gfx950:
//   * lightweight analysis      (role of readOperations, reference include/common.cuh:321-459)
//   * row -> kernel-class binning (role of the reference's load balancer,
//     include/GPU/spECK_HashLoadBalancer.cuh:265-347 + scan_largearray_kernel.cuh:182-281;
//     done here as histogram -> offsets -> ordered scatter, wave64 ballots)
//   * exclusive scan of the per-row counts into C.row_offsets
//     (role of cub::DeviceScan::ExclusiveSum, reference source/GPU/Multiply.cu:570)
// All kernels are grid-stride with a bounded grid so that global atomics stay
// at O(grid) instead of O(rows).
#include "device_common.hpp"
#include "launch.hpp"

namespace speck {

// --------------------------------------------------------------------------------
// Analysis: one lane group (2^group_shift lanes) per row of A.
// HBM traffic (algorithmic): 4(m+1) + 4 nnzA + 8 nnzA [B.rowptr pair] + 8 nnzA
// [first/last col of the B row] + 13 m written.
// --------------------------------------------------------------------------------
template <int THREADS>
__global__ __launch_bounds__(THREADS) void analysis_kernel(
    const u32* __restrict__ a_ro, const u32* __restrict__ a_col, const u32* __restrict__ b_ro,
    const u32* __restrict__ b_col, u32 m, u32 group_shift, u32* __restrict__ row_ops,
    u32* __restrict__ row_max_ops, u32* __restrict__ row_col_min, u32* __restrict__ row_col_max,
    u8* __restrict__ sym_cls, u32* __restrict__ counts, DeviceStats* __restrict__ st,
    ClassifyParams cp)
{
    __shared__ u64 s_products;
    __shared__ u32 s_max_ops;
    __shared__ u32 s_hist[8];
    __shared__ u64 s_bytes[8];
    if (threadIdx.x == 0) {
        s_products = 0;
        s_max_ops = 0;
    }
    if (threadIdx.x < 8) {
        s_hist[threadIdx.x] = 0;
        s_bytes[threadIdx.x] = 0;
    }
    __syncthreads();

    const u32 G = 1u << group_shift;
    const u32 rows_per_block = THREADS >> group_shift;
    const u32 gl = threadIdx.x & (G - 1);
    const u32 gsub = threadIdx.x >> group_shift;
    u64 my_products = 0;
    u32 my_max = 0;

    for (u64 row0 = u64(blockIdx.x) * rows_per_block; row0 < m;
         row0 += u64(gridDim.x) * rows_per_block) {
        const u64 row = row0 + gsub;
        u64 ops = 0;
        u32 mx = 0, cmin = 0xFFFFFFFFu, cmax = 0, len_a = 0;
        if (row < m) {
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
        if (gl == 0 && row < m) {
            const u32 ops32 = ops > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)ops;
            if (row_ops) row_ops[row] = ops32;
            if (row_max_ops) row_max_ops[row] = mx;
            if (row_col_min) row_col_min[row] = cmin;
            if (row_col_max) row_col_max[row] = cmax;
            my_products += ops;
            my_max = max(my_max, ops32);
            if (sym_cls) {
                const u8 cls = classify_symbolic(len_a, ops32, cmin, cmax, cp);
                sym_cls[row] = cls;
                if (cls == SYM_NONE) {
                    // empty row, or a single A entry: the C row is a scaled copy of one B row
                    counts[row] = ops32;
                } else {
                    atomicAdd(&s_hist[cls], 1u);
                    atomicAdd(&s_bytes[cls], symbolic_row_bytes(len_a, ops32));
                }
            }
        }
    }
    if (my_products) atomicAdd(&s_products, my_products);
    if (my_max) atomicMax(&s_max_ops, my_max);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_products) atomicAdd(&st->sum_products, s_products);
        if (s_max_ops) atomicMax(&st->max_row_ops, s_max_ops);
    }
    if (threadIdx.x < 8 && sym_cls && s_hist[threadIdx.x]) {
        atomicAdd(&st->sym_count[threadIdx.x], s_hist[threadIdx.x]);
        atomicAdd(&st->sym_bytes[threadIdx.x], s_bytes[threadIdx.x]);
    }
}

// --------------------------------------------------------------------------------
// Binning: counts -> offsets (one thread), then an ordered scatter of row ids.
// --------------------------------------------------------------------------------
__global__ void bin_offsets_kernel(DeviceStats* st, int numeric)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        u32* cnt = numeric ? st->num_count : st->sym_count;
        u32* off = numeric ? st->num_offset : st->sym_offset;
        u32* cur = numeric ? st->num_cursor : st->sym_cursor;
        u32 run = 0;
        for (int b = 0; b < 8; ++b) {
            off[b] = run;
            cur[b] = 0;
            run += cnt[b];
        }
        off[8] = run;
    }
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void bin_scatter_kernel(const u8* __restrict__ cls, u32 m,
                                                              DeviceStats* __restrict__ st,
                                                              int numeric,
                                                              u32* __restrict__ bin_rows)
{
    constexpr int NW = THREADS / 64;
    __shared__ u32 s_wcnt[8][NW];
    __shared__ u32 s_base[8];
    const u32* off = numeric ? st->num_offset : st->sym_offset;
    u32* cur = numeric ? st->num_cursor : st->sym_cursor;
    const u32 lane = lane_id(), wid = threadIdx.x >> 6;

    for (u64 row0 = u64(blockIdx.x) * THREADS; row0 < m; row0 += u64(gridDim.x) * THREADS) {
        const u64 row = row0 + threadIdx.x;
        const u32 c = row < m ? cls[row] : 0xFFu;
        u32 my_rank = 0;
#pragma unroll
        for (u32 b = 0; b < 8; ++b) {
            const u64 mask = __ballot(c == b);
            if (lane == 0) s_wcnt[b][wid] = __popcll(mask);
            if (c == b) my_rank = __popcll(mask & lanemask_lt());
        }
        __syncthreads();
        if (threadIdx.x < 8) {
            u32 run = 0;
            for (int w = 0; w < NW; ++w) {
                const u32 t = s_wcnt[threadIdx.x][w];
                s_wcnt[threadIdx.x][w] = run;
                run += t;
            }
            s_base[threadIdx.x] = run ? off[threadIdx.x] + atomicAdd(&cur[threadIdx.x], run) : 0;
        }
        __syncthreads();
        if (c < 8) bin_rows[s_base[c] + s_wcnt[c][wid] + my_rank] = (u32)row;
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------
// Exclusive scan of counts[0..m) in place -> row_offsets[0..m]; three kernels
// (tile reduce, scan of tile sums, apply), tile = THREADS * ITEMS rows.
// The apply kernel also classifies every row for the numeric phase (it is the
// first place where the exact nnz of a C row is known next to its offset).
// Traffic: 8(m+1) B for the scan itself + 4m re-read of the tile.
// --------------------------------------------------------------------------------
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;

__global__ __launch_bounds__(kScanThreads) void scan_reduce_kernel(const u32* __restrict__ counts,
                                                                   u32 m, u64* __restrict__ tile_sums)
{
    __shared__ u64 s_sum;
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
    const u64 base = u64(blockIdx.x) * kScanTile;
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        const u64 idx = base + u64(i) * kScanThreads + threadIdx.x;
        if (idx < m) acc += counts[idx];
    }
    acc = wave_reduce_add(acc);
    if (lane_id() == 0) atomicAdd(&s_sum, acc);
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = s_sum;
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
    u8* __restrict__ num_cls, DeviceStats* __restrict__ st, ClassifyParams cp, u32 vsize)
{
    __shared__ u32 s_scan[kScanThreads / 64 + 1];
    __shared__ u32 s_hist[8];
    __shared__ u64 s_bytes[8];
    __shared__ u32 s_max;
    if (threadIdx.x < 8) {
        s_hist[threadIdx.x] = 0;
        s_bytes[threadIdx.x] = 0;
    }
    if (threadIdx.x == 0) s_max = 0;
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
                    atomicAdd(&s_hist[cls], 1u);
                    atomicAdd(&s_bytes[cls], numeric_row_bytes(len_a, row_ops[row], c[i], vsize));
                }
            }
        }
    }
    if (my_max) atomicMax(&s_max, my_max);
    __syncthreads();
    if (threadIdx.x < 8 && s_hist[threadIdx.x]) {
        atomicAdd(&st->num_count[threadIdx.x], s_hist[threadIdx.x]);
        atomicAdd(&st->num_bytes[threadIdx.x], s_bytes[threadIdx.x]);
    }
    if (threadIdx.x == 0) {
        if (s_max) atomicMax(&st->max_row_nnz_c, s_max);
        if (blockIdx.x == gridDim.x - 1) counts_inout[m] = (u32)(st->nnz_c);
    }
}

// --------------------------------------------------------------------------------
// host launchers
// --------------------------------------------------------------------------------
static inline u32 cdiv(u64 a, u64 b) { return (u32)((a + b - 1) / b); }

void launch_analysis(hipStream_t s, const u32* a_ro, const u32* a_col, const u32* b_ro,
                     const u32* b_col, u32 m, u64 nnz_a, u32* row_ops, u32* row_max_ops,
                     u32* row_col_min, u32* row_col_max, u8* sym_cls, u32* counts, DeviceStats* st,
                     const ClassifyParams& cp, int max_blocks)
{
    constexpr int THREADS = 256;
    // lanes per row ~ average row length of A, rounded up to a power of two
    const u64 avg = m ? (nnz_a + m - 1) / m : 1;
    u32 shift = 0;
    while ((1ull << shift) < avg && shift < 6) ++shift;
    const u32 rows_per_block = THREADS >> shift;
    u32 blocks = cdiv(m, rows_per_block);
    if (blocks > (u32)max_blocks) blocks = max_blocks;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(analysis_kernel<THREADS>, dim3(blocks), dim3(THREADS), 0, s, a_ro, a_col, b_ro,
                       b_col, m, shift, row_ops, row_max_ops, row_col_min, row_col_max, sym_cls,
                       counts, st, cp);
}

void launch_binning(hipStream_t s, const u8* cls, u32 m, DeviceStats* st, int numeric, u32* bin_rows,
                    int max_blocks)
{
    hipLaunchKernelGGL(bin_offsets_kernel, dim3(1), dim3(64), 0, s, st, numeric);
    constexpr int THREADS = 256;
    u32 blocks = cdiv(m, THREADS);
    if (blocks > (u32)max_blocks) blocks = max_blocks;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(bin_scatter_kernel<THREADS>, dim3(blocks), dim3(THREADS), 0, s, cls, m, st,
                       numeric, bin_rows);
}

size_t scan_scratch_bytes(u32 m) { return size_t(cdiv(m ? m : 1, kScanTile)) * sizeof(u64); }

void launch_scan(hipStream_t s, u32* counts_inout, u32 m, u64* tile_sums, const u32* a_ro,
                 const u32* row_ops, const u32* row_col_min, const u32* row_col_max, u8* num_cls,
                 DeviceStats* st, const ClassifyParams& cp, u32 vsize)
{
    const u32 tiles = cdiv(m ? m : 1, kScanTile);
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(tiles), dim3(kScanThreads), 0, s, counts_inout, m,
                       tile_sums);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(1024), 0, s, tile_sums, tiles, st);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(tiles), dim3(kScanThreads), 0, s, counts_inout, m,
                       tile_sums, a_ro, row_ops, row_col_min, row_col_max, num_cls, st, cp, vsize);
}

}  // namespace speck
