// stages.hip -- the integer-only stages of the pipeline for gfx950:
//   * lightweight analysis      (role of readOperations, reference include/common.cuh:321-459)
//   * row -> kernel-class binning (role of the reference's load balancer,
//     include/GPU/spECK_HashLoadBalancer.cuh:265-347 + scan_largearray_kernel.cuh:182-281;
//     done here as per-block histogram -> single-block scan -> ORDERED scatter: deterministic,
//     rows stay ascending inside every class)
//   * exclusive scan of the per-row counts into C.row_offsets
//     (role of cub::DeviceScan::ExclusiveSum, reference source/GPU/Multiply.cu:570)
// The scatter kernels emit one 32-byte RowRec per row (row id, A-row bounds, C-row base, column
// range): a class kernel then needs ONE coalesced load per row instead of a chain
// bin_rows -> row_offsets -> {c_ro, col_min, col_max}.
// No kernel here issues a global atomic: same-cache-line device atomics cost ~12 ns each on
// MI355X and serialise, so every block leaves a BlockPartial behind (plain stores) and one
// single-block kernel folds them.
#include "device_common.hpp"
#include "launch.hpp"

namespace speck {

static inline u32 cdiv(u64 a, u64 b) { return (u32)((a + b - 1) / b); }

constexpr int kChunk = 256;  // rows per sub-chunk of the analysis / symbolic scatter

// rows per block for the analysis / symbolic-scatter pair: contiguous, multiple of kChunk
static inline void row_chunking(u32 m, u32* rows_per_block, u32* blocks)
{
    u32 r = cdiv(m ? m : 1, 1024);
    r = (r + kChunk - 1u) & ~u32(kChunk - 1);
    *rows_per_block = r;
    *blocks = cdiv(m ? m : 1, r);
}

// --------------------------------------------------------------------------------
// Analysis, nnz-parallel: a block walks its rows in sub-chunks of 256; inside a sub-chunk
// the threads stride over the A ENTRIES (not the rows), so every thread has independent
// 3-deep load chains (A.col -> B.rowptr pair -> first/last B.col) in flight and long rows
// cost nothing extra.  Per-row results are combined with LDS atomics on distinct addresses.
// HBM traffic (algorithmic): 4(m+1) + 4 nnzA + 8 nnzA [B.rowptr pair] + 8 nnzA
// [first/last col of the B row] + 17 m written.
// --------------------------------------------------------------------------------
__global__ __launch_bounds__(kChunk) void analysis_kernel(
    const u32* __restrict__ a_ro, const u32* __restrict__ a_col, const u32* __restrict__ b_ro,
    const u32* __restrict__ b_col, u32 m, u32 rows_per_block, u32* __restrict__ row_ops,
    u32* __restrict__ row_max_ops, u32* __restrict__ row_col_min, u32* __restrict__ row_col_max,
    u8* __restrict__ sym_cls, u32* __restrict__ counts, BlockPartial* __restrict__ partials,
    ClassifyParams cp, u32* __restrict__ b_start, u32* __restrict__ b_len)
{
    constexpr int NW = kChunk / 64;
    constexpr int U = 4;
    const u32 e_base = a_ro[0];  // A may be a row-range view with absolute offsets
    __shared__ u32 s_ro[kChunk + 1];
    __shared__ u64 s_ops[kChunk];
    __shared__ u32 s_mx[kChunk], s_cmin[kChunk], s_cmax[kChunk];
    __shared__ u64 s_products[NW];
    __shared__ u32 s_max[NW];
    __shared__ u32 s_hist[NW][kMaxClasses];
    __shared__ u64 s_bytes[kMaxClasses];
    const u32 t = threadIdx.x;
    if (t < kMaxClasses) s_bytes[t] = 0;

    const u32 row_begin = blockIdx.x * rows_per_block;
    const u32 row_end = min(m, row_begin + rows_per_block);
    u64 my_products = 0;
    u32 my_max = 0;
    u32 hist[SYM_CLASSES];
#pragma unroll
    for (int c = 0; c < SYM_CLASSES; ++c) hist[c] = 0;

    for (u32 row0 = row_begin; row0 < row_end; row0 += kChunk) {
        const u32 nrows = min((u32)kChunk, row_end - row0);
        __syncthreads();
        if (t <= nrows) s_ro[t] = a_ro[row0 + t];
        if (t == 0 && nrows == kChunk) s_ro[kChunk] = a_ro[row0 + kChunk];
        s_ops[t] = 0;
        s_mx[t] = 0;
        s_cmin[t] = 0xFFFFFFFFu;
        s_cmax[t] = 0;
        __syncthreads();
        const u32 e_begin = s_ro[0], e_end = s_ro[nrows];
        for (u32 e0 = e_begin + t; e0 < e_end; e0 += kChunk * U) {
            u32 bs[U], be[U], first[U], last[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const u32 e = e0 + u * kChunk;
                ok[u] = e < e_end;
                const u32 k = ok[u] ? a_col[e] : 0u;
                bs[u] = ok[u] ? b_ro[k] : 0u;
                be[u] = ok[u] ? b_ro[k + 1] : 0u;
                if (ok[u] && b_start) {  // hand the B-row bounds to the symbolic / numeric kernels
                    b_start[e - e_base] = bs[u];
                    b_len[e - e_base] = be[u] - bs[u];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool has = ok[u] && be[u] > bs[u];
                first[u] = has ? b_col[bs[u]] : 0xFFFFFFFFu;
                last[u] = has ? b_col[be[u] - 1] : 0u;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!ok[u]) continue;
                const u32 e = e0 + u * kChunk;
                // local row of entry e: largest r with s_ro[r] <= e
                u32 lo = 0, hi = nrows;
                while (hi - lo > 1) {
                    const u32 mid = (lo + hi) >> 1;
                    if (s_ro[mid] <= e) lo = mid; else hi = mid;
                }
                const u32 len = be[u] - bs[u];
                if (len) {
                    atomicAdd(&s_ops[lo], (u64)len);
                    atomicMax(&s_mx[lo], len);
                    atomicMin(&s_cmin[lo], first[u]);
                    atomicMax(&s_cmax[lo], last[u]);
                }
            }
        }
        __syncthreads();
        u8 cls = SYM_NONE;
        if (t < nrows) {
            const u32 row = row0 + t;
            const u64 ops = s_ops[t];
            const u32 ops32 = ops > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)ops;
            const u32 len_a = s_ro[t + 1] - s_ro[t];
            const u32 cmin = s_cmin[t], cmax = s_cmax[t];
            if (row_ops) row_ops[row] = ops32;
            if (row_max_ops) row_max_ops[row] = s_mx[t];
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
    const u32 wid = t >> 6;
    __syncthreads();
    if (lane_id() == 0) {
        s_products[wid] = my_products;
        s_max[wid] = my_max;
#pragma unroll
        for (int c = 0; c < SYM_CLASSES; ++c) s_hist[wid][c] = hist[c];
    }
    __syncthreads();
    if (t == 0) {
        u64 p = 0;
        u32 mxv = 0;
        for (int w = 0; w < NW; ++w) {
            p += s_products[w];
            mxv = max(mxv, s_max[w]);
        }
        partials[blockIdx.x].products = p;
        partials[blockIdx.x].max_val = mxv;
    }
    if (t < kMaxClasses) {
        u32 h = 0;
        if (t < SYM_CLASSES)
            for (int w = 0; w < NW; ++w) h += s_hist[w][t];
        partials[blockIdx.x].count[t] = h;
        partials[blockIdx.x].bytes[t] = s_bytes[t];
    }
}

// --------------------------------------------------------------------------------
// Fold the block partials: totals, per-class offsets and, per block, the base of each class
// inside the record array.  One workgroup, one wave per class (wave-level scans, no barriers
// inside).  For the numeric phase it also scans the per-tile nnz sums (the middle step of the
// row_offsets scan) and checks the assumptions of a replayed launch sequence.
// --------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void stats_kernel(const BlockPartial* __restrict__ parts, u32 nb,
                                                     int numeric, DeviceStats* __restrict__ st,
                                                     u32* __restrict__ blk_base, u32 allowed_mask,
                                                     u32* __restrict__ tile_off, u64 exact_nnz)
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
        if (!numeric) {
            u64 p = 0;
            for (u32 i = lane; i < nb; i += 64) p += parts[i].products;
            p = wave_reduce_add(p);
            if (lane == 0) st->sum_products = p;
        } else {
            // exclusive scan of the per-tile nnz sums -> offset of each tile's first row
            u64 carry = 0;
            for (u32 base = 0; base < nb; base += 64) {
                const u32 i = base + lane;
                const u64 v = i < nb ? parts[i].products : 0;
                u64 incl = v;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const u64 t = __shfl_up(incl, off, 64);
                    if (lane >= (u32)off) incl += t;
                }
                if (i < nb) tile_off[i] = (u32)(carry + incl - v);
                carry += __shfl(incl, 63, 64);
            }
            if (lane == 0) {
                st->nnz_c = carry;
                if (carry > 0xFFFFFFFFull) st->nnz_overflow = 1;
                if (exact_nnz != ~0ull && carry != exact_nnz) st->capacity_miss = 1;
            }
        }
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
            // a replayed launch sequence only carries the kernels of `allowed_mask`
            if (s_total[c] && !((allowed_mask >> c) & 1u)) st->capacity_miss = 1;
        }
        t.offset[kMaxClasses] = run;
    }
}

// Ordered scatter for the symbolic phase: block b re-reads the classes of its rows and writes
// each row's record at class_offset + block_base + rank (rank from ballots, ascending rows).
__global__ __launch_bounds__(kChunk) void sym_scatter_kernel(
    const u8* __restrict__ cls, u32 m, u32 rows_per_block, const DeviceStats* __restrict__ st,
    const u32* __restrict__ blk_base, const u32* __restrict__ a_ro, const u32* __restrict__ row_ops,
    const u32* __restrict__ row_col_min, const u32* __restrict__ row_col_max,
    RowRec* __restrict__ recs)
{
    constexpr int NW = kChunk / 64;
    __shared__ u32 s_wcnt[SYM_CLASSES][NW];
    __shared__ u32 s_run[SYM_CLASSES];
    const u32 lane = lane_id(), wid = threadIdx.x >> 6;
    if (threadIdx.x < SYM_CLASSES)
        s_run[threadIdx.x] = st->sym.offset[threadIdx.x] +
                             blk_base[size_t(blockIdx.x) * kMaxClasses + threadIdx.x];
    __syncthreads();
    const u32 row_begin = blockIdx.x * rows_per_block;
    const u32 row_end = min(m, row_begin + rows_per_block);
    for (u32 row0 = row_begin; row0 < row_end; row0 += kChunk) {
        const u32 row = row0 + threadIdx.x;
        const u32 c = row < row_end ? cls[row] : 0xFFu;
        u32 my_rank = 0;
#pragma unroll
        for (u32 b = 0; b < SYM_CLASSES; ++b) {
            const u64 mask = __ballot(c == b);
            if (lane == 0) s_wcnt[b][wid] = __popcll(mask);
            if (c == b) my_rank = __popcll(mask & lanemask_lt());
        }
        __syncthreads();
        if (c < SYM_CLASSES) {
            u32 pos = s_run[c] + my_rank;
            for (u32 w = 0; w < wid; ++w) pos += s_wcnt[c][w];
            RowRec r;
            r.row = row;
            r.a0 = a_ro[row];
            r.a1 = a_ro[row + 1];
            r.base = 0;
            r.cmin = row_col_min[row];
            r.cmax = row_col_max[row];
            r.ops = row_ops[row];
            r.nnz = 0;
            recs[pos] = r;
        }
        __syncthreads();
        if (threadIdx.x < SYM_CLASSES) {
            u32 add = 0;
            for (int w = 0; w < NW; ++w) add += s_wcnt[threadIdx.x][w];
            s_run[threadIdx.x] += add;
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------
// row_offsets scan, fused with the numeric classification and scatter.  Tile = 2048 rows,
// thread t owns 8 consecutive rows.
//   num_count_kernel : per tile  nnz sum + class histogram (+ max row nnz), class per row
//   stats_kernel     : scan of the tile sums, class offsets / per-tile class bases
//   num_apply_kernel : row_offsets = tile offset + local scan (in place) and the RowRec of
//                      every row at its class position (ascending rows inside a class)
// Traffic: 8(m+1) B for the scan itself + 4m re-read of the counts + 33 m for
// classes and records.
// --------------------------------------------------------------------------------
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;
constexpr int kFieldBits = 12;  // per-class counters packed 5 per u64 (<= 512 per wave)

__device__ __forceinline__ void packed_add(u64& lo, u64& hi, u32 cls)
{
    if (cls < 5) lo += 1ull << (kFieldBits * cls); else hi += 1ull << (kFieldBits * (cls - 5));
}
__device__ __forceinline__ u32 packed_get(u64 lo, u64 hi, u32 cls)
{
    return (u32)((cls < 5 ? lo >> (kFieldBits * cls) : hi >> (kFieldBits * (cls - 5))) & 0xFFFu);
}

__global__ __launch_bounds__(kScanThreads) void num_count_kernel(
    const u32* __restrict__ counts, u32 m, const u32* __restrict__ a_ro,
    const u32* __restrict__ row_ops, const u32* __restrict__ row_col_min,
    const u32* __restrict__ row_col_max, u8* __restrict__ num_cls,
    BlockPartial* __restrict__ partials, ClassifyParams cp, u32 vsize)
{
    constexpr int NW = kScanThreads / 64;
    __shared__ u64 s_sum[NW];
    __shared__ u32 s_hist[NW][kMaxClasses];
    __shared__ u64 s_bytes[kMaxClasses];
    __shared__ u32 s_max[NW];
    if (threadIdx.x < kMaxClasses) s_bytes[threadIdx.x] = 0;
    __syncthreads();
    const u64 base = u64(blockIdx.x) * kScanTile + u64(threadIdx.x) * kScanItems;
    u64 tsum = 0, packed_lo = 0, packed_hi = 0;
    u32 my_max = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        const u64 row = base + i;
        if (row < m) {
            const u32 c = counts[row];
            tsum += c;
            my_max = max(my_max, c);
            if (num_cls) {
                const u32 len_a = a_ro[row + 1] - a_ro[row];
                const u8 cls = classify_numeric(len_a, c, row_col_min[row], row_col_max[row], cp);
                num_cls[row] = cls;
                if (cls != NUM_NONE) {
                    packed_add(packed_lo, packed_hi, cls);
                    if (cp.want_bytes)
                        atomicAdd(&s_bytes[cls], numeric_row_bytes(len_a, row_ops[row], c, vsize));
                }
            }
        }
    }
    tsum = wave_reduce_add(tsum);
    packed_lo = wave_reduce_add(packed_lo);
    packed_hi = wave_reduce_add(packed_hi);
    my_max = wave_reduce_max(my_max);
    const u32 wid = threadIdx.x >> 6;
    if (lane_id() == 0) {
        s_sum[wid] = tsum;
        s_max[wid] = my_max;
#pragma unroll
        for (int k = 0; k < kMaxClasses; ++k)
            s_hist[wid][k] = k < 10 ? packed_get(packed_lo, packed_hi, k) : 0u;
    }
    __syncthreads();
    if (threadIdx.x < kMaxClasses) {
        u32 h = 0;
        for (int w = 0; w < NW; ++w) h += s_hist[w][threadIdx.x];
        partials[blockIdx.x].count[threadIdx.x] = h;
        partials[blockIdx.x].bytes[threadIdx.x] = s_bytes[threadIdx.x];
    }
    if (threadIdx.x == 0) {
        u64 s = 0;
        u32 mxv = 0;
        for (int w = 0; w < NW; ++w) {
            s += s_sum[w];
            mxv = max(mxv, s_max[w]);
        }
        partials[blockIdx.x].products = s;  // numeric phase: the tile's nnz sum
        partials[blockIdx.x].max_val = mxv;
    }
}

__global__ __launch_bounds__(kScanThreads) void num_apply_kernel(
    u32* __restrict__ counts_inout, u32 m, const u32* __restrict__ tile_off,
    const DeviceStats* __restrict__ st, const u32* __restrict__ blk_base,
    const u8* __restrict__ num_cls, const u32* __restrict__ a_ro, const u32* __restrict__ row_ops,
    const u32* __restrict__ row_col_min, const u32* __restrict__ row_col_max,
    RowRec* __restrict__ recs)
{
    constexpr int NW = kScanThreads / 64;
    __shared__ u32 s_scan[NW + 1];
    __shared__ u64 s_wlo[NW], s_whi[NW];
    const u32 lane = lane_id(), wid = threadIdx.x >> 6;
    const u64 base = u64(blockIdx.x) * kScanTile + u64(threadIdx.x) * kScanItems;
    u32 c[kScanItems];
    u32 tsum = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        c[i] = (base + i) < m ? counts_inout[base + i] : 0;
        tsum += c[i];
    }
    u32 total;
    const u32 excl = block_exclusive_scan<kScanThreads>(tsum, s_scan, &total);
    u32 run = tile_off[blockIdx.x] + excl;
    u32 off[kScanItems];
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        off[i] = run;
        if (base + i < m) counts_inout[base + i] = run;
        run += c[i];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) counts_inout[m] = (u32)st->nnz_c;
    if (!num_cls) return;

    // class of my 8 rows, packed per-thread histogram, exclusive scan over the threads
    u8 cls[kScanItems];
    u64 plo = 0, phi = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        cls[i] = (base + i) < m ? num_cls[base + i] : (u8)NUM_NONE;
        if (cls[i] != NUM_NONE) packed_add(plo, phi, cls[i]);
    }
    u64 ilo = plo, ihi = phi;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u64 tl = __shfl_up(ilo, o, 64), th = __shfl_up(ihi, o, 64);
        if (lane >= (u32)o) {
            ilo += tl;
            ihi += th;
        }
    }
    if (lane == 63) {
        s_wlo[wid] = ilo;
        s_whi[wid] = ihi;
    }
    __syncthreads();
    u64 blo = ilo - plo, bhi = ihi - phi;  // exclusive inside the wave
    for (u32 w = 0; w < wid; ++w) {
        blo += s_wlo[w];
        bhi += s_whi[w];
    }
    u64 used_lo = 0, used_hi = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        const u64 row = base + i;
        if (cls[i] == NUM_NONE) continue;
        const u32 k = cls[i];
        const u32 pos = st->num.offset[k] + blk_base[size_t(blockIdx.x) * kMaxClasses + k] +
                        packed_get(blo, bhi, k) + packed_get(used_lo, used_hi, k);
        packed_add(used_lo, used_hi, k);
        RowRec r;
        r.row = (u32)row;
        r.a0 = a_ro[row];
        r.a1 = a_ro[row + 1];
        r.base = off[i];
        r.cmin = row_col_min[row];
        r.cmax = row_col_max[row];
        r.ops = row_ops[row];
        r.nnz = c[i];
        recs[pos] = r;
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

void launch_analysis(hipStream_t s, const u32* a_ro, const u32* a_col, const u32* b_ro,
                     const u32* b_col, u32 m, u64 /*nnz_a*/, u32* row_ops, u32* row_max_ops,
                     u32* row_col_min, u32* row_col_max, u8* sym_cls, u32* counts,
                     BlockPartial* partials, u32* blk_base, RowRec* recs, DeviceStats* st,
                     const ClassifyParams& cp, u32* b_start, u32* b_len)
{
    u32 rows_per_block, blocks;
    row_chunking(m, &rows_per_block, &blocks);
    hipLaunchKernelGGL(analysis_kernel, dim3(blocks), dim3(kChunk), 0, s, a_ro, a_col, b_ro, b_col, m,
                       rows_per_block, row_ops, row_max_ops, row_col_min, row_col_max, sym_cls, counts,
                       partials, cp, b_start, b_len);
    hipLaunchKernelGGL(stats_kernel, dim3(1), dim3(1024), 0, s, partials, blocks, 0, st, blk_base,
                       cp.sym_allowed, (u32*)nullptr, ~0ull);
    if (sym_cls)
        hipLaunchKernelGGL(sym_scatter_kernel, dim3(blocks), dim3(kChunk), 0, s, sym_cls, m,
                           rows_per_block, st, blk_base, a_ro, row_ops, row_col_min, row_col_max, recs);
}

void launch_scan(hipStream_t s, u32* counts_inout, u32 m, u32* tile_off, const u32* a_ro,
                 const u32* row_ops, const u32* row_col_min, const u32* row_col_max, u8* num_cls,
                 BlockPartial* partials, u32* blk_base, RowRec* recs, DeviceStats* st,
                 const ClassifyParams& cp, u32 vsize, u64 exact_nnz)
{
    const u32 tiles = scan_tiles(m);
    hipLaunchKernelGGL(num_count_kernel, dim3(tiles), dim3(kScanThreads), 0, s, counts_inout, m, a_ro,
                       row_ops, row_col_min, row_col_max, num_cls, partials, cp, vsize);
    hipLaunchKernelGGL(stats_kernel, dim3(1), dim3(1024), 0, s, partials, tiles, 1, st, blk_base,
                       num_cls ? cp.num_allowed : 0xFFFFFFFFu, tile_off, exact_nnz);
    hipLaunchKernelGGL(num_apply_kernel, dim3(tiles), dim3(kScanThreads), 0, s, counts_inout, m, tile_off,
                       st, blk_base, (const u8*)num_cls, a_ro, row_ops, row_col_min, row_col_max, recs);
}

}  // namespace speck
