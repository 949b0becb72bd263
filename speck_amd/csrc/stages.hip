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
#include <type_traits>

#include "device_common.hpp"
#include "launch.hpp"

namespace speck {

static inline u32 cdiv(u64 a, u64 b) { return (u32)((a + b - 1) / b); }

typedef u32 RowPtrPair __attribute__((ext_vector_type(2), aligned(4)));
#ifdef SPECK_PHASE_CLOCKS
static __device__ unsigned long long g_an_clk[1024 * 8];
#define AN_BEGIN() long long an_t_ = clock64()
#define AN_MARK(i_) \
    do { \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
        const long long an_n_ = clock64(); \
        if (lane_id() == 0) atomicAdd(&g_an_clk[(blockIdx.x % 1024) * 8 + (i_)], (unsigned long long)(an_n_ - an_t_)); \
        an_t_ = an_n_; \
    } while (0)
#else
#define AN_BEGIN()
#define AN_MARK(i_)
#endif
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
// Analysis, nnz-parallel and WAVE-granular: a block owns a contiguous row range, each of its
// waves walks that range in sub-chunks of 64 rows, and inside a sub-chunk the lanes stride over
// the A ENTRIES (not the rows), so every lane has independent 3-deep load chains
// (A.col -> B.rowptr pair -> first/last B.col) in flight and long rows cost nothing extra.
// Nothing but the final histogram crosses a wave: no workgroup barrier inside the loops (with
// 256-row sub-chunks per workgroup, 75 % of the wave cycles of this kernel were barrier waits).
// Per-row results are combined with LDS atomics in the wave's own staging area.
// HBM traffic (algorithmic): 4(m+1) + 4 nnzA + 8 nnzA [B.rowptr pair] + 8 nnzA
// [first/last col of the B row] + 17 m written (+ 8 nnzA for b_start / b_len).
// --------------------------------------------------------------------------------
constexpr u32 kAnRowPathMax = 8;  // rows per lane up to this many entries (sub-chunk maximum)
// Two shapes, one kChunk of rows per pass of a block either way: 8 waves x 32 rows (the default) and 4 waves x 64 rows
// for inputs of short rows -- there a lane per row is the common path, 32 rows leave half a wave idle, and at 91 VGPRs
// (5 waves per SIMD) the 8 x 32 shape needs more wave slots than the chip has for ~200 k rows: two rounds of
// workgroups, each as long as its chain of dependent gathers (scircuit / mac_econ stand-ins).
constexpr u32 kAnCoopMax = 64;        // ... at most this many per workgroup (the others stay with their wave)
constexpr u32 kAnCoopRowLen = 256, kAnCoopEntries = 2048;  // sub-chunks (32 rows) with more entries than this, one row
                                                            //   holding more than that, are walked by the whole workgroup
// Input check (eager path; the long comment is further down): workgroup `vb` of `nvb` walks its share of B's
// entries -- col[e] < col[e + 1] unless e + 1 starts a row, looked up (binary search in the row offsets) only for the
// pairs that are NOT ascending, i.e. almost never -- and stores the call's epoch on a violation.
__device__ __forceinline__ void validate_b_slice(const u32* __restrict__ b_ro, const u32* __restrict__ b_col, u32 b_rows,
                                                 u32 b_cols, u32 vb, u32 nvb, u32 epoch, DeviceStats* __restrict__ st)
{
    const u32 e_first = b_ro[0], e_last = b_ro[b_rows];
    bool bad = e_last < e_first;
    for (u64 i = u64(vb) * blockDim.x + threadIdx.x; e_first + i < e_last; i += u64(nvb) * blockDim.x) {
        const u32 e = e_first + (u32)i;
        const u32 c = b_col[e];
        if (c >= b_cols) bad = true;
        if (e + 1 < e_last && b_col[e + 1] <= c) {
            u32 lo = 0, hi = b_rows;  // first row whose offset is >= e + 1
            while (lo < hi) {
                const u32 mid = lo + ((hi - lo) >> 1);
                if (b_ro[mid] < e + 1) lo = mid + 1; else hi = mid;
            }
            if (b_ro[lo] != e + 1) bad = true;
        }
    }
    if (__ballot(bad) != 0 && lane_id() == 0) st->b_bad_epoch = epoch;  // plain store: every writer stores the same value
}

// VERIFY (replayed sequence with the analysis OFF the critical path): nothing is written.  Every
// quantity this kernel would produce -- the (start, length) pair of every entry, ops / longest B row / column range / class
// of every row, A's row offsets -- is recomputed from the inputs as they are NOW and compared with what the previous
// identical call left in the arena (which the symbolic, scan and numeric kernels of this sequence are reading while this
// kernel runs beside them on its own stream); any difference raises capacity_miss and the eager path re-runs the call.
template <int NW, u32 R, bool VERIFY = false>
__global__ __launch_bounds__(NW * 64) void analysis_kernel(
    const u32* __restrict__ a_ro, const u32* __restrict__ a_col, const u32* __restrict__ b_ro,
    const u32* __restrict__ b_col, u32 m, u32 rows_per_block, u32* row_ops,
    u32* row_max_ops, u32* row_col_min, u32* row_col_max,
    u8* sym_cls, u32* __restrict__ counts, BlockPartial* __restrict__ partials,
    ClassifyParams cp, uint2* b_sl, DeviceStats* __restrict__ st, u32 b_rows,
    const u32* __restrict__ pred_block, const DeviceStats* __restrict__ pred_stats, RowRec* __restrict__ recs,
    u32 an_blocks, u32 b_cols, u32 validate_epoch, u32* a_ro_copy, u32* __restrict__ verdict)
{
    // workgroups behind the analysis grid (eager path): the input check of B, next to the analysis instead of in a
    // launch of its own behind it
    if (blockIdx.x >= an_blocks) {
        validate_b_slice(b_ro, b_col, b_rows, b_cols, blockIdx.x - an_blocks, gridDim.x - an_blocks, validate_epoch, st);
        return;
    }
    constexpr int kAnThreads = NW * 64;
    constexpr int U = 4;   // entries per lane and tile: 256 entries cover most 32-row sub-chunks in ONE
                           //   round of the dependent chain A.col -> B.rowptr -> B.col
    // R = rows per sub-chunk (one per lane when they are finalised): the kernel
                           //   lasts as long as its slowest wave, so the waves are kept short and many
    static_assert(NW * R == kChunk, "one pass of a block covers one kChunk of rows");
    static_assert(kChunk / 64 == 4, "sym_scatter_kernel sums four per-wave counters");
    // the statistics block of this call starts from zero (no memset node in the launch sequence;
    // nothing reads or writes it before the scatter kernel that follows)
    // (a replayed sequence whose symbolic binning is PREDICTED -- pred_block, below -- has no scatter kernel: its
    //  blocks raise the flags of the statistics block themselves, so block 0 must not wipe them.  They are zero
    //  when such a sequence starts: it only ever follows a call that completed -- or it finds a flag of a failed one,
    //  stops, and the eager path, which starts from zero, re-runs.  Block 0 then writes what the symbolic kernels
    //  read: the class table of the predicted call, which every block checks its own share of.)
    if (!VERIFY && blockIdx.x == 0 && !pred_block)
        for (u32 i = threadIdx.x; i < sizeof(DeviceStats) / 4 - 1; i += kAnThreads) reinterpret_cast<u32*>(st)[i] = 0;  // (all but b_bad_epoch)
    if (!VERIFY && blockIdx.x == 0 && pred_block) {
        constexpr u32 kWords = sizeof(BinTable) / 4;
        const u32* src = reinterpret_cast<const u32*>(&pred_stats->sym);
        u32* dst = reinterpret_cast<u32*>(&st->sym);
        for (u32 i = threadIdx.x; i < kWords; i += kAnThreads) dst[i] = src[i];
        if (threadIdx.x == 0) st->nf_entries = pred_stats->nf_entries;
    }
    const u32 e_base = a_ro[0];  // A may be a row-range view with absolute offsets
    __shared__ u32 s_ro_all[NW][R + 1];
    __shared__ u64 s_ops_all[NW][R];
    __shared__ u32 s_mx_all[NW][R], s_cmin_all[NW][R], s_cmax_all[NW][R];
    __shared__ u64 s_products[NW], s_nf[NW];
    __shared__ u32 s_max[NW], s_nfr[NW];
    __shared__ u32 s_hist[NW][kMaxClasses];
    __shared__ u64 s_bytes[kMaxClasses];
    const u32 t = threadIdx.x, lane = lane_id(), wid = t >> 6;
    AN_BEGIN();
    u32* s_ro = s_ro_all[wid];
    u64* s_ops = s_ops_all[wid];
    u32 *s_mx = s_mx_all[wid], *s_cmin = s_cmin_all[wid], *s_cmax = s_cmax_all[wid];
    if (t < kMaxClasses) s_bytes[t] = 0;
    __syncthreads();

    const u32 row_begin = blockIdx.x * rows_per_block;
    const u32 row_end = min(m, row_begin + rows_per_block);
    u64 my_products = 0, my_nf = 0;
    u32 my_max = 0, my_nfr = 0;
    bool bad_col = false;  // a column id of A beyond the rows of B: clamped here, reported through the partials
    bool bad_meta = false; // VERIFY: something differs from what the previous identical call left behind
    u32 hist[SYM_CLASSES];
#pragma unroll
    for (int c = 0; c < SYM_CLASSES; ++c) hist[c] = 0;

    // The entry-parallel walk of one sub-chunk: entries [e_first, e_end) in steps of e_step, 64 x U per tile, per-row
    // reductions into the LDS arrays of the sub-chunk (ro_ = its A row offsets).  A wave alone: e_first = e_begin +
    // lane, e_step = 64 U; the whole workgroup on one sub-chunk: e_first = e_begin + wid * 64 U + lane, e_step = NW * 64 U.
    auto walk_entries = [&](u32 e_first, u32 e_end, u32 e_step, u32 nrows, const u32* ro_, u64* ops_, u32* mx_, u32* cmin_,
                            u32* cmax_) {
        for (u32 t0 = e_first - lane; t0 < e_end; t0 += e_step) {  // (a tile at a time: uniform for the wave)
            const u32 e0 = t0 + lane;
            u32 bs[U], be[U], first[U], last[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const u32 e = e0 + u * 64;
                ok[u] = e < e_end;
                u32 k = ok[u] ? a_col[e] : 0u;
                if (k >= b_rows) {
                    bad_col = true;
                    k = 0;
                }
                // B.rowptr[k], B.rowptr[k+1] as ONE 8-byte gather (4-byte aligned): the random
                // gathers of this kernel are bound by addresses per cycle, not by bytes
                const RowPtrPair pr = *reinterpret_cast<const RowPtrPair*>(b_ro + k);
                bs[u] = ok[u] ? pr.x : 0u;
                be[u] = ok[u] ? pr.y : 0u;
                // hand the B-row bounds to the symbolic / numeric kernels
                if constexpr (VERIFY) {
                    if (ok[u]) {
                        const uint2 was = b_sl[e - e_base];
                        bad_meta |= was.x != bs[u] || was.y != be[u] - bs[u];
                    }
                } else if (ok[u] && b_sl) b_sl[e - e_base] = make_uint2(bs[u], be[u] - bs[u]);
            }
            AN_MARK(1);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool has = ok[u] && be[u] > bs[u];
                first[u] = has ? b_col[bs[u]] : 0xFFFFFFFFu;
                last[u] = has ? b_col[be[u] - 1] : 0u;
            }
            AN_MARK(2);
            // local row of each entry: largest r with ro_[r] <= e; the U searches advance in lock
            // step (independent LDS reads)
            u32 lo[U], hi[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                lo[u] = 0;
                hi[u] = nrows;
            }
#pragma unroll
            for (int step = 0; step < (R == 32 ? 5 : 6); ++step) {  // 2^steps = R
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const u32 mid = (lo[u] + hi[u]) >> 1;
                    const bool act = hi[u] - lo[u] > 1;
                    const bool le = ro_[mid] <= e0 + u * 64;
                    lo[u] = (act && le) ? mid : lo[u];
                    hi[u] = (act && !le) ? mid : hi[u];
                }
            }
            AN_MARK(3);
            // The lanes of a row are contiguous: reduce every run of equal rows inside its 16-lane DPP row first
            // (segmented scan, pure VALU) and let the LAST lane of the run issue the LDS atomics.  With all
            // entries of a long row adding to the same LDS word the atomics serialised (27 lanes per address
            // on the nlpkkt stand-in: most of that kernel's 2.3 ms, 81 % of its LDS cycles were conflicts).
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const u32 len = be[u] - bs[u];
                const bool live = ok[u] && len;
                const u32 key = ok[u] ? lo[u] : 0xFFFFFFFFu;  // entries with an empty B row stay inside their run
                u32 v_sum = live ? len : 0u, v_mx = v_sum, v_min = first[u], v_max = live ? last[u] : 0u;
#define SPECK_SEG_STEP(S_)                                                                        \
                {                                                                                 \
                    const bool same = dpp_move<kDppRowShr + S_>(0xFFFFFFFEu, key) == key;         \
                    const u32 t_sum = dpp_move<kDppRowShr + S_>(0u, v_sum);                       \
                    const u32 t_mx = dpp_move<kDppRowShr + S_>(0u, v_mx);                         \
                    const u32 t_min = dpp_move<kDppRowShr + S_>(0xFFFFFFFFu, v_min);              \
                    const u32 t_max = dpp_move<kDppRowShr + S_>(0u, v_max);                       \
                    v_sum += same ? t_sum : 0u;                                                   \
                    v_mx = same ? max(v_mx, t_mx) : v_mx;                                         \
                    v_min = same ? min(v_min, t_min) : v_min;                                     \
                    v_max = same ? max(v_max, t_max) : v_max;                                     \
                }
                SPECK_SEG_STEP(1)
                SPECK_SEG_STEP(2)
                SPECK_SEG_STEP(4)
                SPECK_SEG_STEP(8)
#undef SPECK_SEG_STEP
                const bool tail = dpp_move<kDppRowShl + 1>(0xFFFFFFFEu, key) != key;  // lane 15 of a row: no source
                if (tail && v_sum != 0u) {  // (a run of empty B rows only adds nothing; lanes past the tile hold 0)
                    atomicAdd(&ops_[key], (u64)v_sum);
                    atomicMax(&mx_[key], v_mx);
                    atomicMin(&cmin_[key], v_min);
                    atomicMax(&cmax_[key], v_max);
                }
            }
        }
    };
    // Rows of a sub-chunk from their LDS accumulators to the output arrays, class and block statistics (one lane
    // per row, executed by a whole wave: the class histogram is ballots).
    auto finish_rows = [&](u32 row0, u32 nrows, const u32* ro_, const u64* ops_, const u32* mx_, const u32* cmin_,
                           const u32* cmax_) {
        u8 cls = SYM_NONE;
        if (lane < nrows) {
            const u32 row = row0 + lane;
            const u64 ops = ops_[lane];
            const u32 ops32 = ops > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)ops;
            const u32 len_a = ro_[lane + 1] - ro_[lane];
            const u32 cmin = cmin_[lane], cmax = cmax_[lane];
            if constexpr (VERIFY) {
                bad_meta |= row_ops[row] != ops32 || row_max_ops[row] != mx_[lane] || row_col_min[row] != cmin ||
                            row_col_max[row] != cmax || sym_cls[row] != classify_symbolic(len_a, ops32, cmin, cmax, cp);
            } else {
            if (row_ops) row_ops[row] = ops32;
            if (row_max_ops) row_max_ops[row] = mx_[lane];
            if (row_col_min) row_col_min[row] = cmin;
            if (row_col_max) row_col_max[row] = cmax;
            my_products += ops;
            my_max = max(my_max, ops32);
            }
            if (!VERIFY && sym_cls) {
                cls = classify_symbolic(len_a, ops32, cmin, cmax, cp);
                sym_cls[row] = cls;
                if (cls == SYM_NF) {
                    my_nf += nf_slot_entries(cmin, cmax, ops32);  // scratch slot: nnz <= min(column range, products)
                    my_nfr = max(my_nfr, cmax - cmin + 1);
                }
                if (cls == SYM_GH) my_nf += gh_table_slots(ops32);  // ... = the row's key set in global memory
                if (cls == SYM_NONE) {
                    // empty row, or a single A entry: the C row is a scaled copy of one B row
                    counts[row] = ops32;
                } else if (cp.want_bytes) {
                    atomicAdd(&s_bytes[cls], symbolic_row_bytes(len_a, ops32));
                }
            }
        }
        if (!VERIFY && sym_cls) {
#pragma unroll
            for (int c = 0; c < SYM_CLASSES; ++c) hist[c] += __popcll(__ballot(cls == c));
        }
    };

    // Sub-chunks with a HUB row (power-law inputs: thousands of entries in one row) are not walked by
    // the wave that meets them -- its chain of dependent gathers would be the lifetime of the kernel -- but put on
    // a list and walked by ALL waves of the workgroup together afterwards (webbase stand-in: 170 -> 90 us; uniformly long rows stay with their waves: every sub-chunk of the cant stand-in on the list cost it 30 us).
    __shared__ u32 s_coop[kAnCoopMax];
    __shared__ u32 s_ncoop;
    if (t == 0) s_ncoop = 0;
    __syncthreads();
    for (u32 row0 = row_begin + wid * R; row0 < row_end; row0 += NW * R) {
        const u32 nrows = min(R, row_end - row0);
        wave_lds_fence();
        if (lane <= nrows) {
            const u32 v = a_ro[row0 + lane];
            s_ro[lane] = v;
            if constexpr (VERIFY) bad_meta |= a_ro_copy[row0 + lane] != v;
            else if (a_ro_copy) a_ro_copy[row0 + lane] = v;
        }
        if (lane == 0 && nrows == R) {
            const u32 v = a_ro[row0 + R];
            s_ro[R] = v;
            if constexpr (VERIFY) bad_meta |= a_ro_copy[row0 + R] != v;
            else if (a_ro_copy) a_ro_copy[row0 + R] = v;
        }
        if (lane < R) {
            s_ops[lane] = 0;
            s_mx[lane] = 0;
            s_cmin[lane] = 0xFFFFFFFFu;
            s_cmax[lane] = 0;
        }
        wave_lds_fence();
        AN_MARK(0);
        const u32 e_begin = s_ro[0], e_end = s_ro[nrows];
        // Sub-chunks of SHORT rows (two thirds of the webbase-like rows hold one entry): a lane per row, the
        // row's entries in registers -- no row search, no LDS atomics, and the entry-parallel tile below
        // would run at a fraction of its lanes (webbase stand-in: 172 -> ~60 us for this kernel).
        const u32 my_len = lane < nrows ? s_ro[lane + 1] - s_ro[lane] : 0u;
        const u32 max_len = wave_reduce_max(my_len);
        if (max_len > kAnCoopRowLen && e_end - e_begin > kAnCoopEntries * (R / 32)) {  // (uniform) a hub row: later, by the whole workgroup -- if the list has room
            u32 at = 0;
            if (lane == 0) at = atomicAdd(&s_ncoop, 1u);
            at = (u32)__builtin_amdgcn_readfirstlane((int)at);
            if (at < kAnCoopMax) {
                if (lane == 0) s_coop[at] = row0;
                continue;
            }
        }
        if (max_len <= kAnRowPathMax) {
            const u32 e_lo = lane < nrows ? s_ro[lane] : 0u;
            u64 r_ops = 0;
            u32 r_mx = 0, r_min = 0xFFFFFFFFu, r_max = 0;
            for (u32 j0 = 0; j0 < max_len; j0 += U) {
                u32 bs[U], be[U];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    ok[u] = j0 + u < my_len;
                    const u32 e = e_lo + j0 + u;
                    u32 k = ok[u] ? a_col[e] : 0u;
                    if (k >= b_rows) {
                        bad_col = true;
                        k = 0;
                    }
                    const RowPtrPair pr = *reinterpret_cast<const RowPtrPair*>(b_ro + k);
                    bs[u] = ok[u] ? pr.x : 0u;
                    be[u] = ok[u] ? pr.y : 0u;
                    if constexpr (VERIFY) {
                        if (ok[u]) {
                            const uint2 was = b_sl[e - e_base];
                            bad_meta |= was.x != bs[u] || was.y != be[u] - bs[u];
                        }
                    } else if (ok[u] && b_sl) b_sl[e - e_base] = make_uint2(bs[u], be[u] - bs[u]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const u32 len = be[u] - bs[u];
                    if (ok[u] && len) {
                        r_ops += len;
                        r_mx = max(r_mx, len);
                        r_min = min(r_min, b_col[bs[u]]);
                        r_max = max(r_max, b_col[be[u] - 1]);
                    }
                }
            }
            if (lane < nrows) {
                s_ops[lane] = r_ops;
                s_mx[lane] = r_mx;
                s_cmin[lane] = r_min;
                s_cmax[lane] = r_max;
            }
        } else
            walk_entries(e_begin + lane, e_end, 64 * U, nrows, s_ro, s_ops, s_mx, s_cmin, s_cmax);
        wave_lds_fence();
        AN_MARK(4);
        finish_rows(row0, nrows, s_ro, s_ops, s_mx, s_cmin, s_cmax);
    }
    // the listed sub-chunks, one after the other, by all waves (wave 0's staging arrays; workgroup barriers)
    __syncthreads();
    const u32 ncoop = min(s_ncoop, kAnCoopMax);
    for (u32 i = 0; i < ncoop; ++i) {
        const u32 row0 = s_coop[i];
        const u32 nrows = min(R, row_end - row0);
        u32* ro0 = s_ro_all[0];
        if (wid == 0) {
            if (lane <= nrows) ro0[lane] = a_ro[row0 + lane];
            if (lane == 0 && nrows == R) ro0[R] = a_ro[row0 + R];
            if (lane < R) {
                s_ops_all[0][lane] = 0;
                s_mx_all[0][lane] = 0;
                s_cmin_all[0][lane] = 0xFFFFFFFFu;
                s_cmax_all[0][lane] = 0;
            }
        }
        __syncthreads();
        walk_entries(ro0[0] + wid * 64 * U + lane, ro0[nrows], NW * 64 * U, nrows, ro0, s_ops_all[0], s_mx_all[0],
                     s_cmin_all[0], s_cmax_all[0]);
        __syncthreads();
        if (wid == 0) finish_rows(row0, nrows, ro0, s_ops_all[0], s_mx_all[0], s_cmin_all[0], s_cmax_all[0]);
        __syncthreads();
    }
    AN_MARK(5);
    if constexpr (VERIFY) {
        // (plain stores of the same value by whoever objects; the join before the sequence's ticket orders them)
        // (the verdict goes to pinned host memory: this kernel runs on a stream of its own beside the sequence, whose
        //  last kernel mirrors the statistics block -- the host reads both once both streams are idle)
        const bool any_bad = __ballot(bad_meta || bad_col) != 0, any_col = __ballot(bad_col) != 0;
        if (any_bad && lane == 0) __hip_atomic_fetch_or(verdict, any_col ? 3u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    my_products = wave_reduce_add(my_products);
    my_nf = wave_reduce_add(my_nf);
    my_max = wave_reduce_max(my_max);
    my_nfr = wave_reduce_max(my_nfr);
    if (__ballot(bad_col) != 0) my_nfr = 0xFFFFFFFFu;  // folded with max: survives to the scatter kernel
    __syncthreads();
    AN_MARK(6);
    if (lane == 0) {
        s_products[wid] = my_products;
        s_nf[wid] = my_nf;
        s_max[wid] = my_max;
        s_nfr[wid] = my_nfr;
#pragma unroll
        for (int c = 0; c < SYM_CLASSES; ++c) s_hist[wid][c] = hist[c];
    }
    __syncthreads();
    const PartialArrays pa(partials, an_blocks);
    if (t == 0) {
        u64 p = 0, nf = 0;
        u32 mxv = 0, nfr = 0;
        for (int w = 0; w < NW; ++w) {
            p += s_products[w];
            nf += s_nf[w];
            mxv = max(mxv, s_max[w]);
            nfr = max(nfr, s_nfr[w]);
        }
        pa.products[blockIdx.x] = p;
        pa.max_val[blockIdx.x] = mxv;
        pa.aux_max[blockIdx.x] = nfr;
        pa.g_ops[blockIdx.x] = nf;  // symbolic phase: scratch entries of the block's numeric-first rows
    }
    if (t < kMaxClasses) {
        u32 h = 0;
        if (t < SYM_CLASSES)
            for (int w = 0; w < NW; ++w) h += s_hist[w][t];
        pa.count[t * pa.cap + blockIdx.x] = h;
        if (cp.want_bytes) pa.bytes[t * pa.cap + blockIdx.x] = s_bytes[t];
    }
    if (!pred_block) return;

    // ---- predicted symbolic binning (replayed sequence): the scatter kernel's work for my rows, at the list
    // positions the previous identical call gave this block -- if my rows are, class by class, as many as then.
    __shared__ u32 s_wcnt[SYM_CLASSES][kChunk / 64];
    __shared__ u32 s_run[SYM_CLASSES];
    __shared__ u32 s_bad;
    const u32* tab = pred_block + size_t(blockIdx.x) * kPredBlockWords;
    if (t == 0) s_bad = 0;
    __syncthreads();
    if (t < SYM_CLASSES) {
        u32 h = 0;
        for (int w = 0; w < NW; ++w) h += s_hist[w][t];
        if (h != tab[kMaxClasses + t]) s_bad = 1;
        s_run[t] = tab[t];
    }
    if (t == 0) {
        u64 nf = 0;
        u32 nfr = 0;
        for (int w = 0; w < NW; ++w) {
            nf += s_nf[w];
            nfr = max(nfr, s_nfr[w]);
        }
        if (nf != ((u64(tab[2 * kMaxClasses + 1]) << 32) | tab[2 * kMaxClasses])) s_bad = 1;
        if (nfr == 0xFFFFFFFFu) {  // a column id of A >= rows(B)
            st->a_invalid = 1;
            s_bad = 1;
        }
    }
    __threadfence_block();  // my rows' classes and bounds (global, written by the waves of this block) before the reads below
    __syncthreads();
    if (s_bad) {  // (uniform) not the rows of the predicted call: nothing is written, the eager path re-runs
        if (t == 0) st->capacity_miss = 1;
        return;
    }
    for (u32 row0 = row_begin; row0 < row_end; row0 += kChunk) {
        const u32 row = row0 + t;
        const u32 c = (t < kChunk && row < row_end) ? sym_cls[row] : 0xFFu;
        u32 my_rank = 0;
        if (t < kChunk) {
#pragma unroll
            for (u32 b = 0; b < SYM_CLASSES; ++b) {
                const u64 mask = __ballot(c == b);
                if (lane == 0) s_wcnt[b][wid] = __popcll(mask);
                if (c == b) my_rank = __popcll(mask & lanemask_lt());
            }
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
        if (t < SYM_CLASSES) {
            u32 add = 0;
            for (int w = 0; w < kChunk / 64; ++w) add += s_wcnt[t][w];
            s_run[t] += add;
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------
// Folding the block partials.  There is no separate "stats" kernel: every block of the two
// scatter kernels folds the (<= ~1k) partials itself -- a few coalesced loads and wave
// reductions, ~2 us, all blocks in parallel -- instead of waiting for a single-workgroup
// kernel (~12 us of dependent latency plus a kernel boundary).  Block 0 publishes the totals
// to the DeviceStats block that the class kernels and the host read.
// --------------------------------------------------------------------------------
struct Fold {
    u32 prefix[kMaxClasses];  // rows of each class in the blocks before mine
    u32 total[kMaxClasses];   // rows of each class in all blocks
    u64 sum_prefix, sum_total;  // products (analysis) / nnz (numeric) before mine / overall
    u64 g_total;                // products of the NUM_G rows (numeric) / scratch entries of the SYM_NF rows
    u64 g_prefix;               //   ... in the blocks before mine
    u32 max_val;
    u32 aux_max;                // analysis: widest SYM_NF row
};

template <int THREADS, int NCLS>
__device__ __forceinline__ void fold_partials(BlockPartial* parts, u32 nb,
                                              u32 my_block, Fold* s_fold /*LDS*/, u64* s_bytes /*LDS*/,
                                              bool want_bytes)
{
    constexpr int NW = THREADS / 64;
    __shared__ u32 s_pre[NW][NCLS], s_tot[NW][NCLS], s_mx[NW], s_ax[NW];
    __shared__ u64 s_sp[NW], s_st[NW], s_by[NW][NCLS], s_g[NW], s_gp[NW];
    const PartialArrays pa(parts, nb);
    u32 pre[NCLS], tot[NCLS];
    u64 by[NCLS];
#pragma unroll
    for (int c = 0; c < NCLS; ++c) pre[c] = tot[c] = 0, by[c] = 0;
    u64 sp = 0, stt = 0, gs = 0, gp = 0;
    u32 mx = 0, ax = 0;
    for (u32 b = threadIdx.x; b < nb; b += THREADS) {  // consecutive threads, consecutive blocks: coalesced
        const bool before = b < my_block;
        const u64 gv = pa.g_ops[b];
        gs += gv;
        if (before) gp += gv;
#pragma unroll
        for (int c = 0; c < NCLS; ++c) {
            const u32 v = pa.count[c * pa.cap + b];
            tot[c] += v;
            if (before) pre[c] += v;
            if (want_bytes) by[c] += pa.bytes[c * pa.cap + b];
        }
        const u64 p = pa.products[b];
        stt += p;
        if (before) sp += p;
        mx = max(mx, pa.max_val[b]);
        ax = max(ax, pa.aux_max[b]);
    }
    const u32 wid = threadIdx.x >> 6, lane = lane_id();
#pragma unroll
    for (int c = 0; c < NCLS; ++c) {
        const u32 a = wave_reduce_add(pre[c]), t = wave_reduce_add(tot[c]);
        const u64 y = want_bytes ? wave_reduce_add(by[c]) : 0ull;
        if (lane == 0) {
            s_pre[wid][c] = a;
            s_tot[wid][c] = t;
            s_by[wid][c] = y;
        }
    }
    sp = wave_reduce_add(sp);
    stt = wave_reduce_add(stt);
    mx = wave_reduce_max(mx);
    ax = wave_reduce_max(ax);
    gs = wave_reduce_add(gs);
    gp = wave_reduce_add(gp);
    if (lane == 0) {
        s_sp[wid] = sp;
        s_st[wid] = stt;
        s_mx[wid] = mx;
        s_ax[wid] = ax;
        s_g[wid] = gs;
        s_gp[wid] = gp;
    }
    __syncthreads();
    if (threadIdx.x < kMaxClasses) {
        u32 a = 0, t = 0;
        u64 y = 0;
        if (threadIdx.x < NCLS)
            for (int w = 0; w < NW; ++w) {
                a += s_pre[w][threadIdx.x];
                t += s_tot[w][threadIdx.x];
                y += s_by[w][threadIdx.x];
            }
        s_fold->prefix[threadIdx.x] = a;
        s_fold->total[threadIdx.x] = t;
        s_bytes[threadIdx.x] = y;
    }
    if (threadIdx.x == 0) {
        u64 a = 0, t = 0, gt = 0, gpre = 0;
        u32 m = 0, axm = 0;
        for (int w = 0; w < NW; ++w) {
            a += s_sp[w];
            t += s_st[w];
            gt += s_g[w];
            gpre += s_gp[w];
            m = max(m, s_mx[w]);
            axm = max(axm, s_ax[w]);
        }
        s_fold->aux_max = axm;
        s_fold->sum_prefix = a;
        s_fold->sum_total = t;
        s_fold->g_total = gt;
        s_fold->g_prefix = gpre;
        s_fold->max_val = m;
    }
    __syncthreads();
}

// class offsets = exclusive scan of the class totals (tiny, every thread computes what it needs)
__device__ __forceinline__ u32 class_offset(const Fold& f, u32 cls)
{
    u32 run = 0;
    for (u32 c = 0; c < cls; ++c) run += f.total[c];
    return run;
}

__device__ __forceinline__ void publish_bins(BinTable& t, const Fold& f, const u64* bytes, u32 allowed_mask,
                                             DeviceStats* st)
{
    u32 run = 0;
    for (int c = 0; c < kMaxClasses; ++c) {
        t.count[c] = f.total[c];
        t.offset[c] = run;
        t.bytes[c] = bytes[c];
        run += f.total[c];
        // a replayed launch sequence only carries the kernels of `allowed_mask`
        if (f.total[c] && !((allowed_mask >> c) & 1u)) st->capacity_miss = 1;
    }
    t.offset[kMaxClasses] = run;
}

// Ordered scatter for the symbolic phase: block b re-reads the classes of its rows and writes
// each row's record at class_offset + rows-before-my-block + rank (ballots, ascending rows).
__global__ __launch_bounds__(kChunk) void sym_scatter_kernel(
    const u8* __restrict__ cls, u32 m, u32 rows_per_block, DeviceStats* __restrict__ st,
    BlockPartial* __restrict__ parts, u32 nb, const u32* __restrict__ a_ro,
    const u32* __restrict__ row_ops, const u32* __restrict__ row_col_min,
    const u32* __restrict__ row_col_max, RowRec* __restrict__ recs, ClassifyParams cp,
    u64* __restrict__ nf_off, u64 expect_nf, u32* __restrict__ pred_block_out)
{
    constexpr int NW = kChunk / 64;
    __shared__ Fold s_fold;
    __shared__ u64 s_bytes[kMaxClasses];
    __shared__ u32 s_wcnt[SYM_CLASSES][NW];
    __shared__ u32 s_run[SYM_CLASSES];
    __shared__ u32 s_nfscan[NW + 2];
    __shared__ u64 s_nfrun;
    const u32 lane = lane_id(), wid = threadIdx.x >> 6;
    const u32 row_begin = blockIdx.x * rows_per_block;
    const u32 row_end = min(m, row_begin + rows_per_block);
    // what the first chunk of rows needs is requested BEFORE the fold (a chain of dependent loads and
    // barriers): the two latencies overlap
    u32 p_c = 0xFFu, p_a0 = 0, p_a1 = 0, p_min = 0, p_max = 0, p_ops = 0;
    if (cls && row_begin + threadIdx.x < row_end) {
        const u32 row = row_begin + threadIdx.x;
        p_c = cls[row];
        p_a0 = a_ro[row];
        p_a1 = a_ro[row + 1];
        p_min = row_col_min[row];
        p_max = row_col_max[row];
        p_ops = row_ops[row];
    }
    fold_partials<kChunk, SYM_CLASSES>(parts, nb, blockIdx.x, &s_fold, s_bytes, cp.want_bytes != 0);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->sum_products = s_fold.sum_total;
        st->max_row_ops = s_fold.max_val;
        st->nf_entries = s_fold.g_total;
        st->nf_max_range = s_fold.aux_max;
        if (s_fold.aux_max == 0xFFFFFFFFu) {  // the analysis met a column id of A >= rows(B)
            st->nf_max_range = 0;
            st->a_invalid = 1;
            st->capacity_miss = 1;  // a replayed sequence stops here; the eager path returns the status
        }
        // the scratch pool of a replayed launch sequence was sized for `expect_nf` entries
        if (expect_nf != ~0ull && s_fold.g_total > expect_nf) st->capacity_miss = 1;
        publish_bins(st->sym, s_fold, s_bytes, cp.sym_allowed, st);
    }
    if (!cls) return;
    if (threadIdx.x == 0) s_nfrun = s_fold.g_prefix;
    if (threadIdx.x < SYM_CLASSES) s_run[threadIdx.x] = class_offset(s_fold, threadIdx.x) + s_fold.prefix[threadIdx.x];
    // what a replay of this call may take for granted and verify (launch.hpp, kPredBlockWords)
    if (pred_block_out && threadIdx.x < kMaxClasses) {
        const PartialArrays pa(parts, nb);
        u32* out = pred_block_out + size_t(blockIdx.x) * kPredBlockWords;
        const u32 k = threadIdx.x;
        out[k] = k < SYM_CLASSES ? class_offset(s_fold, k) + s_fold.prefix[k] : 0u;
        out[kMaxClasses + k] = pa.count[k * pa.cap + blockIdx.x];
        if (k == 0) {
            const u64 g = pa.g_ops[blockIdx.x];
            out[2 * kMaxClasses] = (u32)g;
            out[2 * kMaxClasses + 1] = (u32)(g >> 32);
        }
    }
    __syncthreads();
    for (u32 row0 = row_begin; row0 < row_end; row0 += kChunk) {
        const u32 row = row0 + threadIdx.x;
        const bool first = row0 == row_begin;
        const u32 c = first ? p_c : (row < row_end ? cls[row] : 0xFFu);
        u32 my_rank = 0;
#pragma unroll
        for (u32 b = 0; b < SYM_CLASSES; ++b) {
            const u64 mask = __ballot(c == b);
            if (lane == 0) s_wcnt[b][wid] = __popcll(mask);
            if (c == b) my_rank = __popcll(mask & lanemask_lt());
        }
        __syncthreads();
        const u32 r_min = first ? p_min : (c < SYM_CLASSES ? row_col_min[row] : 0u);
        const u32 r_max = first ? p_max : (c < SYM_CLASSES ? row_col_max[row] : 0u);
        if (c < SYM_CLASSES) {
            u32 pos = s_run[c] + my_rank;
            for (u32 w = 0; w < wid; ++w) pos += s_wcnt[c][w];
            RowRec r;
            r.row = row;
            r.a0 = first ? p_a0 : a_ro[row];
            r.a1 = first ? p_a1 : a_ro[row + 1];
            r.base = 0;
            r.cmin = r_min;
            r.cmax = r_max;
            r.ops = first ? p_ops : row_ops[row];
            r.nnz = 0;
            recs[pos] = r;
        }
        // scratch slots of the numeric-first rows (and key sets of the SYM_GH rows): exclusive prefix of their
        // column ranges (table sizes), in row order
        if (s_wcnt[SYM_NF][0] + s_wcnt[SYM_NF][1] + s_wcnt[SYM_NF][2] + s_wcnt[SYM_NF][3] + s_wcnt[SYM_GH][0] +
                s_wcnt[SYM_GH][1] + s_wcnt[SYM_GH][2] + s_wcnt[SYM_GH][3] != 0) {  // uniform
            const u32 row_ops_v = first ? p_ops : ((c == SYM_GH || c == SYM_NF) ? row_ops[row] : 0u);
            const u32 ub = c == SYM_NF ? nf_slot_entries(r_min, r_max, row_ops_v) : (c == SYM_GH ? gh_table_slots(row_ops_v) : 0u);
            u32 chunk_total;
            const u32 excl = block_exclusive_scan<kChunk>(ub, s_nfscan, &chunk_total);
            if (c == SYM_NF || c == SYM_GH) nf_off[row] = s_nfrun + excl;
            __syncthreads();
            if (threadIdx.x == 0) s_nfrun += chunk_total;
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
//   num_apply_kernel : folds the tile partials (tile offset = nnz before my tile, class bases),
//                      row_offsets = tile offset + local scan (in place) and the RowRec of
//                      every row at its class position (ascending rows inside a class)
// Traffic: 8(m+1) B for the scan itself + 4m re-read of the counts + 33 m for
// classes and records.
// --------------------------------------------------------------------------------
constexpr int kScanThreads = 256;
// rows per thread: chosen by the host so that the tile count stays <= ~4096 (every block of
// num_apply_kernel folds all tile partials) while small inputs still get >= ~300 blocks.  32 rows per thread
// are a last resort: a thread's rows are contiguous, so every load instruction of a wave touches 64 cache
// lines, and at 32 x 128 B per thread and array the L1 no longer holds them between the 32 loads (8.4 M rows:
// 0.79 + 0.69 ms for the two kernels against 0.12 + 0.25 ms at 8 rows per thread).
static inline int scan_items(u32 m) { return m <= (1u << 19) ? 2 : (m <= (1u << 23) ? 8 : 32); }
// One 16-bit counter per numeric class, four per u64: a count never exceeds the rows of a block
// (256 threads x 32 items = 8192), and sums of whole structs are plain u64 additions.
struct PackedCounts {
    u64 a = 0, b = 0, c = 0, d = 0;
    __device__ __forceinline__ void add(u32 cls)
    {
        const u64 one = 1ull << (16 * (cls & 3u));
        if (cls < 4) a += one; else if (cls < 8) b += one; else if (cls < 12) c += one; else d += one;
    }
    __device__ __forceinline__ u32 get(u32 cls) const
    {
        const u64 w = cls < 4 ? a : (cls < 8 ? b : (cls < 12 ? c : d));
        return (u32)(w >> (16 * (cls & 3u))) & 0xFFFFu;
    }
    __device__ __forceinline__ PackedCounts& operator+=(const PackedCounts& o)
    {
        a += o.a;
        b += o.b;
        c += o.c;
        d += o.d;
        return *this;
    }
};
static_assert(kMaxClasses <= 16, "PackedCounts holds 16 classes");

template <int ITEMS>
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
    __shared__ u64 s_gops[NW];
    if (threadIdx.x < kMaxClasses) s_bytes[threadIdx.x] = 0;
    __syncthreads();
    const u64 base = u64(blockIdx.x) * (kScanThreads * ITEMS) + u64(threadIdx.x) * ITEMS;
    u64 tsum = 0, g_ops = 0;
    PackedCounts packed;
    u32 my_max = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const u64 row = base + i;
        if (row < m) {
            const u32 c = counts[row];
            tsum += c;
            my_max = max(my_max, c);
            if (num_cls) {
                const u32 len_a = a_ro[row + 1] - a_ro[row];
                const u8 cls = classify_numeric(len_a, row_ops[row], c, row_col_min[row], row_col_max[row], cp);
                num_cls[row] = cls;
                if (cls == NUM_G) g_ops += row_ops[row];
                if (cls != NUM_NONE) {
                    packed.add(cls);
                    if (cp.want_bytes)
                        atomicAdd(&s_bytes[cls], numeric_row_bytes(len_a, row_ops[row], c, vsize));
                }
            }
        }
    }
    tsum = wave_reduce_add(tsum);
    packed.a = wave_reduce_add(packed.a);
    packed.b = wave_reduce_add(packed.b);
    packed.c = wave_reduce_add(packed.c);
    packed.d = wave_reduce_add(packed.d);
    my_max = wave_reduce_max(my_max);
    g_ops = wave_reduce_add(g_ops);
    const u32 wid = threadIdx.x >> 6;
    if (lane_id() == 0) {
        s_sum[wid] = tsum;
        s_max[wid] = my_max;
        s_gops[wid] = g_ops;
#pragma unroll
        for (int k = 0; k < kMaxClasses; ++k)
            s_hist[wid][k] = packed.get(k);
    }
    __syncthreads();
    const PartialArrays pa(partials, gridDim.x);
    if (threadIdx.x < kMaxClasses) {
        u32 h = 0;
        for (int w = 0; w < NW; ++w) h += s_hist[w][threadIdx.x];
        pa.count[threadIdx.x * pa.cap + blockIdx.x] = h;
        if (cp.want_bytes) pa.bytes[threadIdx.x * pa.cap + blockIdx.x] = s_bytes[threadIdx.x];
    }
    if (threadIdx.x == 0) {
        u64 s = 0, gsum = 0;
        u32 mxv = 0;
        for (int w = 0; w < NW; ++w) {
            s += s_sum[w];
            gsum += s_gops[w];
            mxv = max(mxv, s_max[w]);
        }
        pa.products[blockIdx.x] = s;  // numeric phase: the tile's nnz sum
        pa.max_val[blockIdx.x] = mxv;
        pa.aux_max[blockIdx.x] = 0;
        pa.g_ops[blockIdx.x] = gsum;
    }
}

template <int ITEMS>
__global__ __launch_bounds__(kScanThreads) void num_apply_kernel(
    const u32* counts, u32* offsets_out /* may alias counts */, u32 m, DeviceStats* __restrict__ st,
    BlockPartial* __restrict__ parts, u32 nb, const u8* __restrict__ num_cls,
    const u32* __restrict__ a_ro, const u32* __restrict__ row_ops,
    const u32* __restrict__ row_col_min, const u32* __restrict__ row_col_max,
    RowRec* __restrict__ recs, ClassifyParams cp, u64 exact_nnz, u64 expect_g, u32 expect_g_rows,
    DeviceStats* __restrict__ host_mirror, const u32* __restrict__ pred_off, u32* __restrict__ pred_off_out,
    u32* __restrict__ pred_tile_out, bool pred_fold_esc, u32* __restrict__ dev_ticket, u32* __restrict__ host_ticket)
{
    constexpr int NW = kScanThreads / 64;
    __shared__ Fold s_fold;
    __shared__ u64 s_bytes[kMaxClasses];
    __shared__ u32 s_scan[NW + 1];
    __shared__ u64 s_wave[NW][4];
    const u32 lane = lane_id(), wid = threadIdx.x >> 6;
    // my rows' counts and classes are requested BEFORE the fold (dependent loads + barriers)
    const u64 base = u64(blockIdx.x) * (kScanThreads * ITEMS) + u64(threadIdx.x) * ITEMS;
    u32 c[ITEMS];
    u8 cls[ITEMS];
    u32 tsum = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        c[i] = (base + i) < m ? counts[base + i] : 0;
        cls[i] = (num_cls && (base + i) < m) ? num_cls[base + i] : (u8)NUM_NONE;
    }
    fold_partials<kScanThreads, NUM_CLASSES>(parts, nb, blockIdx.x, &s_fold, s_bytes, cp.want_bytes != 0);
    const u64 nnz_c = s_fold.sum_total;
    // C.row_offsets of a replayed sequence is the caller's buffer: it is rewritten only when this call will
    // complete -- every block sees the flags earlier kernels raised and evaluates this kernel's own checks itself
    const bool miss = st->capacity_miss || st->b_invalid || st->a_invalid || nnz_c > 0xFFFFFFFFull ||
                      (exact_nnz != ~0ull && nnz_c != exact_nnz) || (expect_g != ~0ull && s_fold.g_total != expect_g) ||
                      (expect_g_rows != ~0u && s_fold.total[NUM_G] != expect_g_rows);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->nnz_c = nnz_c;
        st->max_row_nnz_c = s_fold.max_val;
        if (nnz_c > 0xFFFFFFFFull) st->nnz_overflow = 1;
        // the C buffers of a replayed launch sequence were allocated for exactly `exact_nnz`
        if (exact_nnz != ~0ull && nnz_c != exact_nnz) st->capacity_miss = 1;
        // ... and so was the spill pool of the NUM_G rows
        st->g_products = s_fold.g_total;
        if (expect_g != ~0ull && s_fold.g_total != expect_g) st->capacity_miss = 1;
        // ... and its per-row plan / bucket arrays for exactly that many NUM_G rows
        if (expect_g_rows != ~0u && s_fold.total[NUM_G] != expect_g_rows) st->capacity_miss = 1;
        publish_bins(st->num, s_fold, s_bytes, num_cls ? cp.num_allowed : 0xFFFFFFFFu, st);
    }
    // what a replay of this call may take for granted and verify (launch.hpp, kPredTileWords): where my tile's rows
    // of every class go, how many there are -- in the shape the REPLAY classifies (its register-class rows are
    // finished in the symbolic phase and count as rows already in place: pred_fold_esc)
    if (pred_tile_out && threadIdx.x < kMaxClasses) {
        const PartialArrays pa(parts, nb);
        auto shaped = [&](auto&& get, u32 k) -> u32 {
            if (!pred_fold_esc) return get(k);
            if (kNumEscMask >> k & 1u) return 0u;
            if (k != NUM_NFCOPY) return get(k);
            u32 sum = get(NUM_NFCOPY);
            for (u32 q = 0; q < kMaxClasses; ++q)
                if (kNumEscMask >> q & 1u) sum += get(q);
            return sum;
        };
        const u32 k = threadIdx.x;
        u32 pos = shaped([&](u32 q) { return s_fold.prefix[q]; }, k);
        for (u32 q = 0; q < k; ++q) pos += shaped([&](u32 r) { return s_fold.total[r]; }, q);
        u32* out = pred_tile_out + size_t(blockIdx.x) * kPredTileWords;
        out[k] = pos;
        out[kMaxClasses + k] = shaped([&](u32 q) { return pa.count[q * pa.cap + blockIdx.x]; }, k);
        if (k == 0) {
            const u64 g = pa.g_ops[blockIdx.x];
            out[2 * kMaxClasses] = (u32)g;
            out[2 * kMaxClasses + 1] = (u32)(g >> 32);
        }
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) tsum += c[i];
    u32 total;
    const u32 excl = block_exclusive_scan<kScanThreads>(tsum, s_scan, &total);
    // EAGER call: everything the host needs to allocate C and size the numeric launches is final once block 0 has
    // folded -- nothing a later block of this kernel does changes the statistics block.  The first wave of block 0
    // mirrors it into pinned host memory and stores the call's ticket NOW (behind the barrier of the scan above: the
    // fields thread 0 has just written are visible to the wave), instead of a done_kernel behind this kernel: the host
    // has the numeric launches queued by the time the last block is through (-10 us per eager multiply).
    if (host_mirror && blockIdx.x == 0 && wid == 0) {
        const u64* src = reinterpret_cast<const u64*>(st);
        u64* dst = reinterpret_cast<u64*>(host_mirror);
        for (u32 i = lane; i < sizeof(DeviceStats) / 8; i += 64)
            __hip_atomic_store(&dst[i], src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: the mirror is written before the ticket
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            const u32 t = *dev_ticket + 1u;
            *dev_ticket = t;
            __hip_atomic_store(host_ticket, t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    u32 run = (u32)s_fold.sum_prefix + excl;
    u32 off[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        off[i] = run;
        if (base + i < m && !miss) {
            offsets_out[base + i] = run;
            if (pred_off_out) pred_off_out[base + i] = run;  // the config's own copy: C.row_offsets is the caller's
        }
        // rows the numeric-first kernel has already placed by the previous call's offsets: the fresh ones must agree
        // (checked for EVERY row: a shift anywhere before such a row moves it)
        if (pred_off && base + i < m && pred_off[base + i] != run) st->capacity_miss = 1;
        run += c[i];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0 && !miss) {
        offsets_out[m] = (u32)nnz_c;
        if (pred_off_out) pred_off_out[m] = (u32)nnz_c;
    }
    if (!num_cls) return;

    // class of my rows, packed per-thread histogram, exclusive scan over the threads
    PackedCounts mine;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
        if (cls[i] != NUM_NONE) mine.add(cls[i]);
    PackedCounts incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u64 ta = __shfl_up(incl.a, o, 64), tb = __shfl_up(incl.b, o, 64), tc = __shfl_up(incl.c, o, 64),
                  td = __shfl_up(incl.d, o, 64);
        if (lane >= (u32)o) {
            incl.a += ta;
            incl.b += tb;
            incl.c += tc;
            incl.d += td;
        }
    }
    if (lane == 63) {
        s_wave[wid][0] = incl.a;
        s_wave[wid][1] = incl.b;
        s_wave[wid][2] = incl.c;
        s_wave[wid][3] = incl.d;
    }
    __syncthreads();
    PackedCounts before;  // rows of each class in the threads before mine (exclusive)
    before.a = incl.a - mine.a;
    before.b = incl.b - mine.b;
    before.c = incl.c - mine.c;
    before.d = incl.d - mine.d;
    for (u32 w = 0; w < wid; ++w) {
        before.a += s_wave[w][0];
        before.b += s_wave[w][1];
        before.c += s_wave[w][2];
        before.d += s_wave[w][3];
    }
    PackedCounts used;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const u64 row = base + i;
        if (cls[i] == NUM_NONE) continue;
        if (cls[i] == NUM_NFCOPY && pred_off) continue;  // already in place: no launch reads that list
        const u32 k = cls[i];
        const u32 pos = class_offset(s_fold, k) + s_fold.prefix[k] + before.get(k) + used.get(k);
        used.add(k);
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

// The scan of a replayed sequence with every row offset and every tile table predicted (launch.hpp,
// launch_scan_predicted): no count kernel, no fold.
template <int ITEMS>
__global__ __launch_bounds__(kScanThreads) void num_apply_pred_kernel(
    const u32* counts, u32* offsets_out /* may alias counts */, u32 m, DeviceStats* __restrict__ st,
    const u32* __restrict__ a_ro, const u32* __restrict__ row_ops, const u32* __restrict__ row_col_min,
    const u32* __restrict__ row_col_max, RowRec* __restrict__ recs, ClassifyParams cp,
    const u32* __restrict__ pred_off, const u32* __restrict__ pred_tile, const DeviceStats* __restrict__ pred_stats,
    BlockPartial* __restrict__ an_parts, u32 an_blocks, bool totals_from_pred)
{
    constexpr int NW = kScanThreads / 64;
    __shared__ u32 s_scan[NW + 1];
    __shared__ u64 s_wave[NW][4];
    __shared__ u64 s_g[NW];
    __shared__ u32 s_pos[kMaxClasses];
    __shared__ u32 s_bad;
    const u32 lane = lane_id(), wid = threadIdx.x >> 6;
    const u64 tile0 = u64(blockIdx.x) * (kScanThreads * ITEMS);
    const u64 base = tile0 + u64(threadIdx.x) * ITEMS;
    const u32* tab = pred_tile + size_t(blockIdx.x) * kPredTileWords;
    if (threadIdx.x == 0) s_bad = 0;
    if (threadIdx.x < kMaxClasses) s_pos[threadIdx.x] = tab[threadIdx.x];
    u32 c[ITEMS], po[ITEMS + 1];
    u8 cls[ITEMS];
    u32 tsum = 0;
    u64 g_ops = 0;
    PackedCounts mine;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const u64 row = base + i;
        c[i] = row < m ? counts[row] : 0;
        po[i] = row < m ? pred_off[row] : 0;
        cls[i] = NUM_NONE;
        if (row < m) {
            const u32 ops = row_ops[row];
            cls[i] = classify_numeric(a_ro[row + 1] - a_ro[row], ops, c[i], row_col_min[row], row_col_max[row], cp);
            if (cls[i] == NUM_G) g_ops += ops;
            if (cls[i] != NUM_NONE) mine.add(cls[i]);
        }
        tsum += c[i];
    }
    po[ITEMS] = base < m ? pred_off[base + ITEMS <= m ? base + ITEMS : m] : 0;  // where my last row ends
    const u32 tile_base = pred_off[tile0];  // (tile0 < m: the grid has no empty tile)
    // the statistics of the predicted call: what the numeric kernels (class lists) and the host read
    if (blockIdx.x == 0) {
        constexpr u32 kWords = sizeof(BinTable) / 4;
        const u32* src = reinterpret_cast<const u32*>(&pred_stats->num);
        u32* dst = reinterpret_cast<u32*>(&st->num);
        for (u32 i = threadIdx.x; i < kWords; i += kScanThreads) dst[i] = src[i];
        if (threadIdx.x == 0) {
            st->nnz_c = pred_stats->nnz_c;
            st->max_row_nnz_c = pred_stats->max_row_nnz_c;
            st->g_products = pred_stats->g_products;
        }
        // the analysis of this sequence only VERIFIES, beside it (every row's products compared with the previous
        // identical call's): the totals are that call's
        if (totals_from_pred && threadIdx.x == 0) {
            st->sum_products = pred_stats->sum_products;
            st->max_row_ops = pred_stats->max_row_ops;
            st->nf_max_range = pred_stats->nf_max_range;
        }
        // the sequence had no scatter kernel (predicted symbolic binning): the totals of the analysis are folded here
        if (an_parts) {
            __shared__ u64 s_ap[NW];
            __shared__ u32 s_am[NW], s_ar[NW];
            const PartialArrays pa(an_parts, an_blocks);
            u64 p = 0;
            u32 mx = 0, ar = 0;
            for (u32 b = threadIdx.x; b < an_blocks; b += kScanThreads) {
                p += pa.products[b];
                mx = max(mx, pa.max_val[b]);
                ar = max(ar, pa.aux_max[b]);
            }
            p = wave_reduce_add(p);
            mx = wave_reduce_max(mx);
            ar = wave_reduce_max(ar);
            if (lane == 0) {
                s_ap[wid] = p;
                s_am[wid] = mx;
                s_ar[wid] = ar;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                u64 tp = 0;
                u32 tm = 0, tr = 0;
                for (int w = 0; w < NW; ++w) {
                    tp += s_ap[w];
                    tm = max(tm, s_am[w]);
                    tr = max(tr, s_ar[w]);
                }
                st->sum_products = tp;
                st->max_row_ops = tm;
                st->nf_max_range = tr == 0xFFFFFFFFu ? 0u : tr;
            }
        }
    }
    u32 total;
    const u32 excl = block_exclusive_scan<kScanThreads>(tsum, s_scan, &total);
    u32 run = tile_base + excl;
    bool bad = false;
    u32 off[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        off[i] = run;
        if (base + i < m && po[i] != run) bad = true;
        run += c[i];
    }
    if (base < m && po[ITEMS] != run) bad = true;  // (rows past m add nothing: the last row of C ends at pred_off[m])
    // class histogram of the tile and the products of its NUM_G rows against the prediction
    PackedCounts incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u64 ta = __shfl_up(incl.a, o, 64), tb = __shfl_up(incl.b, o, 64), tc = __shfl_up(incl.c, o, 64),
                  td = __shfl_up(incl.d, o, 64);
        if (lane >= (u32)o) {
            incl.a += ta;
            incl.b += tb;
            incl.c += tc;
            incl.d += td;
        }
    }
    g_ops = wave_reduce_add(g_ops);
    if (lane == 63) {
        s_wave[wid][0] = incl.a;
        s_wave[wid][1] = incl.b;
        s_wave[wid][2] = incl.c;
        s_wave[wid][3] = incl.d;
    }
    if (lane == 0) s_g[wid] = g_ops;
    if (__ballot(bad) != 0 && lane == 0) s_bad = 1;
    __syncthreads();
    if (threadIdx.x < kMaxClasses) {
        PackedCounts all;
        for (int w = 0; w < NW; ++w) {
            all.a += s_wave[w][0];
            all.b += s_wave[w][1];
            all.c += s_wave[w][2];
            all.d += s_wave[w][3];
        }
        if (all.get(threadIdx.x) != tab[kMaxClasses + threadIdx.x]) s_bad = 1;
        if (threadIdx.x == 0) {
            u64 g = 0;
            for (int w = 0; w < NW; ++w) g += s_g[w];
            if (g != ((u64(tab[2 * kMaxClasses + 1]) << 32) | tab[2 * kMaxClasses])) s_bad = 1;
        }
    }
    __syncthreads();
    if (s_bad) {  // (uniform) this tile is not what it was: nothing of it is written, the eager path re-runs the call
        if (threadIdx.x == 0) st->capacity_miss = 1;
        return;
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
        if (base + i < m) offsets_out[base + i] = off[i];  // (= the prediction = what the previous call left there)
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) offsets_out[m] = pred_off[m];
    PackedCounts before;  // rows of each class in the threads before mine (exclusive)
    before.a = incl.a - mine.a;
    before.b = incl.b - mine.b;
    before.c = incl.c - mine.c;
    before.d = incl.d - mine.d;
    for (u32 w = 0; w < wid; ++w) {
        before.a += s_wave[w][0];
        before.b += s_wave[w][1];
        before.c += s_wave[w][2];
        before.d += s_wave[w][3];
    }
    PackedCounts used;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const u64 row = base + i;
        if (cls[i] == NUM_NONE) continue;
        if (cls[i] == NUM_NFCOPY) continue;  // already in place (direct placement): no launch reads that list
        const u32 k = cls[i];
        const u32 pos = s_pos[k] + before.get(k) + used.get(k);
        used.add(k);
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
// Input precondition (undocumented upstream): the column ids of every row of B are
// strictly ascending (the min/max column range of the analysis, the scaled-copy rows and the bitmap
// sorts rely on it; the reference's loader guarantees it and silently computes garbage otherwise).
// One coalesced pass over B.col_ids, eager path only (a replayed sequence runs on unchanged inputs): validate_b_slice,
// run by extra workgroups of the analysis launch (above).
// --------------------------------------------------------------------------------
// Last node of a replayed launch sequence: a ticket in pinned host memory the host spins on (a blocking
// stream synchronisation costs ~10-20 us of wake-up latency: a tenth of a 200 us multiply).
// The same wave first copies the (final) statistics block into the pinned mirror with system-scope stores, so
// the host may read it as soon as it sees the ticket: nothing relies on an earlier kernel's plain stores to
// host memory being visible by then.
__global__ __launch_bounds__(64) void done_kernel(u32* __restrict__ dev_ticket, u32* __restrict__ host_ticket,
                                                  const DeviceStats* __restrict__ st, DeviceStats* __restrict__ host_mirror)
{
    const u64* src = reinterpret_cast<const u64*>(st);
    u64* dst = reinterpret_cast<u64*>(host_mirror);
    for (u32 i = threadIdx.x; i < sizeof(DeviceStats) / 8; i += 64)
        __hip_atomic_store(&dst[i], src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: the mirror is written before the ticket
    __builtin_amdgcn_wave_barrier();
    if (threadIdx.x == 0) {
        const u32 t = *dev_ticket + 1u;
        *dev_ticket = t;
        __hip_atomic_store(host_ticket, t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// The input check of B (validate_b_slice) as a kernel of its own, for the verifier's stream: an eager call launches it
// first and looks at the verdict -- bit 2 of the pinned word -- when it reads back the statistics of its scan, long after
// this kernel is through; nothing of C is written before that (riding in the analysis launch it cost that launch 13 us).
__global__ __launch_bounds__(256) void validate_b_kernel(const u32* __restrict__ b_ro, const u32* __restrict__ b_col, u32 b_rows,
                                                          u32 b_cols, u32* __restrict__ verdict)
{
    // four consecutive entries (and the one behind them) per lane and step, all five loads in flight together: the
    // kernel runs beside the analysis of the call and should be gone before it competes with anything else
    constexpr u32 E = 4;
    const u32 e_first = b_ro[0], e_last = b_ro[b_rows];
    bool bad = e_last < e_first;
    const u64 n = e_last > e_first ? u64(e_last - e_first) : 0;
    for (u64 i = (u64(blockIdx.x) * 256 + threadIdx.x) * E; i < n; i += u64(gridDim.x) * 256 * E) {
        const u32 e = e_first + (u32)i;
        u32 c[E + 1];
#pragma unroll
        for (u32 k = 0; k <= E; ++k) c[k] = i + k < n ? b_col[e + k] : 0xFFFFFFFFu;
#pragma unroll
        for (u32 k = 0; k < E; ++k) {
            if (i + k >= n) continue;
            if (c[k] >= b_cols) bad = true;
            if (i + k + 1 < n && c[k + 1] <= c[k]) {  // not ascending: fine only where a row starts (almost never looked up)
                const u32 at = e + k + 1;
                u32 lo = 0, hi = b_rows;  // first row whose offset is >= at
                while (lo < hi) {
                    const u32 mid = lo + ((hi - lo) >> 1);
                    if (b_ro[mid] < at) lo = mid + 1; else hi = mid;
                }
                if (b_ro[lo] != at) bad = true;
            }
        }
    }
    if (__ballot(bad) != 0 && lane_id() == 0) __hip_atomic_fetch_or(verdict, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_validate_b(hipStream_t s, const u32* b_ro, const u32* b_col, u32 b_rows, u32 b_cols, u64 b_nnz, u32* verdict)
{
    if (b_rows == 0) return;
    const u32 want = cdiv(b_nnz ? b_nnz : 1, 256 * 4);
    hipLaunchKernelGGL(validate_b_kernel, dim3(want > 2048u ? 2048u : want), dim3(256), 0, s, b_ro, b_col, b_rows, b_cols, verdict);
}

// ... and the ticket of the verifier's stream (launch_verifier): the kernel boundary in front of it orders the verifier's
// verdict (system-scope atomics on pinned memory) before the ticket
__global__ __launch_bounds__(64) void ticket_kernel(u32* __restrict__ dev_ticket, u32* __restrict__ host_ticket)
{
    if (threadIdx.x == 0) {
        const u32 t = *dev_ticket + 1u;
        *dev_ticket = t;
        __hip_atomic_store(host_ticket, t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// ---- the inputs of the analysis, kept and compared (pipeline.hip: launch_verifier) ------------------------------------
// Everything the analysis leaves in the arena is a function of A.row_offsets, A.col_ids, B.row_offsets and the first and
// last column id of every row of B.  A replayed sequence whose analysis only VERIFIES (ReplayPlan::overlap) therefore does
// not have to recompute that function and compare its results entry by entry -- a chain of gathers A.col -> B.rowptr ->
// B.col per entry of A, 18 ms beside the numeric launch of the nlpkkt stand-in (4.5 ms alone) and a tenth of that launch's
// bandwidth: it compares the INPUTS with the copy the last writing analysis went with -- four streams, no gather but the
// two column ids per row of B.  b_snap: B.row_offsets [k + 1] | first, last column id per row [2 k].
__global__ __launch_bounds__(256) void snapshot_inputs_kernel(const u32* __restrict__ a_ro, const u32* __restrict__ a_col,
                                                               u32* __restrict__ a_col_copy, u64 nnz_a,
                                                               const u32* __restrict__ b_ro, const u32* __restrict__ b_col,
                                                               u32 b_rows, u32* __restrict__ b_snap)
{
    const u32 e_base = a_ro[0];  // (A may be a row-range view with absolute offsets)
    const u64 tid = u64(blockIdx.x) * 256 + threadIdx.x, nthreads = u64(gridDim.x) * 256;
    for (u64 i = tid; i < nnz_a; i += nthreads) a_col_copy[i] = a_col[e_base + i];
    for (u64 r = tid; r <= b_rows; r += nthreads) {
        const u32 lo = b_ro[r];
        b_snap[r] = lo;
        if (r < b_rows) {
            const u32 hi = b_ro[r + 1];
            b_snap[size_t(b_rows) + 1 + 2 * r] = hi > lo ? b_col[lo] : 0xFFFFFFFFu;
            b_snap[size_t(b_rows) + 2 + 2 * r] = hi > lo ? b_col[hi - 1] : 0u;
        }
    }
}
__global__ __launch_bounds__(256) void verify_inputs_kernel(const u32* __restrict__ a_ro, const u32* __restrict__ a_ro_copy,
                                                             u32 m, const u32* __restrict__ a_col,
                                                             const u32* __restrict__ a_col_copy, u64 nnz_a,
                                                             const u32* __restrict__ b_ro, const u32* __restrict__ b_col,
                                                             u32 b_rows, const u32* __restrict__ b_snap, u32* __restrict__ verdict)
{
    const u32 e_base = a_ro[0];
    const u64 tid = u64(blockIdx.x) * 256 + threadIdx.x, nthreads = u64(gridDim.x) * 256;
    bool bad = e_base != a_ro_copy[0];
    for (u64 i = tid; i <= m; i += nthreads) bad |= a_ro[i] != a_ro_copy[i];
    if (!bad) {  // (same first entry: the copies line up)
        constexpr u32 U = 4;  // entries per thread and step, all loads in flight
        for (u64 i0 = tid * U; i0 < nnz_a; i0 += nthreads * U) {
            u32 x[U], y[U];
#pragma unroll
            for (u32 u = 0; u < U; ++u) {
                const u64 i = i0 + u;
                x[u] = i < nnz_a ? a_col[e_base + i] : 0u;
                y[u] = i < nnz_a ? a_col_copy[i] : 0u;
            }
#pragma unroll
            for (u32 u = 0; u < U; ++u) bad |= x[u] != y[u];
        }
    }
    for (u64 r = tid; r <= b_rows; r += nthreads) {
        const u32 lo = b_ro[r];
        if (lo != b_snap[r]) {
            bad = true;
            continue;
        }
        if (r < b_rows) {
            const u32 hi = b_ro[r + 1];
            if (hi != b_snap[r + 1]) {  // (the offsets are the ones the snapshot was taken with: inside B's arrays)
                bad = true;
                continue;
            }
            const u32 first = hi > lo ? b_col[lo] : 0xFFFFFFFFu, last = hi > lo ? b_col[hi - 1] : 0u;
            bad |= first != b_snap[size_t(b_rows) + 1 + 2 * r] || last != b_snap[size_t(b_rows) + 2 + 2 * r];
        }
    }
    if (__ballot(bad) != 0 && lane_id() == 0) __hip_atomic_fetch_or(verdict, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
static u32 input_blocks(u64 nnz_a, u32 b_rows)
{
    const u64 want = cdiv(std::max<u64>(nnz_a, b_rows) + 1, 256 * 8);
    return (u32)std::min<u64>(std::max<u64>(want, 1), 4096);
}
void launch_snapshot_inputs(hipStream_t s, const u32* a_ro, const u32* a_col, u32* a_col_copy, u64 nnz_a, const u32* b_ro,
                            const u32* b_col, u32 b_rows, u32* b_snap)
{
    hipLaunchKernelGGL(snapshot_inputs_kernel, dim3(input_blocks(nnz_a, b_rows)), dim3(256), 0, s, a_ro, a_col, a_col_copy,
                       nnz_a, b_ro, b_col, b_rows, b_snap);
}
void launch_verify_inputs(hipStream_t s, const u32* a_ro, const u32* a_ro_copy, u32 m, const u32* a_col,
                          const u32* a_col_copy, u64 nnz_a, const u32* b_ro, const u32* b_col, u32 b_rows, const u32* b_snap,
                          u32* verdict)
{
    hipLaunchKernelGGL(verify_inputs_kernel, dim3(input_blocks(nnz_a, b_rows)), dim3(256), 0, s, a_ro, a_ro_copy, m, a_col,
                       a_col_copy, nnz_a, b_ro, b_col, b_rows, b_snap, verdict);
}

// One wave that does nothing for `us` microseconds (constant 100 MHz counter): the profiled pre-pass of a long sequence
// puts it in front of the verifier, which the host launches ~30 us behind a replayed graph (pipeline.hip, gate_verifier)
__global__ __launch_bounds__(64) void delay_kernel(u32 ticks)
{
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
void launch_delay(hipStream_t s, u32 us) { hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, s, us * 100u); }

void launch_ticket(hipStream_t s, u32* dev_ticket, u32* host_ticket)
{
    hipLaunchKernelGGL(ticket_kernel, dim3(1), dim3(64), 0, s, dev_ticket, host_ticket);
}
void launch_done(hipStream_t s, u32* dev_ticket, u32* host_ticket, const DeviceStats* st, DeviceStats* host_mirror)
{
    hipLaunchKernelGGL(done_kernel, dim3(1), dim3(64), 0, s, dev_ticket, host_ticket, st, host_mirror);
}

// --------------------------------------------------------------------------------
// host launchers
// --------------------------------------------------------------------------------
static u32 g_an_wide_rows = 16;
void set_analysis_wide_rows(u32 avg_len) { g_an_wide_rows = avg_len; }

u32 analysis_blocks(u32 m)
{
    u32 r, b;
    row_chunking(m, &r, &b);
    return b;
}
u32 scan_tiles(u32 m) { return cdiv(m ? m : 1, kScanThreads * scan_items(m)); }

void launch_analysis(hipStream_t s, const u32* a_ro, const u32* a_col, const u32* b_ro,
                     const u32* b_col, u32 m, u64 nnz_a, u32* row_ops, u32* row_max_ops,
                     u32* row_col_min, u32* row_col_max, u8* sym_cls, u32* counts,
                     BlockPartial* partials, RowRec* recs, DeviceStats* st, const ClassifyParams& cp,
                     uint2* b_sl, hipEvent_t between, u64* nf_off, u64 expect_nf, u32 b_rows, u32* pred_block_out,
                     const u32* pred_block, const DeviceStats* pred_stats, u32 b_cols, u64 b_nnz, u32 validate_epoch,
                     u32* a_ro_copy, u32* verdict)
{
    const bool verify = verdict != nullptr;
    // eager path with the input check on: workgroups behind the analysis grid walk B's entries (validate_b_slice)
    auto vblocks = [&](u32 threads) -> u32 {
        if (!validate_epoch || b_rows == 0 || b_rows == ~0u) return 0u;
        const u32 want = cdiv(b_nnz ? b_nnz : 1, threads * 4);
        return want > 2048u ? 2048u : want;
    };
    u32 rows_per_block, blocks;
    row_chunking(m, &rows_per_block, &blocks);
    // 64 rows per wave for short rows (g_an_wide_rows: average entries per row up to which; 0 = never)
    const bool wide = g_an_wide_rows && m && nnz_a / m <= g_an_wide_rows;
    if (verify) {  // replayed sequence with the analysis beside it: compare, write nothing (analysis_kernel, VERIFY)
        if (wide)
            hipLaunchKernelGGL((analysis_kernel<4, 64, true>), dim3(blocks), dim3(256), 0, s, a_ro, a_col, b_ro, b_col, m,
                               rows_per_block, row_ops, row_max_ops, row_col_min, row_col_max, sym_cls, counts,
                               partials, cp, b_sl, st, b_rows, (const u32*)nullptr, (const DeviceStats*)nullptr,
                               (RowRec*)nullptr, blocks, 0u, 0u, a_ro_copy, verdict);
        else
            hipLaunchKernelGGL((analysis_kernel<8, 32, true>), dim3(blocks), dim3(512), 0, s, a_ro, a_col, b_ro, b_col, m,
                               rows_per_block, row_ops, row_max_ops, row_col_min, row_col_max, sym_cls, counts,
                               partials, cp, b_sl, st, b_rows, (const u32*)nullptr, (const DeviceStats*)nullptr,
                               (RowRec*)nullptr, blocks, 0u, 0u, a_ro_copy, verdict);
        return;
    }
    if (pred_block && sym_cls) {  // replayed sequence, symbolic binning predicted: no scatter kernel
        if (wide)
            hipLaunchKernelGGL((analysis_kernel<4, 64>), dim3(blocks), dim3(256), 0, s, a_ro, a_col, b_ro, b_col, m,
                               rows_per_block, row_ops, row_max_ops, row_col_min, row_col_max, sym_cls, counts,
                               partials, cp, b_sl, st, b_rows, pred_block, pred_stats, recs, blocks, 0u, 0u, a_ro_copy, (u32*)nullptr);
        else
            hipLaunchKernelGGL((analysis_kernel<8, 32>), dim3(blocks), dim3(512), 0, s, a_ro, a_col, b_ro, b_col, m,
                               rows_per_block, row_ops, row_max_ops, row_col_min, row_col_max, sym_cls, counts,
                               partials, cp, b_sl, st, b_rows, pred_block, pred_stats, recs, blocks, 0u, 0u, a_ro_copy, (u32*)nullptr);
        if (between) (void)hipEventRecord(between, s);
        return;
    }
    if (wide)
        hipLaunchKernelGGL((analysis_kernel<4, 64>), dim3(blocks + vblocks(256)), dim3(256), 0, s, a_ro, a_col, b_ro, b_col, m,
                           rows_per_block, row_ops, row_max_ops, row_col_min, row_col_max, sym_cls, counts,
                           partials, cp, b_sl, st, b_rows, (const u32*)nullptr, (const DeviceStats*)nullptr,
                           (RowRec*)nullptr, blocks, b_cols, validate_epoch, a_ro_copy, (u32*)nullptr);
    else
        hipLaunchKernelGGL((analysis_kernel<8, 32>), dim3(blocks + vblocks(512)), dim3(512), 0, s, a_ro, a_col, b_ro, b_col, m,
                           rows_per_block, row_ops, row_max_ops, row_col_min, row_col_max, sym_cls, counts,
                           partials, cp, b_sl, st, b_rows, (const u32*)nullptr, (const DeviceStats*)nullptr,
                           (RowRec*)nullptr, blocks, b_cols, validate_epoch, a_ro_copy, (u32*)nullptr);
    if (between) (void)hipEventRecord(between, s);  // analysis | binning (Timings::countProducts / loadBalanceCounting)
    // with sym_cls == nullptr only block 0 does anything: it folds the totals (P, max row ops)
    hipLaunchKernelGGL(sym_scatter_kernel, dim3(sym_cls ? blocks : 1), dim3(kChunk), 0, s,
                       (const u8*)sym_cls, m, rows_per_block, st, partials, blocks, a_ro,
                       (const u32*)row_ops, (const u32*)row_col_min, (const u32*)row_col_max, recs, cp, nf_off,
                       expect_nf, sym_cls ? pred_block_out : nullptr);
}

void launch_scan(hipStream_t s, const u32* counts, u32* offsets_out, u32 m, const u32* a_ro, const u32* row_ops,
                 const u32* row_col_min, const u32* row_col_max, u8* num_cls, BlockPartial* partials,
                 RowRec* recs, DeviceStats* st, const ClassifyParams& cp, u32 vsize, u64 exact_nnz,
                 DeviceStats* host_mirror, u64 expect_g, u32 expect_g_rows, const u32* pred_off, u32* pred_off_out,
                 u32* pred_tile_out, bool pred_fold_esc, u32* dev_ticket, u32* host_ticket)
{
    const u32 tiles = scan_tiles(m);
    auto go = [&](auto items) {
        constexpr int I = decltype(items)::value;
        hipLaunchKernelGGL(num_count_kernel<I>, dim3(tiles), dim3(kScanThreads), 0, s,
                           counts, m, a_ro, row_ops, row_col_min, row_col_max, num_cls,
                           partials, cp, vsize);
        hipLaunchKernelGGL(num_apply_kernel<I>, dim3(tiles), dim3(kScanThreads), 0, s, counts, offsets_out, m, st,
                           partials, tiles, (const u8*)num_cls, a_ro, row_ops,
                           row_col_min, row_col_max, recs, cp, exact_nnz, expect_g, expect_g_rows, host_mirror, pred_off,
                           pred_off_out, num_cls ? pred_tile_out : nullptr, pred_fold_esc, dev_ticket, host_ticket);
    };
    switch (scan_items(m)) {
        case 2: go(std::integral_constant<int, 2>{}); break;
        case 8: go(std::integral_constant<int, 8>{}); break;
        default: go(std::integral_constant<int, 32>{}); break;
    }
}

void launch_scan_predicted(hipStream_t s, const u32* counts, u32* offsets_out, u32 m, const u32* a_ro,
                           const u32* row_ops, const u32* row_col_min, const u32* row_col_max, RowRec* recs,
                           DeviceStats* st, const ClassifyParams& cp, const u32* pred_off, const u32* pred_tile,
                           const DeviceStats* pred_stats, BlockPartial* analysis_partials, bool totals_from_pred)
{
    const u32 tiles = scan_tiles(m);
    auto go = [&](auto items) {
        constexpr int I = decltype(items)::value;
        hipLaunchKernelGGL(num_apply_pred_kernel<I>, dim3(tiles), dim3(kScanThreads), 0, s, counts, offsets_out, m, st,
                           a_ro, row_ops, row_col_min, row_col_max, recs, cp, pred_off, pred_tile, pred_stats,
                           analysis_partials, analysis_partials ? analysis_blocks(m) : 0u, totals_from_pred);
    };
    switch (scan_items(m)) {
        case 2: go(std::integral_constant<int, 2>{}); break;
        case 8: go(std::integral_constant<int, 8>{}); break;
        default: go(std::integral_constant<int, 32>{}); break;
    }
}

}  // namespace speck
#ifdef SPECK_PHASE_CLOCKS
extern "C" int speck_debug_analysis_clocks(unsigned long long* out8)
{
    static unsigned long long all[1024 * 8];
    if (hipMemcpyFromSymbol(all, HIP_SYMBOL(speck::g_an_clk), sizeof(all)) != hipSuccess) return 3;
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    for (int b = 0; b < 1024; ++b)
        for (int i = 0; i < 8; ++i) out8[i] += all[b * 8 + i];
    for (auto& x : all) x = 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(speck::g_an_clk), all, sizeof(all)) == hipSuccess ? 0 : 3;
}
#endif
