// stages.hip -- the integer-only stages of the pipeline for gfx950:
//   * lightweight analysis      (role of readOperations, reference include/common.cuh:321-459)
//   * row -> kernel-class binning (role of the reference's load balancer,
//     include/GPU/spECK_HashLoadBalancer.cuh:265-347 + scan_largearray_kernel.cuh:182-281)
//   * exclusive scan of the per-row counts into C.row_offsets
//     (role of cub::DeviceScan::ExclusiveSum, reference source/GPU/Multiply.cu:570)
// Round 5: TWO kernels for all of it (five before: analysis, sym_scatter, num_count, num_apply + a ticket).  Each of
// the two is a single pass whose workgroups learn what the workgroups before them found through the look-back chain
// of chain.hpp -- no second kernel that folds block partials, no global atomic, no fence:
//   analysis_kernel : per row products / longest B row / column range / symbolic class; per entry of A the (start,
//                     length) of its B row; then, behind the chain, every row's 32-byte RowRec (ROW order) and its
//                     row id in the list of its symbolic class, at (rows of the class before my workgroup) + rank
//                     -- deterministic, rows ascending inside every class;  the last workgroup writes the statistics
//   scan_kernel     : per row numeric class, C.row_offsets, RowRec.base / nnz, row id in the numeric class list;
//                     the last workgroup writes the statistics and (eager call) mirrors them to the host + ticket
// Class lists hold ROW IDS in regions at FIXED places (ClassLists, launch.hpp: two classes share a region of rows(A)
// words and grow towards each other), so a producer needs only the rows BEFORE it -- never the totals -- and a class
// kernel can request its first list entries before it has seen the class table.
#include <algorithm>
#include <type_traits>

#include "chain.hpp"
#include "device_common.hpp"
#include "launch.hpp"

namespace speck {

static inline u32 cdiv(u64 a, u64 b) { return (u32)((a + b - 1) / b); }

typedef u32 RowPtrPair __attribute__((ext_vector_type(2), aligned(4)));
#ifdef SPECK_PHASE_CLOCKS
static __device__ unsigned long long g_an_clk[1024 * 16];
#define AN_BEGIN() long long an_t_ = clock64()
#define AN_MARK(i_) \
    do { \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
        const long long an_n_ = clock64(); \
        if (lane_id() == 0) atomicAdd(&g_an_clk[(blockIdx.x % 1024) * 16 + (i_)], (unsigned long long)(an_n_ - an_t_)); \
        an_t_ = an_n_; \
    } while (0)
#else
#define AN_BEGIN()
#define AN_MARK(i_)
#endif
constexpr int kChunk = 256;  // rows per pass of an analysis workgroup

// rows per block for the analysis: contiguous, multiple of kChunk, at most 1024 blocks
static inline void row_chunking(u32 m, u32* rows_per_block, u32* blocks)
{
    u32 r = cdiv(m ? m : 1, 1024);
    r = (r + kChunk - 1u) & ~u32(kChunk - 1);
    *rows_per_block = r;
    *blocks = cdiv(m ? m : 1, r);
}

// --------------------------------------------------------------------------------
// Analysis, nnz-parallel and WAVE-granular: a block owns a contiguous row range, each of its
// waves walks that range in sub-chunks of 32 / 64 rows, and inside a sub-chunk the lanes stride over
// the A ENTRIES (not the rows), so every lane has independent 3-deep load chains
// (A.col -> B.rowptr pair -> first/last B.col) in flight and long rows cost nothing extra.
// Nothing but the final histogram crosses a wave: no workgroup barrier inside the loops (with
// 256-row sub-chunks per workgroup, 75 % of the wave cycles of this kernel were barrier waits).
// Per-row results are combined with LDS atomics in the wave's own staging area.
// HBM traffic (algorithmic): 4(m+1) + 4 nnzA + 8 nnzA [B.rowptr pair] + 8 nnzA
// [first/last col of the B row] + 17 m written (+ 8 nnzA for b_start / b_len, + 36 m for records and lists).
// --------------------------------------------------------------------------------
constexpr u32 kAnRowPathMax = 8;  // rows per lane up to this many entries (sub-chunk maximum)
// Two shapes, one kChunk of rows per pass of a block either way: 8 waves x 32 rows (the default) and 4 waves x 64 rows
// for inputs of short rows -- there a lane per row is the common path, 32 rows leave half a wave idle, and at 91 VGPRs
// (5 waves per SIMD) the 8 x 32 shape needs more wave slots than the chip has for ~200 k rows: two rounds of
// workgroups, each as long as its chain of dependent gathers (scircuit / mac_econ stand-ins).
constexpr u32 kAnCoopMax = 64;        // ... at most this many per workgroup (the others stay with their wave)
constexpr u32 kAnCoopRowLen = 256, kAnCoopEntries = 2048;  // sub-chunks (32 rows) with more entries than this, one row
                                                            //   holding more than that, are walked by the whole workgroup

// VERIFY (replayed sequence with the analysis OFF the critical path): nothing is written.  Every
// quantity this kernel would produce -- the (start, length) pair of every entry, ops / longest B row / column range / class
// of every row, A's row offsets -- is recomputed from the inputs as they are NOW and compared with what the previous
// identical call left in the arena (which the symbolic, scan and numeric kernels of this sequence are reading while this
// kernel runs beside them on its own stream); any difference raises the verdict word and the eager path re-runs the call.
template <int NW, u32 R, bool VERIFY = false>
__global__ __launch_bounds__(NW * 64) void analysis_kernel(
    const u32* __restrict__ a_ro, const u32* __restrict__ a_col, const u32* __restrict__ b_ro,
    const u32* __restrict__ b_col, u32 m, u32 rows_per_block, u32* row_ops,
    u32* row_max_ops, u32* row_col_min, u32* row_col_max,
    u8* sym_cls, u32* __restrict__ counts, ClassifyParams cp, uint2* b_sl, DeviceStats* __restrict__ st, u32 b_rows,
    RowRec* __restrict__ sym_recs, u64* __restrict__ nf_off, u64 expect_nf, Chain chain,
    u32* a_ro_copy, u32* __restrict__ verdict, u64* __restrict__ bytes_acc, u64 b_nnz)
{
    SPECK_POISON();
    constexpr int kAnThreads = NW * 64;
    constexpr int U = 4;   // entries per lane and tile: 256 entries cover most 32-row sub-chunks in ONE
                           //   round of the dependent chain A.col -> B.rowptr -> B.col
    // R = rows per sub-chunk (one per lane when they are finalised): the kernel
                           //   lasts as long as its slowest wave, so the waves are kept short and many
    static_assert(NW * R == kChunk, "one pass of a block covers one kChunk of rows");
    const u32 e_base = a_ro[0];  // A may be a row-range view with absolute offsets
    // Every (start, end) pair read from B.row_offsets is clamped to the entries B holds: the input check that says whether
    // those offsets ascend runs BESIDE this kernel, and every later kernel takes its B rows from the pairs written here --
    // so no kernel of the call indexes B.col_ids / B.data out of bounds whatever B.row_offsets holds.
    const u32 b_lo = b_ro[0];
    const u32 b_hi = (u32)std::min<u64>(u64(b_lo) + b_nnz, 0xFFFFFFFFull);
    __shared__ u32 s_ro_all[NW][R + 1];
    __shared__ u64 s_ops_all[NW][R];
    __shared__ u32 s_mx_all[NW][R], s_cmin_all[NW][R], s_cmax_all[NW][R];
    __shared__ u64 s_products[NW], s_nf[NW];
    __shared__ u32 s_max[NW], s_nfr[NW], s_badcol[NW];
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
    bool bad_col = false;  // a column id of A beyond the rows of B: clamped here, reported through the chain
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
                bs[u] = ok[u] ? min(max(pr.x, b_lo), b_hi) : 0u;
                be[u] = ok[u] ? min(max(pr.y, bs[u]), b_hi) : 0u;
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
                // one-walk call (walk.hip): a register-class row is finished into a slot as well
                if (cp.one_walk && cls < SYM_CLASSES && ((kSymEscMask >> cls) & 1u)) my_nf += nf_slot_entries(cmin, cmax, ops32);
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
                    bs[u] = ok[u] ? min(max(pr.x, b_lo), b_hi) : 0u;
                    be[u] = ok[u] ? min(max(pr.y, bs[u]), b_hi) : 0u;
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
    const bool wave_bad = __ballot(bad_col) != 0;
    __syncthreads();
    AN_MARK(6);
    if (lane == 0) {
        s_products[wid] = my_products;
        s_nf[wid] = my_nf;
        s_max[wid] = my_max;
        s_nfr[wid] = my_nfr;
        s_badcol[wid] = wave_bad ? 1u : 0u;
#pragma unroll
        for (int c = 0; c < SYM_CLASSES; ++c) s_hist[wid][c] = hist[c];
    }
    __syncthreads();
    // ---- my aggregate -> the chain -> what the workgroups before me found (chain.hpp)
    __shared__ u32 s_mine[kChainWords];
    __shared__ u64 s_pref[kChainWords], s_tmp[2 * kChainWords + 2];
    if (t < kChainWords) {
        u32 v = 0;
        if (t < SYM_CLASSES)
            for (int w = 0; w < NW; ++w) v += s_hist[w][t];
        u64 p = 0, nf = 0;
        u32 mxv = 0, nfr = 0, bad = 0;
        for (int w = 0; w < NW; ++w) {
            p += s_products[w];
            nf += s_nf[w];
            mxv = max(mxv, s_max[w]);
            nfr = max(nfr, s_nfr[w]);
            bad |= s_badcol[w];
        }
        if (t == kCwPfxLo) v = (u32)nf;
        if (t == kCwPfxHi) v = (u32)(nf >> 32);
        if (t == kCwTotLo) v = (u32)p;
        if (t == kCwTotHi) v = (u32)(p >> 32);
        if (t == kCwFlags) v = bad;
        if (t == kCwMax) v = mxv;
        if (t == kCwAuxMax) v = nfr;
        s_mine[t] = v;
    }
    if (cp.want_bytes && bytes_acc && t < SYM_CLASSES && s_bytes[t]) atomicAdd((unsigned long long*)&bytes_acc[t], (unsigned long long)s_bytes[t]);
    __threadfence_block();  // my rows' classes and bounds (global, written by the waves of this block) before the reads below
    __syncthreads();
    const u32 nb = gridDim.x;
    // my aggregate goes out FIRST; what the records of my first chunk of rows need is requested while it travels
    chain_publish_own(chain, blockIdx.x, s_mine);
    u32 p_c = 0xFFu, p_a0 = 0, p_a1 = 0, p_min = 0, p_max = 0, p_ops = 0;
    if (sym_cls && t < kChunk && row_begin + t < row_end) {
        const u32 row = row_begin + t;
        p_c = sym_cls[row];
        p_a0 = a_ro[row];
        p_a1 = a_ro[row + 1];
        p_min = row_col_min[row];
        p_max = row_col_max[row];
        p_ops = row_ops[row];
    }
    // (false: a wait of the chain timed out -- this workgroup places nothing; the last one reports, chain.hpp)
    const bool chain_ok = chain_exclusive(chain, blockIdx.x, nb, s_mine, s_pref, s_tmp);
    AN_MARK(7);

    // ---- binning: my rows' records (row order) and their row ids in the class lists, behind the rows of the
    // workgroups before me; scratch slots of the numeric-first rows / key sets of the SYM_GH rows in row order
    if (sym_cls && chain_ok) {
        __shared__ u32 s_wcnt[SYM_CLASSES][kChunk / 64];
        __shared__ u32 s_run[SYM_CLASSES];
        __shared__ u32 s_nfscan[kChunk / 64 + 2];
        __shared__ u64 s_nfrun;
        if (t < SYM_CLASSES) s_run[t] = (u32)s_pref[kCwClass + t];
        if (t == 0) s_nfrun = chain_u64(s_pref, kCwPfxLo, kCwPfxHi);
        __syncthreads();
        for (u32 row0 = row_begin; row0 < row_end; row0 += kChunk) {
            const u32 row = row0 + t;
            const bool in = t < kChunk && row < row_end, first = row0 == row_begin;
            const u32 c = first ? p_c : (in ? sym_cls[row] : 0xFFu);
            u32 my_rank = 0, r_ops = p_ops, r_min = p_min, r_max = p_max, r_a0 = p_a0, r_a1 = p_a1;
            if (t < kChunk) {
#pragma unroll
                for (u32 b = 0; b < SYM_CLASSES; ++b) {
                    const u64 mask = __ballot(c == b);
                    if (lane == 0) s_wcnt[b][wid] = __popcll(mask);
                    if (c == b) my_rank = __popcll(mask & lanemask_lt());
                }
            }
            if (!first && c < SYM_CLASSES) {
                r_ops = row_ops[row];
                r_min = row_col_min[row];
                r_max = row_col_max[row];
                r_a0 = a_ro[row];
                r_a1 = a_ro[row + 1];
            }
            __syncthreads();
            if (c < SYM_CLASSES) {
                u32 pos = s_run[c] + my_rank;
                for (u32 w = 0; w < wid; ++w) pos += s_wcnt[c][w];
                RowRec r;
                r.row = row;
                r.a0 = r_a0;
                r.a1 = r_a1;
                r.base = 0;
                r.cmin = r_min;
                r.cmax = r_max;
                r.ops = r_ops;
                r.nnz = 0;
                if (pos < m) *class_rec_at(sym_recs, m, c, pos) = r;   // (pos < m always; a chain that timed out must not scribble)
            }
            u32 any_slot = 0;
            for (int w = 0; w < kChunk / 64; ++w) any_slot += s_wcnt[SYM_NF][w] + s_wcnt[SYM_GH][w];
            if (cp.one_walk)
                for (int w = 0; w < kChunk / 64; ++w)
                    any_slot += s_wcnt[SYM_G8][w] + s_wcnt[SYM_G16][w] + s_wcnt[SYM_R32][w] + s_wcnt[SYM_R64][w];
            const bool esc_slot = cp.one_walk && c < SYM_CLASSES && ((kSymEscMask >> c) & 1u);
            if (any_slot) {  // (uniform)
                const u32 ub = (c == SYM_NF || esc_slot) ? nf_slot_entries(r_min, r_max, r_ops) : (c == SYM_GH ? gh_table_slots(r_ops) : 0u);
                // (the scan runs over the first kChunk threads' values: the others hold 0)
                u32 chunk_total = 0;
                u32 excl = 0;
                if (t < kChunk) {
                    const u32 incl = wave_inclusive_scan(ub);
                    if (lane == 63) s_nfscan[wid] = incl;
                    excl = incl - ub;
                }
                __syncthreads();
                if (t < kChunk) {
                    for (u32 w = 0; w < wid; ++w) excl += s_nfscan[w];
                    for (int w = 0; w < kChunk / 64; ++w) chunk_total += s_nfscan[w];
                    if (c == SYM_NF || c == SYM_GH || esc_slot) nf_off[row] = s_nfrun + excl;
                }
                __syncthreads();
                if (t == 0) s_nfrun += chunk_total;
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
    AN_MARK(8);
    if (blockIdx.x != nb - 1) return;
    // ---- the LAST workgroup has seen everything: it writes the statistics block of the call -- ALL of it, zeros
    // included (no memset node, and no other workgroup of this kernel touches the block: two XCDs' L2s writing the
    // same word would race at write-back)
    if (t < sizeof(DeviceStats) / 4) reinterpret_cast<u32*>(st)[t] = 0;
    for (u32 i = t + kAnThreads; i < sizeof(DeviceStats) / 4; i += kAnThreads) reinterpret_cast<u32*>(st)[i] = 0;
    __syncthreads();
    if (t < kMaxClasses) {
        const u32 total = t < SYM_CLASSES ? (u32)s_pref[kCwClass + t] + s_mine[kCwClass + t] : 0u;
        st->sym.count[t] = total;
        st->sym.offset[t] = 0;
        // a replayed launch sequence only carries the kernels of `sym_allowed`
        if (total && !((cp.sym_allowed >> t) & 1u)) st->capacity_miss = 1;
    }
    if (t == 0) {
        const u64 products = chain_u64(s_pref, kCwTotLo, kCwTotHi) + (u64(s_mine[kCwTotHi]) << 32) + s_mine[kCwTotLo];
        const u64 nf = chain_u64(s_pref, kCwPfxLo, kCwPfxHi) + (u64(s_mine[kCwPfxHi]) << 32) + s_mine[kCwPfxLo];
        st->sum_products = products;
        st->max_row_ops = max((u32)s_pref[kCwMax], s_mine[kCwMax]);
        st->nf_entries = nf;
        st->nf_max_range = max((u32)s_pref[kCwAuxMax], s_mine[kCwAuxMax]);
        if (s_pref[kCwFlags] + s_mine[kCwFlags] != 0) {  // a column id of A >= rows(B): clamped by the walk, reported here
            st->a_invalid = 1;
            st->capacity_miss = 1;  // a replayed sequence stops here; the eager path returns the status
        }
        // the scratch pool of a launch sequence sized from an earlier call holds `expect_nf` entries
        if (expect_nf != ~0ull && nf > expect_nf) st->capacity_miss = 1;
        if (!chain_ok || chain_error(chain)) {  // some workgroup placed nothing: the kernels queued behind walk nothing either
            st->chain_error = 1;
            st->capacity_miss = 1;
        }
    }
}

// --------------------------------------------------------------------------------
// row_offsets scan, fused with the numeric classification and binning: ONE pass.  A tile is 256 threads x ITEMS
// consecutive rows per thread (x SUB sub-tiles for inputs beyond 2^25 rows, so that the chain never sees more than
// kChainMaxBlocks workgroups).  Per tile: nnz sum + class histogram -> the chain -> row_offsets = (nnz before my tile)
// + local scan, RowRec.base / nnz of every row, the row's id in its numeric class list at (rows of the class before my
// tile) + rank (ascending rows inside a class).  The last tile writes the statistics the host and the numeric kernels read.
// Traffic: 8(m+1) B for the scan itself + 16 m read for the classification + 36 m for records and lists.
// --------------------------------------------------------------------------------
constexpr int kScanThreads = 256;
// rows per thread: small inputs still get >= ~300 tiles, big ones at most kChainMaxBlocks.  32 rows per thread are a
// last resort: a thread's rows are contiguous, so every load instruction of a wave touches 64 cache lines.
// (rows per tile for rows(A) <= 2^19: 256 x 3 -- 223 tiles for the 171 k rows of the scircuit stand-in, one per CU, instead of 334
//  with 256 x 2: complete call 0.1363 -> 0.1345 ms, uniform 0.1257 -> 0.1238, mac_econ +-0, two alternating pairs; 256 x 4: +-0.
//  Option scan_small_items)
static int g_scan_small = 3;
void set_scan_small_items(int i) { g_scan_small = (i == 4 || i == 2) ? i : 3; }
static inline int scan_items(u32 m) { return m <= (1u << 19) ? g_scan_small : (m <= (1u << 23) ? 8 : 32); }
static inline u32 scan_subtiles(u32 m) { return m <= (1u << 25) ? 1u : cdiv(m, (1u << 25)); }
// One 16-bit counter per numeric class, four per u64: a count never exceeds the rows of a sub-tile
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

template <int ITEMS, bool WIDE>
__global__ __launch_bounds__(kScanThreads) void scan_kernel(
    const u32* __restrict__ counts, u32* __restrict__ offsets_out, u32 m, u32 sub, DeviceStats* __restrict__ st, Chain chain,
    const u32* __restrict__ a_ro, const u32* __restrict__ row_ops, const u32* __restrict__ row_col_min,
    const u32* __restrict__ row_col_max, RowRec* __restrict__ recs /* numeric class lists; nullptr: offsets only */,
    ClassifyParams cp, u32 vsize, u64 exact_nnz, u64 expect_g, u32 expect_g_rows, DeviceStats* __restrict__ host_mirror,
    const u32* __restrict__ pred_off, u32* __restrict__ pred_off_out, u32* __restrict__ dev_ticket,
    u32* __restrict__ host_ticket, u64* __restrict__ bytes_acc, const u32* __restrict__ gate, u32 gate_ticket)
{
    SPECK_POISON();
    constexpr int NW = kScanThreads / 64;
    constexpr u32 kSubRows = kScanThreads * ITEMS;
    __shared__ u32 s_mine[kChainWords];
    __shared__ u64 s_pref[kChainWords], s_tmp[2 * kChainWords + 2];
    __shared__ u32 s_scan[NW + 1];
    __shared__ u64 s_wave[NW][4];
    __shared__ u64 s_sum[NW], s_gops[NW];
    __shared__ u32 s_max[NW];
    __shared__ u32 s_hist[NW][kMaxClasses];
    __shared__ u64 s_bytes[kMaxClasses];
    __shared__ u32 s_run[kMaxClasses];  // rows of each class before the current sub-tile
    __shared__ u64 s_off_run;           // nnz before the current sub-tile
    const u32 lane = lane_id(), wid = threadIdx.x >> 6, t = threadIdx.x;
    const u64 tile0 = u64(blockIdx.x) * kSubRows * sub;
    u32 c[ITEMS];
    u8 cls[ITEMS];
    if (t < kMaxClasses) s_bytes[t] = 0;
    u64 g_ops = 0;
    // one sub-tile's rows into registers (+ their classes; what the classification reads is read again, from the L2,
    // when the records are written: 32 rows per thread would not fit the register file otherwise)
    auto load = [&](u32 s, bool count_g) {
        const u64 base = tile0 + u64(s) * kSubRows + u64(t) * ITEMS;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const u64 row = base + i;
            const bool in = row < m;
            c[i] = in ? counts[row] : 0u;
            cls[i] = NUM_NONE;
            if (recs && in) {
                const u32 len_a = a_ro[row + 1] - a_ro[row], ops = row_ops[row];
                cls[i] = classify_numeric(len_a, ops, c[i], row_col_min[row], row_col_max[row], cp);
                if (count_g && cls[i] == NUM_G) g_ops += ops;
                if (count_g && cp.want_bytes && cls[i] != NUM_NONE)
                    atomicAdd((unsigned long long*)&s_bytes[cls[i]], (unsigned long long)numeric_row_bytes(len_a, ops, c[i], vsize));
            }
        }
    };
    // ---- pass 1: the tile's aggregate
    u64 tsum = 0;
    u32 my_max = 0;
    PackedCounts packed_all;   // (per thread <= 32 x sub rows of a class: fits 16 bits up to sub = 2048)
    u32 hist_wide[WIDE ? kMaxClasses : 1];
    constexpr bool wide = WIDE;  // (sub > 1) counts of several sub-tiles: 32-bit accumulators
    if constexpr (WIDE) {
#pragma unroll
        for (int k = 0; k < kMaxClasses; ++k) hist_wide[k] = 0;
    }
    __syncthreads();
    for (u32 s = 0; s < sub; ++s) {
        load(s, true);
        PackedCounts packed;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            tsum += c[i];
            my_max = max(my_max, c[i]);
            if (cls[i] != NUM_NONE) packed.add(cls[i]);
        }
        if constexpr (WIDE) {
#pragma unroll
            for (int k = 0; k < kMaxClasses; ++k) hist_wide[k] += packed.get(k);
        } else
            packed_all = packed;
    }
    tsum = wave_reduce_add(tsum);
    g_ops = wave_reduce_add(g_ops);
    my_max = wave_reduce_max(my_max);
    if constexpr (WIDE) {
#pragma unroll
        for (int k = 0; k < kMaxClasses; ++k) {
            const u32 v = wave_reduce_add(hist_wide[k]);
            if (lane == 0) s_hist[wid][k] = v;
        }
    } else {
        packed_all.a = wave_reduce_add(packed_all.a);
        packed_all.b = wave_reduce_add(packed_all.b);
        packed_all.c = wave_reduce_add(packed_all.c);
        packed_all.d = wave_reduce_add(packed_all.d);
        if (lane == 0)
#pragma unroll
            for (int k = 0; k < kMaxClasses; ++k) s_hist[wid][k] = packed_all.get(k);
    }
    if (lane == 0) {
        s_sum[wid] = tsum;
        s_gops[wid] = g_ops;
        s_max[wid] = my_max;
    }
    __syncthreads();
    if (t < kChainWords) {
        u32 v = 0;
        if (t < NUM_CLASSES)
            for (int w = 0; w < NW; ++w) v += s_hist[w][t];
        u64 sum = 0, gs = 0;
        u32 mx = 0;
        for (int w = 0; w < NW; ++w) {
            sum += s_sum[w];
            gs += s_gops[w];
            mx = max(mx, s_max[w]);
        }
        if (t == kCwTotLo) v = (u32)gs;
        if (t == kCwTotHi) v = (u32)(gs >> 32);
        if (t == kCwPfxLo) v = (u32)sum;
        if (t == kCwPfxHi) v = (u32)(sum >> 32);
        if (t == kCwMax) v = mx;
        s_mine[t] = v;
    }
    if (cp.want_bytes && bytes_acc && t < NUM_CLASSES && s_bytes[t]) atomicAdd((unsigned long long*)&bytes_acc[kMaxClasses + t], (unsigned long long)s_bytes[t]);
    __syncthreads();
    const u32 nb = gridDim.x;
    chain_publish_own(chain, blockIdx.x, s_mine);
    const bool chain_ok = chain_exclusive(chain, blockIdx.x, nb, s_mine, s_pref, s_tmp);
    const u64 nnz_before = chain_u64(s_pref, kCwPfxLo, kCwPfxHi);
    const bool last = blockIdx.x == nb - 1;
    // ---- the LAST tile has the totals: statistics, the checks of a sequence sized from an earlier call, and -- eager
    // call -- the mirror + ticket for the host, NOW: nothing the writes below do changes what the host needs
    if (last) {
        u32 front_miss = 0;
        if (t == 0) front_miss = st->capacity_miss;  // (before this kernel's own checks below)
        __syncthreads();
            const u64 nnz_c = nnz_before + (u64(s_mine[kCwPfxHi]) << 32) + s_mine[kCwPfxLo];
        const u64 g_total = chain_u64(s_pref, kCwTotLo, kCwTotHi) + (u64(s_mine[kCwTotHi]) << 32) + s_mine[kCwTotLo];
        if (t < kMaxClasses && recs) {
            const u32 total = t < NUM_CLASSES ? (u32)s_pref[kCwClass + t] + s_mine[kCwClass + t] : 0u;
            st->num.count[t] = total;
            st->num.offset[t] = 0;
            if (total && !((cp.num_allowed >> t) & 1u)) st->capacity_miss = 1;
            if (t == NUM_G && expect_g_rows != ~0u && total != expect_g_rows) st->capacity_miss = 1;
        }
        if (t == 0) {
            st->nnz_c = nnz_c;
            st->max_row_nnz_c = max((u32)s_pref[kCwMax], s_mine[kCwMax]);
            if (nnz_c > 0xFFFFFFFFull) st->nnz_overflow = 1;
            // the C buffers of a replayed launch sequence were allocated for exactly `exact_nnz`
            if (exact_nnz != ~0ull && nnz_c != exact_nnz) st->capacity_miss = 1;
            // ... and so was the spill pool of the NUM_G rows
            st->g_products = g_total;
            st->front_miss = front_miss;
            if (expect_g != ~0ull && g_total != expect_g) st->capacity_miss = 1;
            // a call whose numeric launches are already queued behind this kernel (pipeline.hip, the through call): the
            // input check of B has finished (the stream waited for it) -- a violation voids them, C stays as it is
            // (gate_ticket: the check's stream was NOT joined -- a cross-queue wait costs ~6 us even on a finished branch --
            //  the check counts as done only if its ticket kernel has stored this call's ticket; else: void, the call re-runs)
            if (gate) {
                const u32 ticket = gate_ticket ? __hip_atomic_load(gate + 16, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
                const u32 verdict = __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((verdict & 4u) || ticket != gate_ticket) st->capacity_miss = 1;
            }
            if (!chain_ok || chain_error(chain)) {
                st->chain_error = 1;
                st->capacity_miss = 1;
            }
        }
        __syncthreads();
        if (host_mirror && wid == 0) {
            const u64* src = reinterpret_cast<const u64*>(st);
            u64* dst = reinterpret_cast<u64*>(host_mirror);
            for (u32 i = lane; i < sizeof(DeviceStats) / 8; i += 64)
                __hip_atomic_store(&dst[i], src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: the mirror is written before the ticket
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                const u32 tk = *dev_ticket + 1u;
                *dev_ticket = tk;
                __hip_atomic_store(host_ticket, tk, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        if (t == 0) {
            offsets_out[m] = (u32)nnz_c;
            if (pred_off_out) pred_off_out[m] = (u32)nnz_c;
            if (pred_off && pred_off[m] != (u32)nnz_c) st->capacity_miss = 1;
        }
    }
    // ---- pass 2: offsets, records, class lists (nothing from a truncated prefix: chain.hpp)
    if (!chain_ok) return;
    if (t < kMaxClasses) s_run[t] = (u32)s_pref[kCwClass + t];
    if (t == 0) s_off_run = nnz_before;
    __syncthreads();
    for (u32 s = 0; s < sub; ++s) {
        if (wide) load(s, false);
        const u64 base = tile0 + u64(s) * kSubRows + u64(t) * ITEMS;
        u32 tsum32 = 0;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) tsum32 += c[i];
        u32 total;
        const u32 excl = block_exclusive_scan<kScanThreads>(tsum32, s_scan, &total);
        u32 run = (u32)s_off_run + excl;
        u32 off[ITEMS];
        bool moved = false;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            off[i] = run;
            if (base + i < m) {
                offsets_out[base + i] = run;
                if (pred_off_out) pred_off_out[base + i] = run;  // the config's own copy: C.row_offsets is the caller's
                // rows a replayed sequence has already placed by the previous call's offsets: the fresh ones must agree
                // (checked for EVERY row: a shift anywhere before such a row moves it)
                if (pred_off && pred_off[base + i] != run) moved = true;
            }
            run += c[i];
        }
        if (pred_off && __ballot(moved) != 0 && lane == 0) st->capacity_miss = 1;  // (only ever set, by whoever objects)
        if (recs) {
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
                RowRec r;
                r.row = (u32)row;
                r.a0 = a_ro[row];
                r.a1 = a_ro[row + 1];
                r.base = off[i];
                r.cmin = row_col_min[row];
                r.cmax = row_col_max[row];
                r.ops = row_ops[row];
                r.nnz = c[i];
                const u32 k = cls[i];
                const u32 pos = s_run[k] + before.get(k) + used.get(k);
                used.add(k);
                if (k == NUM_NFCOPY && pred_off) continue;  // already in place: no launch reads that list
                if (pos < m) *class_rec_at(recs, m, k, pos) = r;
            }
            __syncthreads();
            if (t < kMaxClasses) {
                PackedCounts all;
                for (int w = 0; w < NW; ++w) {
                    all.a += s_wave[w][0];
                    all.b += s_wave[w][1];
                    all.c += s_wave[w][2];
                    all.d += s_wave[w][3];
                }
                s_run[t] += all.get(t);
            }
        }
        if (t == 0) s_off_run += total;
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------
// Input precondition (undocumented upstream): the column ids of every row of B are
// strictly ascending (the min/max column range of the analysis, the scaled-copy rows and the bitmap
// sorts rely on it; the reference's loader guarantees it and silently computes garbage otherwise).
// One coalesced pass over B.col_ids with EVERY call, complete or reuse sequence (the interior of a row is no input of
// the analysis: nothing else would notice ids reordered in place): validate_b_kernel, beside the call on the verifier's
// stream.
// --------------------------------------------------------------------------------
// Last node of a replayed launch sequence: a ticket in pinned host memory the host spins on (a blocking
// stream synchronisation costs ~10-20 us of wake-up latency: a tenth of a 200 us multiply).
// The same wave first copies the (final) statistics block into the pinned mirror with system-scope stores, so
// the host may read it as soon as it sees the ticket: nothing relies on an earlier kernel's plain stores to
// host memory being visible by then.
__global__ __launch_bounds__(64) void done_kernel(u32* __restrict__ dev_ticket, u32* __restrict__ host_ticket,
                                                  const DeviceStats* __restrict__ st, DeviceStats* __restrict__ host_mirror)
{
    SPECK_POISON();
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
constexpr u32 kValChunk = 8192;  // entries of B per workgroup and step: eight 16-byte loads per lane, all in flight
__global__ __launch_bounds__(256) void validate_b_kernel(const u32* __restrict__ b_ro, const u32* __restrict__ b_col, u32 b_rows,
                                                          u32 b_cols, u64 b_nnz, u32* __restrict__ verdict)
{
    SPECK_POISON();
    // O(nnz + rows), streaming: a workgroup owns a contiguous span of entries and walks the row offsets alongside it.  Per
    // chunk of 8192 entries: the row STARTS inside the chunk as a bitmap in LDS (a pair of neighbours that does not ascend
    // is fine exactly where a row starts), then the entries as 16-byte loads.  (Round 4 looked every such pair up by binary
    // search over B.row_offsets -- once per row of B, 23 dependent loads each: 5.9 ms for the nlpkkt stand-in.)
    __shared__ u32 s_bits[kValChunk / 32];
    __shared__ u32 s_cursor;
    const u32 tid = threadIdx.x;
    // (the offsets span exactly the nnz entries the container holds: an array whose last offset lies beyond them would
    //  send this very kernel out of bounds -- ADVICE round 5)
    const u32 e_first = b_ro[0];
    u32 e_last = b_ro[b_rows];
    bool bad = e_last < e_first || u64(e_last) - e_first != b_nnz;
    if (u64(e_last) > u64(e_first) + b_nnz) e_last = (u32)(u64(e_first) + b_nnz);
    // the offsets themselves: ascending (so every one of them lies in [e_first, e_last])
    for (u64 r = u64(blockIdx.x) * 256 + tid; r < b_rows; r += u64(gridDim.x) * 256) bad |= b_ro[r + 1] < b_ro[r];
    const u64 base = e_first & ~3u;  // (groups of four aligned in the array, so that the 16-byte loads are)
    const u64 nchunks = e_last > base ? (u64(e_last) - base + kValChunk - 1) / kValChunk : 0;
    const u64 per = (nchunks + gridDim.x - 1) / gridDim.x;
    const u64 c0 = u64(blockIdx.x) * per, c1 = c0 + per < nchunks ? c0 + per : nchunks;
    if (c0 < c1 && e_last >= e_first) {
        if (tid < 64) {  // first row that starts behind the first entry of the span: a 64-ary search by one wave
            const u64 cs0 = base + c0 * kValChunk;
            u32 lo = 0, hi = b_rows + 1;  // rows below lo start at or before cs0, rows from hi on behind it
            while (lo < hi) {
                const u32 step = (hi - lo + 63u) >> 6;
                const u64 p = u64(lo) + u64(tid) * step;
                const bool behind = p >= hi || u64(b_ro[p]) > cs0;
                const u64 mask = __ballot(behind);
                const u32 f = mask ? (u32)__builtin_ctzll(mask) : 64u;
                const u32 lo0 = lo;
                if (f < 64u) hi = min(hi, lo0 + f * step);
                if (f > 0u) lo = lo0 + (f - 1u) * step + 1u;
            }
            if (tid == 0) s_cursor = lo;
        }
        __syncthreads();
        u32 rc = s_cursor;
        for (u64 c = c0; c < c1; ++c) {
            const u64 cs = base + c * kValChunk;
            s_bits[tid] = 0;
            __syncthreads();
            // bit i: a row starts at entry cs + 1 + i
            while (true) {
                const u64 r = u64(rc) + tid;
                bool in = false;
                if (r <= b_rows) {
                    const u64 v = b_ro[r];
                    in = v <= cs + kValChunk;
                    if (in && v > cs) atomicOr(&s_bits[(u32)(v - cs - 1) >> 5], 1u << ((u32)(v - cs - 1) & 31u));
                }
                const u32 n = (u32)__syncthreads_count(in);  // (offsets that do not ascend are `bad` already)
                rc += n;
                if (n < 256u) break;
            }
            uint4 q[8];
            u32 nx[8];
#pragma unroll
            for (u32 k = 0; k < 8; ++k) {
                const u64 e = cs + 4ull * (tid + 256u * k);
                q[k] = make_uint4(0u, 0u, 0u, 0u);
                nx[k] = 0xFFFFFFFFu;
                if (e + 4 <= e_last) q[k] = *reinterpret_cast<const uint4*>(b_col + e);
                else if (e < e_last)  // (the last, partial group: nothing beyond the array is touched)
                    q[k] = make_uint4(b_col[e], e + 1 < e_last ? b_col[e + 1] : 0u, e + 2 < e_last ? b_col[e + 2] : 0u, 0u);
                if (e + 4 < e_last) nx[k] = b_col[e + 4];
            }
#pragma unroll
            for (u32 k = 0; k < 8; ++k) {
                const u64 e = cs + 4ull * (tid + 256u * k);
                const u32 col[5] = {q[k].x, q[k].y, q[k].z, q[k].w, nx[k]};
#pragma unroll
                for (u32 j = 0; j < 4; ++j) {
                    const u64 at = e + j;
                    if (at < e_first || at >= e_last) continue;
                    if (col[j] >= b_cols) bad = true;
                    if (at + 1 < e_last && col[j + 1] <= col[j]) {
                        const u32 bit = (u32)(at - cs);  // (entry at + 1)
                        if (!((s_bits[bit >> 5] >> (bit & 31u)) & 1u)) bad = true;
                    }
                }
            }
            __syncthreads();  // the next chunk clears the bitmap
        }
    }
    if (__ballot(bad) != 0 && lane_id() == 0) __hip_atomic_fetch_or(verdict, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_validate_b(hipStream_t s, const u32* b_ro, const u32* b_col, u32 b_rows, u32 b_cols, u64 b_nnz, u32* verdict)
{
    if (b_rows == 0) return;
    const u64 want = std::max<u64>(cdiv(b_nnz + 4, kValChunk), cdiv(u64(b_rows), 256 * 16));
    SPECK_LAUNCH(validate_b_kernel, dim3((u32)std::min<u64>(std::max<u64>(want, 1), 2048)), dim3(256), 0, s, b_ro, b_col,
                       b_rows, b_cols, b_nnz, verdict);
}

// ... and the ticket of the verifier's stream (launch_verifier): the kernel boundary in front of it orders the verifier's
// verdict (system-scope atomics on pinned memory) before the ticket
__global__ __launch_bounds__(64) void ticket_kernel(u32* __restrict__ dev_ticket, u32* __restrict__ host_ticket)
{
    SPECK_POISON();
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
    SPECK_POISON();
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
    SPECK_POISON();
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
    SPECK_LAUNCH(snapshot_inputs_kernel, dim3(input_blocks(nnz_a, b_rows)), dim3(256), 0, s, a_ro, a_col, a_col_copy,
                       nnz_a, b_ro, b_col, b_rows, b_snap);
}
void launch_verify_inputs(hipStream_t s, const u32* a_ro, const u32* a_ro_copy, u32 m, const u32* a_col,
                          const u32* a_col_copy, u64 nnz_a, const u32* b_ro, const u32* b_col, u32 b_rows, const u32* b_snap,
                          u32* verdict)
{
    SPECK_LAUNCH(verify_inputs_kernel, dim3(input_blocks(nnz_a, b_rows)), dim3(256), 0, s, a_ro, a_ro_copy, m, a_col,
                       a_col_copy, nnz_a, b_ro, b_col, b_rows, b_snap, verdict);
}

void launch_ticket(hipStream_t s, u32* dev_ticket, u32* host_ticket)
{
    SPECK_LAUNCH(ticket_kernel, dim3(1), dim3(64), 0, s, dev_ticket, host_ticket);
}
void launch_done(hipStream_t s, u32* dev_ticket, u32* host_ticket, const DeviceStats* st, DeviceStats* host_mirror)
{
    SPECK_LAUNCH(done_kernel, dim3(1), dim3(64), 0, s, dev_ticket, host_ticket, st, host_mirror);
}
// the staged row offsets of a call -> C.row_offsets, for a numeric phase without the light launch that carries them
// (RowWork::off_src); a sequence an earlier kernel has declared void leaves the caller's buffer alone
__global__ __launch_bounds__(256) void copy_offsets_kernel(const u32* __restrict__ src, u32* __restrict__ dst, u32 n,
                                                            const DeviceStats* __restrict__ st)
{
    SPECK_POISON();
    if (st->capacity_miss) return;
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) dst[i] = src[i];
}
void launch_copy_offsets(hipStream_t s, const u32* src, u32* dst, u32 n, const DeviceStats* st)
{
    SPECK_LAUNCH(copy_offsets_kernel, dim3(std::min<u32>(cdiv(n ? n : 1, 2048), 512u)), dim3(256), 0, s, src, dst, n, st);
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
u32 scan_tiles(u32 m) { return cdiv(m ? m : 1, u64(kScanThreads) * scan_items(m) * scan_subtiles(m)); }

void launch_analysis(hipStream_t s, const u32* a_ro, const u32* a_col, const u32* b_ro, const u32* b_col, u32 m, u64 nnz_a,
                     u32* row_ops, u32* row_max_ops, u32* row_col_min, u32* row_col_max, u8* sym_cls, u32* counts,
                     RowRec* sym_recs, DeviceStats* st, const ClassifyParams& cp, uint2* b_sl, const Chain& chain,
                     u64* nf_off, u64 expect_nf, u32 b_rows, u32* a_ro_copy, u32* verdict, u64* bytes_acc, u64 b_nnz)
{
    u32 rows_per_block, blocks;
    row_chunking(m, &rows_per_block, &blocks);
    // 64 rows per wave for short rows (g_an_wide_rows: average entries per row up to which; 0 = never)
    const bool wide = g_an_wide_rows && m && nnz_a / m <= g_an_wide_rows;
    auto go = [&](auto kernel, int threads) {
        SPECK_LAUNCH(kernel, dim3(blocks), dim3(threads), 0, s, a_ro, a_col, b_ro, b_col, m, rows_per_block, row_ops,
                           row_max_ops, row_col_min, row_col_max, sym_cls, counts, cp, b_sl, st, b_rows, sym_recs, nf_off,
                           expect_nf, chain, a_ro_copy, verdict, bytes_acc, b_nnz);
    };
    if (verdict) {  // replayed sequence with the analysis beside it: compare, write nothing (analysis_kernel, VERIFY)
        if (wide) go(analysis_kernel<4, 64, true>, 256);
        else go(analysis_kernel<8, 32, true>, 512);
    } else if (wide)
        go(analysis_kernel<4, 64>, 256);
    else
        go(analysis_kernel<8, 32>, 512);
}

void launch_scan(hipStream_t s, const u32* counts, u32* offsets_out, u32 m, const u32* a_ro, const u32* row_ops,
                 const u32* row_col_min, const u32* row_col_max, RowRec* num_recs, DeviceStats* st,
                 const ClassifyParams& cp, u32 vsize, u64 exact_nnz, const Chain& chain, DeviceStats* host_mirror, u64 expect_g,
                 u32 expect_g_rows, const u32* pred_off, u32* pred_off_out, u32* dev_ticket, u32* host_ticket, u64* bytes_acc,
                 const u32* gate, u32 gate_ticket)
{
    const u32 tiles = scan_tiles(m), sub = scan_subtiles(m);
    auto go = [&](auto items) {
        constexpr int I = decltype(items)::value;
        if (sub > 1)
            SPECK_LAUNCH((scan_kernel<32, true>), dim3(tiles), dim3(kScanThreads), 0, s, counts, offsets_out, m, sub, st, chain, a_ro,
                               row_ops, row_col_min, row_col_max, num_recs, cp, vsize, exact_nnz, expect_g, expect_g_rows,
                               host_mirror, pred_off, pred_off_out, dev_ticket, host_ticket, bytes_acc, gate, gate_ticket);
        else
        SPECK_LAUNCH((scan_kernel<I, false>), dim3(tiles), dim3(kScanThreads), 0, s, counts, offsets_out, m, sub, st, chain, a_ro,
                           row_ops, row_col_min, row_col_max, num_recs, cp, vsize, exact_nnz, expect_g, expect_g_rows,
                           host_mirror, pred_off, pred_off_out, dev_ticket, host_ticket, bytes_acc, gate, gate_ticket);
    };
    switch (scan_items(m)) {
        case 2: go(std::integral_constant<int, 2>{}); break;
        case 3: go(std::integral_constant<int, 3>{}); break;
        case 4: go(std::integral_constant<int, 4>{}); break;
        case 8: go(std::integral_constant<int, 8>{}); break;
        default: go(std::integral_constant<int, 32>{}); break;
    }
}

}  // namespace speck
#ifdef SPECK_PHASE_CLOCKS
extern "C" int speck_debug_analysis_clocks(unsigned long long* out16)
{
    static unsigned long long all[1024 * 16];
    if (hipMemcpyFromSymbol(all, HIP_SYMBOL(speck::g_an_clk), sizeof(all)) != hipSuccess) return 3;
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    for (int b = 0; b < 1024; ++b)
        for (int i = 0; i < 16; ++i) out16[i] += all[b * 16 + i];
    for (auto& x : all) x = 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(speck::g_an_clk), all, sizeof(all)) == hipSuccess ? 0 : 3;
}
#endif
