// launch.hpp -- host-side launcher declarations (one per pipeline stage).
#pragma once
#include <hip/hip_ext.h>

#include "device_common.hpp"

namespace speck {

u32 analysis_blocks(u32 m);
void set_analysis_wide_rows(u32 avg_len);  // analysis: 64 rows per wave when nnz(A) / rows(A) <= avg_len (0: never)
u32 scan_tiles(u32 m);

// analysis (+ stats fold + ordered scatter of the symbolic row records when sym_cls != nullptr)
void launch_analysis(hipStream_t s, const u32* a_ro, const u32* a_col, const u32* b_ro,
                     const u32* b_col, u32 m, u64 nnz_a, u32* row_ops, u32* row_max_ops,
                     u32* row_col_min, u32* row_col_max, u8* sym_cls, u32* counts,
                     BlockPartial* partials, RowRec* recs, DeviceStats* st, const ClassifyParams& cp,
                     uint2* b_sl, hipEvent_t between = nullptr, u64* nf_off = nullptr,
                     u64 expect_nf = ~0ull, u32 b_rows = ~0u, u32* pred_block_out = nullptr,
                     const u32* pred_block = nullptr, const DeviceStats* pred_stats = nullptr, u32 b_cols = 0,
                     u64 b_nnz = 0, u32 validate_epoch = 0 /* != 0: also check B's rows (DeviceStats::b_bad_epoch) */,
                     u32* a_ro_copy = nullptr /* A's row offsets as this call saw them (a later VERIFY compares) */,
                     u32* verdict = nullptr /* != nullptr: VERIFY -- compare everything with what is stored, write nothing but
                                               this word (pinned host memory) on a difference: 1, | 2 for a bad column of A */);

// completion ticket of a replayed launch sequence (pinned host word the host spins on)
// (the kernel also copies the statistics block into its pinned mirror, before the ticket)
void launch_done(hipStream_t s, u32* dev_ticket, u32* host_ticket, const DeviceStats* st, DeviceStats* host_mirror);
void launch_ticket(hipStream_t s, u32* dev_ticket, u32* host_ticket);  // the ticket alone
// the inputs the analysis depends on, kept (by every call whose analysis WRITES the arena) and compared (by the verifier of
// a replayed sequence): stages.hip.  b_snap: 3 * b_rows + 1 words
void launch_snapshot_inputs(hipStream_t s, const u32* a_ro, const u32* a_col, u32* a_col_copy, u64 nnz_a, const u32* b_ro,
                            const u32* b_col, u32 b_rows, u32* b_snap);
void launch_verify_inputs(hipStream_t s, const u32* a_ro, const u32* a_ro_copy, u32 m, const u32* a_col,
                          const u32* a_col_copy, u64 nnz_a, const u32* b_ro, const u32* b_col, u32 b_rows, const u32* b_snap,
                          u32* verdict);
void launch_delay(hipStream_t s, u32 us);                               // a wave that idles for `us` microseconds
// B's rows strictly ascending and in range (eager path): bit 2 of *verdict (pinned) on a violation
void launch_validate_b(hipStream_t s, const u32* b_ro, const u32* b_col, u32 b_rows, u32 b_cols, u64 b_nnz, u32* verdict);

// exclusive scan of the row counts into offsets_out (may alias counts; + numeric classification, stats fold,
// ordered scatter of the numeric row records when num_cls != nullptr).  offsets_out is left as it was when a
// check of the call fails (capacity_miss, invalid input, nnz overflow).
void launch_scan(hipStream_t s, const u32* counts, u32* offsets_out, u32 m, const u32* a_ro, const u32* row_ops,
                 const u32* row_col_min, const u32* row_col_max, u8* num_cls, BlockPartial* partials,
                 RowRec* recs, DeviceStats* st, const ClassifyParams& cp, u32 vsize, u64 exact_nnz,
                 DeviceStats* host_mirror = nullptr, u64 expect_g = ~0ull, u32 expect_g_rows = ~0u,
                 const u32* pred_off = nullptr, u32* pred_off_out = nullptr, u32* pred_tile_out = nullptr,
                 bool pred_fold_esc = false, u32* dev_ticket = nullptr, u32* host_ticket = nullptr
                 /* with host_mirror: block 0 mirrors the statistics and stores the ticket as soon as it has folded */);

// What a call leaves behind for a replay of the SAME call to verify instead of recompute (pipeline.hip: Prediction):
// per scan tile the position of its rows in every numeric class list, how many there are, and the products of its
// NUM_G rows.  Layout of one tile: pos[kMaxClasses] | count[kMaxClasses] | g_ops (lo, hi).
constexpr u32 kPredTileWords = 2 * kMaxClasses + 2;
// ... and per analysis block the same for the symbolic class lists (+ the scratch entries of its numeric-first rows)
constexpr u32 kPredBlockWords = 2 * kMaxClasses + 2;

// The scan of a replayed sequence whose every row offset is predicted (pred_off) and whose tile tables are known
// (pred_tile): ONE kernel, no fold over the tiles -- each tile scans its rows from its predicted base, compares
// every fresh offset and its class histogram with the prediction (capacity_miss on any difference; nothing is then
// written by that tile) and writes the row records at the predicted list positions.  The statistics the numeric
// kernels and the host read are those of the predicted call (pred_stats), valid if no tile objects.
void launch_scan_predicted(hipStream_t s, const u32* counts, u32* offsets_out, u32 m, const u32* a_ro,
                           const u32* row_ops, const u32* row_col_min, const u32* row_col_max, RowRec* recs,
                           DeviceStats* st, const ClassifyParams& cp, const u32* pred_off, const u32* pred_tile,
                           const DeviceStats* pred_stats, BlockPartial* analysis_partials = nullptr,
                           bool totals_from_pred = false);

// Global-memory buffers of the NUM_G spill path (numeric.hip): per-row plan, per-bucket counters,
// and two product pools (expanded by column bucket; reduced and sorted).
constexpr u32 kGCellsPerBucket = 128;  // fine column cells per wanted bucket (plan kernel and host pool sizing): a
                                       //   cell cannot be split, so coarse cells over clustered columns make oversized buckets
constexpr u32 kGBucketTarget = 1024;  // products per bucket aimed at (skewed columns exceed it)
struct GRowPlan {
    u64 pbase;        // first pool slot of the row's products
    u32 bbase, nb;    // first bucket, number of buckets
    u32 fbase, nf;    // first cell of the fine column grid, number of cells
    u32 shift, unit;  // cell = (col - cmin) >> shift; bucket of a cell = products before it / unit
};
struct SpillBuffers {
    GRowPlan* plan;                    // one per NUM_G row, class-list order
    u32* fcount;                       // per cell: products   (fcount | bcount | bcursor | dcount are
    u32 *bcount, *bcursor, *dcount;    // per bucket: products, scatter cursor, distinct columns
    u32* big_count;                    //   length of big_list;  contiguous: one memset)
    u64* big_list;                     // (row << 32 | bucket) of the buckets beyond the small table
    u32* fmap;                         // per cell: its bucket
    u64* bstart;                       // per bucket: first pool slot
    u32 *clo, *chi;                    // per bucket: column span
    u32* pcol[2];
    void* pval[2];
    u32 bucket_cap, cell_cap;
};

// Everything a symbolic / numeric kernel needs besides the matrices.
struct RowWork {
    const RowRec* recs;     // row records grouped by class (device)
    const DeviceStats* st;  // offsets/counts live here (device)
    const uint2* b_sl;      // per A entry (relative to the first entry of the A view): (start, length) of the
                            //   referenced B row, written by the analysis
    SpillBuffers spill;     // NUM_G class (all null when no row needs it)
    const u64* nf_off;      // numeric-first rows (SYM_NF / NUM_NFCOPY): scratch slot of row r = nf_col/nf_val + nf_off[r]
    u32* nf_col;
    void* nf_val;
    u64 nf_cap;             // entries the pool holds (a slot never ends beyond it)
    // Replayed sequence only: the numeric-first kernel writes a finished row STRAIGHT to its place in
    // C -- at the row offset of the previous identical call (pred_off, the config's own copy) -- when the row's fresh
    // nnz is what that call found; the scan of this call recomputes every offset and rejects the replay if one
    // differs.  No scratch slot, no copy kernel.  All null on the eager path.
    const u32* nf_pred_off;
    u32* nf_direct_col;
    void* nf_direct_val;
    uint2* w_sl;            // per A entry: (start, length) of its B row INSIDE the current column window
                            //   (multi-window rows, row_groups.hpp WindowCursors); a row's entries belong to its workgroup
    u32 xcd_aware;          // class lists are walked in per-XCD contiguous slices (row_groups.hpp)
    // Eager call: the scan staged the row offsets of C in scratch (C.row_offsets -- possibly the caller's reused buffer --
    // is written only once nothing can fail any more); extra workgroups of the numeric light launch move them to C
    // instead of a copy of its own in front of the launch (-8 us).  off_n = 0: nothing to move.
    const u32* off_src;
    u32* off_dst;
    u32 off_n;
    // Replayed sequence WITHOUT a scan kernel (ReplayPlan::skip_scan): every kernel that produces a row's nnz compares it
    // with the room the previous identical call gave the row (nf_pred_off) -- if every row keeps its length, every offset,
    // class and record of the numeric phase is what that call's scan left in the arena.
    u32 verify_counts;
    // ... and WITHOUT a symbolic pass for the rows of the hash / dense classes of the numeric light launch
    // (ReplayPlan::num_verify): their numeric bodies check the nnz themselves (numeric.hip, VERIFY) -- one walk per row.
    u32 verify_numeric;
};

#ifdef __HIPCC__
// nnz of a row of C, as a symbolic kernel found it (see RowWork::verify_counts)
__device__ __forceinline__ void store_row_count(const RowWork& w, u32* __restrict__ counts, u32 row, u32 cnt)
{
    counts[row] = cnt;
    if (w.verify_counts && cnt != w.nf_pred_off[row + 1] - w.nf_pred_off[row])
        const_cast<DeviceStats*>(w.st)->capacity_miss = 1;
}
#endif

// Block ranges of the classes inside a merged ("light") launch: class slot k owns the blocks
// [first[k], first[k+1]).
// Host-known position of a class list inside the record array (exact counts of the previous identical call, or
// of a read-back of this one): lets a workgroup request its first record without waiting for the device-side
// table -- the table is still what decides (a stale hint only costs the second request).  cnt = ~0: unknown.
struct ClassHint {
    u32 off, cnt;
};
constexpr ClassHint kNoHint{0xFFFFFFFFu, 0xFFFFFFFFu};
// (Round 4: the merged kernels no longer carry a ClassHint per class -- measured +-0 when they were added, and with ten
//  class bodies in one kernel their 24 kernel-argument words were what pushed the fused light launch into spilling 90
//  scalar registers: 50.4 -> 43.7 us on the scircuit stand-in without them.  The stand-alone class kernels never had any.)
struct ClassGrid {
    u32 first[13];
};
constexpr u32 kSymLightMask = (1u << SYM_BM1) | (1u << SYM_B4K) | (1u << SYM_W1K) | (1u << SYM_W256) | (1u << SYM_G16) |
                              (1u << SYM_G8) | (1u << SYM_W128) | (1u << SYM_R32) | (1u << SYM_R64) | (1u << SYM_G4);
constexpr u32 kNumLightMask = (1u << NUM_D1) | (1u << NUM_B2K) | (1u << NUM_W512) | (1u << NUM_W256) | (1u << NUM_W128) |
                              (1u << NUM_G16) | (1u << NUM_G8) | (1u << NUM_DIRECT) | (1u << NUM_R32) | (1u << NUM_R64) |
                              (1u << NUM_G4);
// the register classes (esc.hpp, esc_wide.hpp): their rows are finished in the symbolic phase of a fused replay
constexpr u32 kNumEscMask = (1u << NUM_G4) | (1u << NUM_G8) | (1u << NUM_G16) | (1u << NUM_R32) | (1u << NUM_R64);
constexpr u32 kSymEscMask = (1u << SYM_G4) | (1u << SYM_G8) | (1u << SYM_G16) | (1u << SYM_R32) | (1u << SYM_R64);

// One launch for all 256-thread classes in `mask`.  counts_hint[cls] sizes each class' block range
// (the kernels read the real counts on the device and stride, so a stale hint only costs speed).
// `exact`: counts_hint holds the exact rows of EVERY class (the kernels then get ClassHints).
void launch_symbolic_light(hipStream_t s, const u32* counts_hint, u32 mask, const u32* a_ro,
                           const uint2* b_sl, const u32* b_col, const RowWork& w,
                           u32* counts, int cu_count, bool exact = false, u32 fused_vsize = 0,
                           const void* a_val = nullptr, const void* b_val = nullptr, hipEvent_t e0 = nullptr,
                           hipEvent_t e1 = nullptr);
template <typename T>
void launch_numeric_light(hipStream_t s, const u32* counts_hint, u32 mask, const CsrView<T>& A,
                          const CsrView<T>& B, const RowWork& w, u32* c_col, T* c_val, int cu_count,
                          bool exact = false, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);

// Launch the symbolic kernel of class `cls`.  `count` is an UPPER BOUND of the class' row count
// (the rows of A): the grid depends only on it, the kernels read the real count from the
// device-side stats block, so the launch sequence is static and can be captured in a hipGraph.
void launch_symbolic(hipStream_t s, int cls, u32 count, const u32* a_ro, const uint2* b_sl,
                     const u32* b_col, const RowWork& w, u32* counts, int cu_count);

// SYM_NF: the dense-window numeric kernel in the symbolic phase (rows to their scratch slots, nnz to `counts`)
template <typename T>
void launch_numeric_first(hipStream_t s, u32 count, const CsrView<T>& A, const CsrView<T>& B, const RowWork& w,
                          u32* counts, int cu_count, u32 wcols, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);

// The three launches that carry most bytes (the two light launches, the numeric-first one) can be timed EXACTLY when
// the caller profiles: with a pair of events the launch goes through hipExtLaunchKernelGGL, which stamps them with the
// kernel's own begin and end -- what rocprofv3 reports -- instead of bracketing the launch with two event records
// (which adds the dispatch latency, ~4 us, to every duration).
#define SPECK_LAUNCH_TIMED(kernel_, grid_, block_, lds_, stream_, e0_, e1_, ...)                            \
    do {                                                                                                   \
        if (e0_) hipExtLaunchKernelGGL(kernel_, grid_, block_, lds_, stream_, e0_, e1_, 0, __VA_ARGS__);   \
        else hipLaunchKernelGGL(kernel_, grid_, block_, lds_, stream_, __VA_ARGS__);                       \
    } while (0)

// Launch the numeric kernel of class `cls` (same convention for `count`).
template <typename T>
void launch_numeric(hipStream_t s, int cls, u32 count, const CsrView<T>& A, const CsrView<T>& B,
                    const RowWork& w, u32* c_col, T* c_val, int cu_count);

u32 grid_for(u32 count, u32 lds, int threads, int cu_count, u32 rows_per_block);
// resident-set multiples a class grid may reach before its workgroups start striding over rows
void set_grid_rounds(u32 block_classes, u32 subwave_classes);
void set_spill_big_grid(u32 blocks);  // workgroups of the NUM_G launch that reduces the oversized buckets
void set_tiny_threads(int threads);  // workgroup size of the merged small-row numeric launch (64 / 128 / 256)

// LDS bytes a class needs (for occupancy-aware grid sizing and DESIGN.md tables)
u32 symbolic_lds_bytes(int cls);
u32 numeric_lds_bytes(int cls, u32 vsize);

}  // namespace speck
