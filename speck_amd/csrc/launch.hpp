// launch.hpp -- host-side launcher declarations (one per pipeline stage).
#pragma once
#include <hip/hip_ext.h>

#include "chain.hpp"
#include "device_common.hpp"

namespace speck {

u32 analysis_blocks(u32 m);
void set_analysis_wide_rows(u32 avg_len);  // analysis: 64 rows per wave when nnz(A) / rows(A) <= avg_len (0: never)
u32 scan_tiles(u32 m);

// ---- class lists: RECORDS at fixed places ------------------------------------------------------------------
// One buffer of kListRegions x rows(A) records per phase.  Class c lives in region c / 2; the even class of a pair grows
// up from the region's first record, the odd one down from its last -- together they hold at most rows(A) records.  So a
// producer places a row knowing only how many rows of the class come BEFORE it (chain.hpp), and a class kernel reads the
// record of list entry i with ONE load, from an address it knows before it has read a single device word.
// (First form of round 5: row ids in the lists, records in row order -- 4 bytes per list entry instead of 32, but every
//  row of every class kernel paid a dependent load: numeric light launch 52 -> 57 us on the scircuit stand-in, the
//  669 k single-entry rows of the webbase stand-in +5 % on the whole multiply.)
// Arena cost: kListRegions x rows(A) x 32 B per phase (two phases) = 448 B per row of A with the 14 classes of either
// phase (seven regions; round 5 sized them for kMaxClasses = 16: an eighth region nobody addressed).
constexpr u32 kListRegions = (((u32)SYM_CLASSES > (u32)NUM_CLASSES ? (u32)SYM_CLASSES : (u32)NUM_CLASSES) + 1u) / 2u;
__host__ __device__ inline size_t class_list_records(u32 m) { return size_t(kListRegions) * (m ? m : 1u); }
__host__ __device__ __forceinline__ RowRec* class_rec_at(RowRec* lists, u32 m, u32 cls, u32 i)
{
    RowRec* region = lists + size_t(cls >> 1) * m;
    return (cls & 1u) ? region + (m - 1u - i) : region + i;
}
__host__ __device__ __forceinline__ const RowRec* class_rec_at(const RowRec* lists, u32 m, u32 cls, u32 i)
{
    return class_rec_at(const_cast<RowRec*>(lists), m, cls, i);
}

// analysis + symbolic binning (ONE kernel, stages.hip).  sym_cls == nullptr: the per-row quantities and the totals only.
// verdict != nullptr: VERIFY -- compare everything with what is stored, write nothing but this word (pinned host
// memory) on a difference: 1, | 2 for a bad column of A.
void launch_analysis(hipStream_t s, const u32* a_ro, const u32* a_col, const u32* b_ro, const u32* b_col, u32 m, u64 nnz_a,
                     u32* row_ops, u32* row_max_ops, u32* row_col_min, u32* row_col_max, u8* sym_cls, u32* counts,
                     RowRec* sym_recs, DeviceStats* st, const ClassifyParams& cp, uint2* b_sl, const Chain& chain,
                     u64* nf_off = nullptr, u64 expect_nf = ~0ull, u32 b_rows = ~0u,
                     u32* a_ro_copy = nullptr /* A's row offsets as this call saw them (a later VERIFY compares) */,
                     u32* verdict = nullptr, u64* bytes_acc = nullptr /* [2][kMaxClasses], with cp.want_bytes */,
                     u64 b_nnz = ~0ull /* entries of B: every B-row bound is clamped to them */);

// completion ticket of a launch sequence (pinned host word the host spins on)
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
// B's rows strictly ascending and in range (eager path): bit 2 of *verdict (pinned) on a violation
void launch_validate_b(hipStream_t s, const u32* b_ro, const u32* b_col, u32 b_rows, u32 b_cols, u64 b_nnz, u32* verdict);
// staged row offsets -> C.row_offsets when no numeric light launch carries them (not if the sequence was declared void)
void launch_copy_offsets(hipStream_t s, const u32* src, u32* dst, u32 n, const DeviceStats* st);

// exclusive scan of the row counts into offsets_out + numeric classification, records and class lists (ONE kernel;
// num_recs == nullptr: the offsets and nnz(C) only).  pred_off: the offsets a replayed sequence has placed rows by -- every
// fresh offset must agree (capacity_miss).  host_mirror: the last tile mirrors the statistics and stores the ticket.
void launch_scan(hipStream_t s, const u32* counts, u32* offsets_out, u32 m, const u32* a_ro, const u32* row_ops,
                 const u32* row_col_min, const u32* row_col_max, RowRec* num_recs, DeviceStats* st,
                 const ClassifyParams& cp, u32 vsize, u64 exact_nnz, const Chain& chain, DeviceStats* host_mirror = nullptr,
                 u64 expect_g = ~0ull, u32 expect_g_rows = ~0u, const u32* pred_off = nullptr, u32* pred_off_out = nullptr,
                 u32* dev_ticket = nullptr, u32* host_ticket = nullptr, u64* bytes_acc = nullptr,
                 const u32* gate = nullptr /* verdict word of the input check: a violation (bit 2) voids what follows */,
                 u32 gate_ticket = 0 /* != 0: ... and so does a check that has not stored this ticket at gate[16] yet */);

template <typename T>
struct ProductSrc;  // row_groups.hpp

// ---- the one-walk kernel (walk.hip): scan + numeric binning + the numeric walk of the register-class rows, one launch
struct WalkArgs {
    const u32* a_ro;                                   // A.row_offsets (may be a row-range view: absolute offsets)
    const u32 *row_ops, *row_col_min, *row_col_max;    // per row, from the analysis
    const u8* cls_sym;                                 // symbolic class of every row (analysis)
    const u32* counts;                                 // nnz of the rows OTHER kernels counted before this launch
    const u64* nf_off;                                 // slot of every staged row in the scratch pool (analysis, one_walk)
    u32 *offsets_out, *pred_off_out;                   // row offsets of C (staged in scratch) / the config's own copy
    RowRec* recs;                                      // numeric class lists (as the scan kernel leaves them)
    DeviceStats* st;
    u32* c_col;                                        // the caller's C buffers and the entries they hold
    void* c_val;
    u64 c_cap;
    u32* pool_col;                                     // scratch pool: column ids | values, pool_cap entries each
    void* pool_val;
    u64 pool_cap;
    u32 m, tile_rows, vsize;
    ClassifyParams cp;
    u64 expect_g;
    u32 expect_g_rows;
    u64* bytes_acc;
    u32 debug;                                         // development switches (set_walk_debug)
};
void set_walk_debug(u32 tile_rows, u32 flags);
u32 walk_tile_rows(u32 m);
u32 walk_max_rows();  // rows(A) up to which the one-walk call exists (the chain holds kChainMaxBlocks tiles)

// ---- the one-walk kernel of the HASH classes (numeric.hip, walk_hash_kernel; chain3.hpp): eight rows per workgroup,
// accumulated in sub-wave LDS tables sized from the product bound, placed through a three-level chain, sorted and stored
struct WalkHashArgs {
    const u32* a_ro;
    const u32 *row_ops, *row_col_min, *row_col_max;
    u32 *offsets_out, *pred_off_out;
    DeviceStats* st;
    u32* c_col;
    void* c_val;
    u64 c_cap;
    u32 m, max_ops, want_bytes, vsize;
    u64* bytes_acc;
    u32 debug;
};
struct Chain3;
template <typename T>
void launch_walk_hash(hipStream_t s, const WalkHashArgs& args, const ProductSrc<T>& src, const Chain3& chain, hipEvent_t e0 = nullptr,
                      hipEvent_t e1 = nullptr);
constexpr u32 kWalkHashRows = 8;  // rows per workgroup (two per wave)

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
    const RowRec* recs;     // class lists of this phase: records (class_rec_at above)
    u32 m;                  // rows(A): extent of a list region
    const DeviceStats* st;  // class counts live here (device)
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
    // The scan staged the row offsets of C in scratch (C.row_offsets -- possibly the caller's reused buffer -- is written
    // only once nothing can fail any more); extra workgroups of the numeric light launch move them to C.  off_n = 0: nothing to move.
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
    u32 sliced;             // the NUM_B8K rows go through num_sliced_kernel (ClassifyParams::slice_ops != 0)
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

// Block ranges of the classes inside a merged ("light") launch: class slot k owns the blocks [first[k], first[k+1]);
// cnt[k] = the rows the host expects in the slot's class (exact when the structure is what it was at the last read-back /
// in the previous call; ~0: unknown): a workgroup requests its first list entries by it BEFORE the device-side class
// table has arrived -- the table decides, a stale hint costs one more round trip.
struct ClassGrid {
    u32 first[13];
    u32 cnt[12];
};
constexpr u32 kNoCount = 0xFFFFFFFFu;
constexpr u32 kSymLightMask = (1u << SYM_BM1) | (1u << SYM_B4K) | (1u << SYM_W1K) | (1u << SYM_W256) | (1u << SYM_G16) |
                              (1u << SYM_G8) | (1u << SYM_W128) | (1u << SYM_R32) | (1u << SYM_R64);
constexpr u32 kNumLightMask = (1u << NUM_D1) | (1u << NUM_B2K) | (1u << NUM_W512) | (1u << NUM_W256) | (1u << NUM_W128) |
                              (1u << NUM_G16) | (1u << NUM_G8) | (1u << NUM_DIRECT) | (1u << NUM_R32) | (1u << NUM_R64);
// the register classes (esc.hpp, esc_wide.hpp): their rows are finished in the symbolic phase of a fused replay
constexpr u32 kNumEscMask = (1u << NUM_G8) | (1u << NUM_G16) | (1u << NUM_R32) | (1u << NUM_R64);
constexpr u32 kSymEscMask = (1u << SYM_G8) | (1u << SYM_G16) | (1u << SYM_R32) | (1u << SYM_R64);

// One launch for all 256-thread classes in `mask`.  counts_hint[cls] sizes each class' block range
// (the kernels read the real counts on the device and stride, so a stale hint only costs speed).
// `exact`: counts_hint holds the rows the host expects in EVERY class (ClassGrid::cnt).
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
// device-side stats block, so the launch sequence is static.
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
// A launch that FAILS (a kernel whose LDS opt-in was refused, an invalid configuration) must not go unnoticed: the
// runtime keeps only the status of the LAST call, so a failed launch followed by a successful event record is gone by
// the time a stage asks hipGetLastError() -- round 6 found a sanitizer build whose >64 KiB-LDS kernels never ran and whose
// multiply returned "ok" with nnz(C) = 0.  Every launch of the library goes through SPECK_LAUNCH: the status is read at
// once and latched per thread; the pipeline collects it behind every batch (take_launch_error) -> SPECK_ERR_HIP.
void note_launch_status(hipError_t e, const char* what);
int take_launch_error();  // the first failure since the last call (0: none); clears the latch
#define SPECK_LAUNCH(kernel_, ...)                                          \
    do {                                                                    \
        hipLaunchKernelGGL(kernel_, __VA_ARGS__);                           \
        ::speck::note_launch_status(hipGetLastError(), #kernel_);           \
    } while (0)
#define SPECK_LAUNCH_TIMED(kernel_, grid_, block_, lds_, stream_, e0_, e1_, ...)                            \
    do {                                                                                                   \
        if (e0_) hipExtLaunchKernelGGL(kernel_, grid_, block_, lds_, stream_, e0_, e1_, 0, __VA_ARGS__);   \
        else hipLaunchKernelGGL(kernel_, grid_, block_, lds_, stream_, __VA_ARGS__);                       \
        ::speck::note_launch_status(hipGetLastError(), #kernel_);                                          \
    } while (0)

// Launch the numeric kernel of class `cls` (same convention for `count`).
template <typename T>
void launch_numeric(hipStream_t s, int cls, u32 count, const CsrView<T>& A, const CsrView<T>& B,
                    const RowWork& w, u32* c_col, T* c_val, int cu_count);

template <typename T>
void launch_walk(hipStream_t s, const WalkArgs& args, const ProductSrc<T>& src, const Chain& chain, hipEvent_t e0 = nullptr,
                 hipEvent_t e1 = nullptr);

u32 grid_for(u32 count, u32 lds, int threads, int cu_count, u32 rows_per_block);
// resident-set multiples a class grid may reach before its workgroups start striding over rows
void set_grid_rounds(u32 block_classes, u32 subwave_classes);
void set_b8k_full_first(u32 mask);  // NUM_B8K: the full-table launch in front of the half-table one (bit 0 complete calls, bit 1 reuse)
void set_scan_small_items(int items);
void set_spill_big_grid(u32 blocks);  // workgroups of the NUM_G launch that reduces the oversized buckets

// LDS bytes a class needs (for occupancy-aware grid sizing and DESIGN.md tables)
u32 symbolic_lds_bytes(int cls);
u32 numeric_lds_bytes(int cls, u32 vsize);

}  // namespace speck
