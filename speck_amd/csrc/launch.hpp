// launch.hpp -- host-side launcher declarations (one per pipeline stage).
#pragma once
#include "device_common.hpp"

namespace speck {

u32 analysis_blocks(u32 m);
u32 scan_tiles(u32 m);
size_t scan_scratch_bytes(u32 m);

// analysis (+ stats fold + ordered scatter of the symbolic classes when sym_cls != nullptr)
void launch_analysis(hipStream_t s, const u32* a_ro, const u32* a_col, const u32* b_ro,
                     const u32* b_col, u32 m, u64 nnz_a, u32* row_ops, u32* row_max_ops,
                     u32* row_col_min, u32* row_col_max, u8* sym_cls, u32* counts,
                     BlockPartial* partials, u32* blk_base, u32* bin_rows, DeviceStats* st,
                     const ClassifyParams& cp);

// exclusive scan of the row counts (+ numeric classification, stats fold, ordered scatter
// when num_cls != nullptr)
void launch_scan(hipStream_t s, u32* counts_inout, u32 m, u64* tile_sums, const u32* a_ro,
                 const u32* row_ops, const u32* row_col_min, const u32* row_col_max, u8* num_cls,
                 BlockPartial* partials, u32* blk_base, u32* bin_rows, DeviceStats* st,
                 const ClassifyParams& cp, u32 vsize);

// Everything a symbolic / numeric kernel needs besides the matrices.
struct RowWork {
    const u32* bin_rows;     // row ids grouped by class
    const u32* row_ops;      // per-row product count (analysis)
    const u32* row_col_min;  // per-row min reachable column
    const u32* row_col_max;  // per-row max reachable column
    const DeviceStats* st;   // offsets/counts live here (device)
};

// Launch the symbolic kernel of class `cls` over `count` rows (host-known count).
void launch_symbolic(hipStream_t s, int cls, u32 count, const u32* a_ro, const u32* a_col,
                     const u32* b_ro, const u32* b_col, const RowWork& w, u32* counts, int cu_count);

// Launch the numeric kernel of class `cls`.
template <typename T>
void launch_numeric(hipStream_t s, int cls, u32 count, const CsrView<T>& A, const CsrView<T>& B,
                    const RowWork& w, const u32* c_ro, u32* c_col, T* c_val, u64 c_capacity,
                    DeviceStats* st_mut, int cu_count);

u32 grid_for(u32 count, u32 lds, int threads, int cu_count, u32 rows_per_block);

// LDS bytes a class needs (for occupancy-aware grid sizing and DESIGN.md tables)
u32 symbolic_lds_bytes(int cls);
u32 numeric_lds_bytes(int cls, u32 vsize);

}  // namespace speck
