// device_common.hpp -- shared definitions for the gfx950 SpGEMM kernels.
// wave64 everywhere: a "wave" below is 64 lanes, ballots are 64-bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace speck {

using u32 = uint32_t;
using u64 = uint64_t;
using u8 = uint8_t;

constexpr u32 kEmptyKey = 0xFFFFFFFFu;
constexpr int kMaxClasses = 16;  // array extent of every per-class table

// Trivially-copyable CSR view handed to kernels (reference: dCSRNoDealloc<T>,
// include/dCSR.h:24-35).  row_offsets may be absolute offsets of a row-range view.
template <typename T>
struct CsrView {
    const u32* __restrict__ row_offsets;
    const u32* __restrict__ col_ids;
    const T* __restrict__ data;
    u32 rows, cols;
};

// ---- kernel classes ("bins") --------------------------------------------------
// A row is processed by a GROUP of lanes: 16 lanes of a wave, a whole 64-lane wave, or a
// workgroup.  Small rows dominate SuiteSparse-like inputs and every row is a chain of four
// dependent global loads (A.rowptr -> A.col -> B.rowptr -> B.col/val), so throughput is
// "rows in flight / chain latency": small rows get small groups.
// Symbolic classes: chosen from the analysis pass' per-row upper bound `ops` (= exact
// product count), the A-row length and the reachable column range.
enum SymClass : u8 {
    SYM_G16 = 0,    // 16 lanes per row, 64-key LDS set       (ops <= 51)
    SYM_W256 = 1,   // one wave per row, 256-key set          (ops <= 204)
    SYM_W1K = 2,    // one wave per row, 1024-key set         (ops <= 819)
    SYM_B4K = 3,    // workgroup(256) per row, 4096-key set   (ops <= 3276)
    SYM_B16K = 4,   // workgroup(512) per row, 16384-key set  (ops <= 13107)
    SYM_B32K = 5,   // workgroup(1024) per row, 32768-key set (ops <= 26214), 128 KiB LDS
    SYM_BM1 = 6,    // column bitmap, workgroup(256), 128 Ki columns per window, rows up to 256 Ki columns
    SYM_BM2 = 7,    // column bitmap, workgroup(1024), 1 Mi columns per window, multi-window
    SYM_NF = 8,     // NUMERIC-FIRST: narrow column range -> the dense-window numeric kernel runs in the
                    //   symbolic phase, writes the finished row to a scratch slot and counts it; no
                    //   symbolic walk at all (numeric phase: NUM_NFCOPY moves the row to its place in C)
    SYM_GH = 9,     // key set in GLOBAL memory (one workgroup(1024) per row, table in the scratch pool): rows wider
                    //   than one SYM_BM2 window with few products per window -- every window of the bitmap costs a
                    //   fixed ~8 us, a global compare-and-swap a fraction of a nanosecond at 4096 in flight
    SYM_G8 = 10,    // 8 lanes per row (8 rows per wave), 32-key LDS set (ops <= 25)
    SYM_W128 = 11,  // 16 lanes per row (four rows per wave), 128-key LDS set (ops <= 102)
    SYM_R32 = 12,   // 32 lanes per row (two rows per wave): <= 128 products from <= 32 entries of A, columns sorted in
                    //   registers (esc_wide.hpp) -- no key set
    SYM_R64 = 13,   // a wave per row: <= 256 products from <= 64 entries of A, sorted in registers
    SYM_CLASSES = 14,
    SYM_NONE = 0xFF  // resolved by the analysis kernel itself (empty / single-entry A rows)
};
// Numeric classes: chosen from the EXACT nnz of the C row (symbolic result).
enum NumClass : u8 {
    NUM_DIRECT = 0,  // A row has one entry: C row = a * B row (already sorted); 16 lanes/row
    NUM_G16 = 1,     // 16 lanes per row, 64-entry table, rank sort        (nnz <= 42)
    NUM_W128 = 2,    // wave per row, 128-entry table, rank sort           (nnz <= 85)
    NUM_W512 = 3,    // wave per row, 512-entry table, 2-level bitmap sort (nnz <= 341)
    NUM_B2K = 4,     // workgroup(256), 2048-entry table, bitmap sort      (nnz <= 1365)
    NUM_B8K = 5,     // workgroup(512), 8192-entry table, bitmap sort      (nnz <= 5461)
    NUM_D1 = 6,      // dense column-window accumulator, narrow column range, workgroup(256)
    NUM_D2 = 7,      // dense accumulator, 16 Ki columns/window, multi-window, workgroup(1024)
    NUM_G = 8,       // global-memory hash spill (heavy rows with a very wide column range)
    NUM_W256 = 9,    // half a wave per row (two rows per wave), 256-entry table, 2-level bitmap sort (nnz <= 170):
                     //   the lower half of what used to be NUM_W512, at twice the rows in flight per wave
    NUM_NFCOPY = 10, // row already computed by the symbolic phase (SYM_NF): copy scratch slot -> C
    NUM_G8 = 11,     // 8 lanes per row (8 rows per wave), 32-entry table, rank sort   (nnz <= 21): the small-row
                     //   kernel is latency x occupancy bound -- twice the rows in flight per wave
    NUM_R32 = 12,    // 32 lanes per row: <= 128 products from <= 32 entries of A, column range < 2^25 -- expand / sort /
                     //   compress in registers (esc_wide.hpp), whatever the nnz
    NUM_R64 = 13,    // a wave per row: <= 256 products from <= 64 entries, column range < 2^24
    NUM_CLASSES = 14,
    NUM_NONE = 0xFF
};

// Key sets of the sub-wave symbolic classes: TWICE the slots the 4/5 rule of the reference (Multiply.cu:263) would give
// them -- a key costs 4 bytes, the LDS of these launches is nowhere near the limit, and at a load <= 0.4 the probing
// loops (whose length is the longest chain of the wave) mostly do not start (mac_econ stand-in -2.5 %).
constexpr u32 kSymG8Cap = 64, kSymG8MaxOps = 25;
constexpr u32 kSymG16Cap = 128, kSymG16MaxOps = 51;
constexpr u32 kSymW128Cap = 256, kSymW128MaxOps = 102;
constexpr u32 kSymW256Cap = 512, kSymW256MaxOps = 204;
constexpr u32 kSymW1KCap = 1024, kSymW1KMaxOps = 819;
constexpr u32 kSymB4KCap = 4096, kSymB4KMaxOps = 3276;
constexpr u32 kSymB16KCap = 16384, kSymB16KMaxOps = 13107;
constexpr u32 kSymB32KCap = 32768, kSymB32KMaxOps = 26214;
constexpr u32 kSymBm1Words = 4096;    // 16 KiB  -> 131072 columns per window: no more LDS than the hash classes it shares
                                      //   the merged light launch with (a 32 KiB window capped that launch at 4
                                      //   workgroups per CU for a handful of rows); wider rows take two windows
constexpr u32 kSymBm1MaxCols = 262144;  // rows up to this column range are SYM_BM1
constexpr u32 kSymBm2Words = 32768;   // 128 KiB -> 1048576 columns per window

// Highest load of a numeric hash table, in percent: a class takes rows up to this share of its capacity, and a
// row's table is the smallest power of two that keeps it.  Two values, measured (gpurun_out/r3m):
//   * the rank-sort classes (8 / 16 / 32 lanes per row) keep the reference's 2/3 (Multiply.cu:142) -- they are
//     bound by VALU issue, a fuller table means longer probing loops for the whole wave, and a fourth key per
//     lane in the rank sort costs the launch its seventh wave per SIMD;
//     (so do the 32-lane and wave classes with the bitmap sort: at 0.85 their launch lost 10 % on the scircuit
//     and mac_econ stand-ins);
//   * the WORKGROUP classes (NUM_B2K / NUM_B8K and the buckets of NUM_G) go to 0.85: with double hashing the
//     probing stays short, and what limits those launches is LDS capacity x row latency -- a smaller table per
//     row (fewer slots to clear, load and rank) and rows moving to the class with half the LDS pay more
//     (webbase stand-in -5 %).
#ifndef SPECK_LOAD_PCT
#define SPECK_LOAD_PCT 85
#endif
#ifndef SPECK_LOAD_TINY_PCT
#define SPECK_LOAD_TINY_PCT 67
#endif
constexpr u32 max_nnz_of(u32 cap, u32 pct) { return pct == 67 ? cap * 2 / 3 : cap * pct / 100; }
// log2 of the table a row with `nnz` entries gets (before the class clamps it to [group width, capacity])
__host__ __device__ inline u32 table_bits(u32 nnz, u32 pct)
{
    const u32 want = pct == 67 ? nnz + (nnz >> 1) : (u32)((u64(nnz) * 100 + pct - 1) / pct);
    u32 bits = 1;
    while ((1u << bits) < want) ++bits;
    return bits;
}
constexpr u32 kNumG8Cap = 32, kNumG8MaxNnz = max_nnz_of(32, SPECK_LOAD_TINY_PCT);
constexpr u32 kNumEscMaxOps = 32, kNumEscMaxLen = 8;  // NUM_G8 / SYM_G8 = the register-resident classes (esc.hpp)
constexpr u32 kNumEsc16MaxOps = 64, kNumEsc16MaxLen = 16;  // NUM_G16 / SYM_G16: 16 lanes; columns < 2^26 (ClassifyParams::esc16)
// the WIDE register classes (esc_wide.hpp): the sort key packs (column - first reachable column of the row) with the
// product number into 32 bits -- a condition on the row's column RANGE (analysis), not on cols(B)
constexpr u32 kNumEsc32MaxOps = 128, kNumEsc32MaxLen = 32, kNumEsc32RangeBits = 25;
constexpr u32 kNumEsc64MaxOps = 256, kNumEsc64MaxLen = 64, kNumEsc64RangeBits = 24;
constexpr u32 kNumG16Cap = 64, kNumG16MaxNnz = max_nnz_of(64, SPECK_LOAD_TINY_PCT);
constexpr u32 kNumW128Cap = 128, kNumW128MaxNnz = max_nnz_of(128, SPECK_LOAD_TINY_PCT);
constexpr u32 kNumW512Cap = 512, kNumW512MaxNnz = max_nnz_of(512, SPECK_LOAD_TINY_PCT);
constexpr u32 kNumW256Cap = 256, kNumW256MaxNnz = max_nnz_of(256, SPECK_LOAD_TINY_PCT);
constexpr u32 kNumB2KCap = 2048, kNumB2KMaxNnz = max_nnz_of(2048, SPECK_LOAD_PCT);
constexpr u32 kNumB8KCap = 8192, kNumB8KMaxNnz = max_nnz_of(8192, SPECK_LOAD_PCT),
              kNumB8KHalfMaxNnz = max_nnz_of(4096, SPECK_LOAD_PCT);
constexpr u32 kNumD1Cols = 4096;   // rows up to this column range are NUM_D1 / numeric-first
constexpr u32 kNumD1Win = 2560;    // NUM_D1's LDS window (wider rows take two windows): no more LDS than NUM_W512 /
                                   //   NUM_B2K, which share the merged light launch with it (5 instead of 4 workgroups per CU)
constexpr u32 kNumD2Cols = 16384;
// NUM_B8K rows in COLUMN SLICES (numeric.hip, num_sliced_body; ClassifyParams::slice_ops): the row's products are counted
// per bin of a column histogram that lives where the 2 Ki table's accumulators will, the bins are cut into slices that
// cannot hold more distinct columns than that table takes, and the slices are accumulated / sorted / stored one after
// the other -- a heavy row holds 30 KiB of LDS and four waves instead of 61-106 KiB and eight.
constexpr u32 kSliceBins = 4096;      // bins of the histogram (= 2 x kNumB2KCap words)
constexpr u32 kSliceMaxWidth = 512;   // columns per bin at most
constexpr u32 kSliceMaxCols = kSliceBins * kSliceMaxWidth;  // cols(B) up to 2 Mi
constexpr u32 kSliceMax = 48;         // slices of a row at most
constexpr u32 kSliceMaxOps = 49152;   // ... which a row of this many products cannot exceed (a slice takes >= 1229 of them)

// Tunables that travel to the classifying kernels.
struct ClassifyParams {
    u32 sym_bitmap_ratio;   // use a bitmap when range <= ratio * ops (and ops > wave limit)
    u32 num_dense_ratio;    // use D1 when range <= kNumD1Cols and range <= ratio * nnz
    u32 num_global_passes;  // use the global-hash spill when dense windows would exceed this
    u32 num_w256;           // rows of 86..170 nnz: 32 lanes per row (else they join NUM_W512)
    u32 esc16;              // cols(B) <= 2^26: the 16-lane register class may pack (column, product number) into 32 bits
    u32 esc32, esc64;       // the wide register classes (32 / 64 lanes per row, 128 / 256 products)
    u32 esc_fused;          // replayed sequence with direct placement: the rows of the register classes are finished
                            //   in the symbolic phase (esc_rows.hpp) -- the numeric phase only accounts for them
    u32 num_g8;             // rows of <= kNumG8MaxNnz entries: 8 lanes per row (else they join NUM_G16)
    u32 sym_g8;             // rows of <= kSymG8MaxOps products: 8 lanes per row (else they join SYM_G16)
    u32 sym_w128;           // rows of 52..102 products: 16 lanes per row (else they join SYM_W256)
    u32 nf_min_ops;         // numeric-first (SYM_NF) for rows with range <= kNumD1Cols and at least this many
                            //   products; 0 = off
    u32 gh_per_window;      // SYM_GH instead of a multi-window SYM_BM2 when the row holds fewer products than this
                            //   per bitmap window; 0 = off
    u32 want_bytes;         // accumulate the per-class algorithmic byte counts (profiling)
    u32 one_walk;           // one-walk call (walk.hip): the rows of the register classes take a slot of the scratch pool
    u32 slice_ops;          // NUM_B8K rows are produced in column slices (num_sliced_body): rows of up to this many
                            //   products are NUM_B8K (kSliceMaxOps; 0 = off: the two workgroup(512) launches)
    u32 sym_allowed;        // classes whose kernels are part of this launch sequence; a row
    u32 num_allowed;        //   outside them raises DeviceStats::capacity_miss (graph replay)
};

// Numeric-first rows: the reachable column range fits ONE dense window, so the numeric kernel needs no
// nnz to size anything -- it can run before the scan.  Same predicate in both phases.
__host__ __device__ inline bool is_numeric_first(u32 len_a, u32 ops, u32 cmin, u32 cmax, const ClassifyParams& p)
{
    return p.nf_min_ops != 0 && len_a > 1 && ops >= p.nf_min_ops && u64(cmax) - u64(cmin) + 1 <= kNumD1Cols;
}

// Scratch slot of a numeric-first row: its nnz can exceed neither its column range nor its product count.
__host__ __device__ inline u32 nf_slot_entries(u32 cmin, u32 cmax, u32 ops)
{
    const u32 range = cmax - cmin + 1u;
    return range < ops ? range : ops;
}

// Slots of a SYM_GH row's key set: a power of two, load <= 1/2.  The classifier keeps ops < 2^22 there.
constexpr u32 kSymGhMinSlots = 65536, kSymGhMaxOps = 1u << 22;
__host__ __device__ inline u32 gh_table_slots(u32 ops)
{
    const u32 want = 2u * (ops < kSymGhMaxOps ? ops : kSymGhMaxOps);
    u32 slots = kSymGhMinSlots;
    while (slots < want) slots <<= 1;
    return slots;
}

// Rows of the wide register classes: the same predicate in both phases (a fused row must be one in BOTH).
__host__ __device__ inline bool is_esc32(u32 len_a, u32 ops, u32 cmin, u32 cmax, const ClassifyParams& p)
{
    return p.esc32 && ops <= kNumEsc32MaxOps && len_a <= kNumEsc32MaxLen && ((cmax - cmin) >> kNumEsc32RangeBits) == 0;
}
__host__ __device__ inline bool is_esc64(u32 len_a, u32 ops, u32 cmin, u32 cmax, const ClassifyParams& p)
{
    return p.esc64 && ops <= kNumEsc64MaxOps && len_a <= kNumEsc64MaxLen && ((cmax - cmin) >> kNumEsc64RangeBits) == 0;
}

__host__ __device__ inline u8 classify_symbolic(u32 len_a, u32 ops, u32 cmin, u32 cmax,
                                                const ClassifyParams& p)
{
    if (ops == 0 || len_a <= 1) return SYM_NONE;
    if (is_numeric_first(len_a, ops, cmin, cmax, p)) return SYM_NF;
    // at most 32 products from at most 8 entries of A: sorted in registers (esc.hpp)
    if (p.sym_g8 && ops <= kNumEscMaxOps && len_a <= kNumEscMaxLen) return SYM_G8;
    if (p.esc16 && ops <= kNumEsc16MaxOps && len_a <= kNumEsc16MaxLen) return SYM_G16;  // 16 lanes, 64 products
    if (is_esc32(len_a, ops, cmin, cmax, p)) return SYM_R32;                             // 32 lanes, 128 products
    if (is_esc64(len_a, ops, cmin, cmax, p)) return SYM_R64;                             // a wave, 256 products
    if (p.sym_w128 && ops <= kSymW128MaxOps) return SYM_W128;
    if (ops <= kSymW256MaxOps) return SYM_W256;
    const u64 range = u64(cmax) - u64(cmin) + 1;
    const bool bitmap_ok = range <= u64(p.sym_bitmap_ratio) * ops;
    if (bitmap_ok && range <= kSymBm1MaxCols) return SYM_BM1;
    if (ops <= kSymW1KMaxOps) return SYM_W1K;
    if (ops <= kSymB4KMaxOps) return SYM_B4K;
    // heavier rows: one bitmap window of the 256-thread class costs 32 words per thread to clear
    // and count -- nothing against >= 3277 products -- and keeps the row in the merged launch
    if (range <= kSymBm1MaxCols) return SYM_BM1;
    if (ops <= kSymB16KMaxOps) return SYM_B16K;
    constexpr u64 kWin = u64(kSymBm2Words) * 32;
    const bool sparse_wide = p.gh_per_window != 0 && range > kWin && ops < kSymGhMaxOps &&
                             u64(ops) < (range + kWin - 1) / kWin * p.gh_per_window;
    if (bitmap_ok && !sparse_wide) return SYM_BM2;
    if (ops <= kSymB32KMaxOps) return SYM_B32K;
    return sparse_wide ? SYM_GH : SYM_BM2;
}

__host__ __device__ inline u8 classify_numeric(u32 len_a, u32 ops, u32 nnz, u32 cmin, u32 cmax,
                                               const ClassifyParams& p)
{
    if (nnz == 0) return NUM_NONE;
    if (len_a == 1) return NUM_DIRECT;
    if (is_numeric_first(len_a, ops, cmin, cmax, p)) return NUM_NFCOPY;
    // at most 32 products from at most 8 entries of A: expand / sort / compress in registers (esc.hpp), whatever
    // the nnz; the hash classes take the rest by nnz
    // (num_g8 and sym_g8 are switched together: a fused row must be a register-class row in BOTH phases)
    if (p.num_g8 && ops <= kNumEscMaxOps && len_a <= kNumEscMaxLen) return p.esc_fused ? NUM_NFCOPY : NUM_G8;
    if (p.esc16 && ops <= kNumEsc16MaxOps && len_a <= kNumEsc16MaxLen) return p.esc_fused ? NUM_NFCOPY : NUM_G16;
    if (is_esc32(len_a, ops, cmin, cmax, p)) return p.esc_fused ? NUM_NFCOPY : NUM_R32;
    if (is_esc64(len_a, ops, cmin, cmax, p)) return p.esc_fused ? NUM_NFCOPY : NUM_R64;
    if (nnz <= kNumW128MaxNnz) return NUM_W128;
    const u64 range = u64(cmax) - u64(cmin) + 1;
    if (range <= kNumD1Cols && range <= u64(p.num_dense_ratio) * nnz) return NUM_D1;
    if (p.num_w256 && nnz <= kNumW256MaxNnz) return NUM_W256;
    if (nnz <= kNumW512MaxNnz) return NUM_W512;
    if (nnz <= kNumB2KMaxNnz) return NUM_B2K;
    if (nnz <= kNumB8KMaxNnz && (p.slice_ops == 0 || ops <= p.slice_ops)) return NUM_B8K;
    const u64 passes = (range + kNumD2Cols - 1) / kNumD2Cols;
    if (passes > p.num_global_passes) return NUM_G;
    return NUM_D2;
}

// Algorithmic byte model per row (restated in DESIGN.md):
//   numeric : 8 [A.rowptr pair] + 12*lenA [A col+val] + 8*lenA [B.rowptr pair]
//             + 12*ops [B col+val per product] + 4 [C.rowptr] + 12*nnz [C col+val]
//   symbolic: 8 + 4*lenA + 8*lenA + 4*ops + 4
__host__ __device__ inline u64 numeric_row_bytes(u32 len_a, u32 ops, u32 nnz, u32 vsize)
{
    return 8ull + u64(4 + vsize) * len_a + 8ull * len_a + u64(4 + vsize) * ops + 4ull +
           u64(4 + vsize) * nnz;
}
__host__ __device__ inline u64 symbolic_row_bytes(u32 len_a, u32 ops)
{
    return 8ull + 12ull * len_a + 4ull * ops + 4ull;
}

// Per-class row list description, shared by the symbolic and the numeric phase.
struct BinTable {
    u32 count[kMaxClasses];
    u32 offset[kMaxClasses + 1];
    u64 bytes[kMaxClasses];
};

// Device-side statistics block; one per config, zeroed at the start of a call.
struct DeviceStats {
    u64 sum_products;
    u64 nnz_c;
    u32 max_row_ops;
    u32 max_row_nnz_c;
    u32 nnz_overflow;
    u32 capacity_miss;       // nnz(C) of this call does not fit the C buffers baked into the launch
    BinTable sym;
    BinTable num;
    u64 g_products;          // products of the NUM_G rows (the host sizes the spill pool from it)
    u64 nf_entries;          // scratch entries of the SYM_NF rows (sum of min(column range, products)) and of the
                             //   SYM_GH rows (slots of their key sets)
    u32 walk_rows;           // rows finished by the one-walk kernel of the call (walk.hip; 0: a two-phase call)
    u32 nf_max_range;        // widest column range among the SYM_NF rows (sizes the LDS window of their kernel)
    u32 a_invalid;           // a column id of A is >= rows(B) (the analysis clamps it, so nothing reads out of bounds)
    u32 chain_error;         // a workgroup of the analysis / scan waited in vain for the workgroups before it (chain.hpp)
    u32 front_miss;          // capacity_miss as the scan FOUND it: raised by the analysis / symbolic side of the call.  0 with
                             //   capacity_miss = 1: only the scan's own checks objected (nnz(C), numeric classes, spill pool,
                             //   input check) -- its offsets, counts and class lists stand (the through call, pipeline.hip)
    u32 pad_;
};
static_assert(sizeof(DeviceStats) % 8 == 0, "mirrored to the host in 8-byte words");

// One row of work as the class kernels see it: ONE 32-byte load per row, from the row's place in its class list
// (launch.hpp, class_rec_at: written by the analysis for the symbolic phase, by the scan for the numeric one); a class
// kernel requests the record of its next row while the current one runs.
struct __attribute__((aligned(32))) RowRec {
    u32 a0, a1;      // bounds of the A row (absolute offsets into A.col_ids / A.data)
    u32 base;        // first entry of the C row (numeric phase)
    u32 nnz;         // nnz of the C row (numeric phase)
    u32 cmin, cmax;  // column range reachable by the row (analysis)
    u32 ops;         // intermediate products of the row
    u32 row;         // row of A / C
};
#ifdef __HIPCC__
// ---- wave64 primitives ---------------------------------------------------------
__device__ __forceinline__ u32 lane_id() { return __lane_id(); }

__device__ __forceinline__ void wave_lds_fence()
{
    // LDS operations of one wave complete in issue order; this only stops the
    // compiler from moving LDS accesses across the point and drains lgkmcnt.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Scans and reductions move data with DPP modifiers (row_shr inside the 16-lane rows, then
// row_bcast15 / row_bcast31 across rows): pure VALU, no ds_bpermute traffic through the LDS
// crossbar -- the class kernels are bound by VALU and LDS instruction issue, not by HBM.
// Lanes without a source keep `old` (the identity of the operation).
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ u32 dpp_move(u32 old, u32 v)
{
    return (u32)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xF, false);
}
constexpr int kDppRowShl = 0x100, kDppRowShr = 0x110, kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;

// inclusive scan inside every 16-lane row
__device__ __forceinline__ u32 row16_inclusive_scan(u32 v)
{
    v += dpp_move<kDppRowShr + 1>(0, v);
    v += dpp_move<kDppRowShr + 2>(0, v);
    v += dpp_move<kDppRowShr + 4>(0, v);
    v += dpp_move<kDppRowShr + 8>(0, v);
    return v;
}
// inclusive scan across the 64 lanes
__device__ __forceinline__ u32 wave_inclusive_scan(u32 v)
{
    v = row16_inclusive_scan(v);
    v += dpp_move<kDppRowBcast15, 0xA>(0, v);
    v += dpp_move<kDppRowBcast31, 0xC>(0, v);
    return v;
}
__device__ __forceinline__ u32 wave_reduce_add(u32 v)
{
    return (u32)__builtin_amdgcn_readlane((int)wave_inclusive_scan(v), 63);
}
__device__ __forceinline__ u64 wave_reduce_add(u64 v)
{
    // per-half sums of the low words cannot overflow a u64 accumulated from 64 lanes
    const u32 lo = (u32)v, hi = (u32)(v >> 32);
    const u64 s_lo16 = wave_reduce_add(lo & 0xFFFFu), s_hi16 = wave_reduce_add(lo >> 16);
    return (u64(wave_reduce_add(hi)) << 32) + (s_hi16 << 16) + s_lo16;
}
__device__ __forceinline__ u32 wave_reduce_max(u32 v)
{
    v = max(v, dpp_move<kDppRowShr + 1>(0, v));
    v = max(v, dpp_move<kDppRowShr + 2>(0, v));
    v = max(v, dpp_move<kDppRowShr + 4>(0, v));
    v = max(v, dpp_move<kDppRowShr + 8>(0, v));
    v = max(v, dpp_move<kDppRowBcast15, 0xA>(0, v));
    v = max(v, dpp_move<kDppRowBcast31, 0xC>(0, v));
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ u32 wave_reduce_min(u32 v)
{
    return ~wave_reduce_max(~v);
}
__device__ __forceinline__ u64 lanemask_lt()
{
    return (1ull << lane_id()) - 1ull;
}

// Multiplicative hash; the table capacity is a power of two, slot = top bits.
template <u32 CAP>
__device__ __forceinline__ u32 hash_slot(u32 key)
{
    static_assert((CAP & (CAP - 1)) == 0, "capacity must be a power of two");
    constexpr int bits = __builtin_ctz(CAP);
    return (key * 0x9E3779B1u) >> (32 - bits);
}

// Block-wide exclusive scan of one u32 per thread (THREADS multiple of 64).
// `warp_sums` is LDS scratch with THREADS/64 + 1 entries. Returns exclusive prefix;
// *total receives the block sum.
template <int THREADS>
__device__ __forceinline__ u32 block_exclusive_scan(u32 v, u32* warp_sums, u32* total)
{
    constexpr int NW = THREADS / 64;
    const u32 lane = lane_id();
    const u32 wid = threadIdx.x >> 6;
    const u32 incl = wave_inclusive_scan(v);
    if (lane == 63) warp_sums[wid] = incl;
    __syncthreads();
    if (wid == 0) {
        u32 s = lane < NW ? warp_sums[lane] : 0;
        const u32 si = wave_inclusive_scan(s);
        if (lane < NW) warp_sums[lane] = si - s;
        if (lane == NW - 1) warp_sums[NW] = si;
    }
    __syncthreads();
    const u32 base = warp_sums[wid];
    *total = warp_sums[NW];
    __syncthreads();
    return base + incl - v;
}
// make POISON=1 (libspeck_amd_poison.so, SPECK_LIB=...): every kernel starts by filling the workgroup's WHOLE LDS allocation
// (static + dynamic: the dispatch packet's group_segment_size) with a word no count, offset or index can be mistaken for.
// LDS keeps what the previous kernel on that CU left in it; a body that reads a word it never wrote works by accident
// until the previous kernel was another one (round 6: the staging entries of waves that had left their workgroup early,
// row_groups.hpp block_void).  With the poison such a read gathers at a wild address or fails its parity test at once.
#ifdef SPECK_POISON_LDS
__device__ __forceinline__ void poison_lds()
{
    const u32 bytes = ((const __attribute__((address_space(4))) u32*)__builtin_amdgcn_dispatch_ptr())[7];  // group_segment_size
    const u32 nt = blockDim.x * blockDim.y * blockDim.z;
    const u32 t = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    for (u32 b = t * 4u; b + 4u <= bytes; b += nt * 4u) asm volatile("ds_write_b32 %0, %1" ::"v"(b), "v"(0xFFFFFFF1u) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
}
#define SPECK_POISON() ::speck::poison_lds()
#else
#define SPECK_POISON()
#endif
#endif  // __HIPCC__

}  // namespace speck
