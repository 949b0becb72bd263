// device_common.hpp -- shared device helpers for the gfx950 SpGEMM kernels.
// wave64 everywhere: a "wave" below is 64 lanes, ballots are 64-bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace speck {

using u32 = uint32_t;
using u64 = uint64_t;
using u8 = uint8_t;

constexpr u32 kEmptyKey = 0xFFFFFFFFu;
constexpr int kWave = 64;

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
// Symbolic classes, chosen from the analysis pass' per-row upper bound `ops`
// (= exact product count), the A-row length and the reachable column range.
enum SymClass : u8 {
    SYM_WAVE = 0,  // one wave per row, 128-key LDS set per wave
    SYM_H1 = 1,    // workgroup hash set, 1024 keys
    SYM_H2 = 2,    // workgroup hash set, 8192 keys
    SYM_H3 = 3,    // workgroup hash set, 32768 keys (128 KiB of LDS)
    SYM_BM1 = 4,   // column bitmap, 256 Ki columns per window
    SYM_BM2 = 5,   // column bitmap, 1 Mi columns per window (128 KiB of LDS), multi-window
    SYM_NONE = 0xFF  // resolved by the analysis kernel itself (empty / single-entry A rows)
};
// Numeric classes, chosen from the EXACT nnz of the C row (symbolic result).
enum NumClass : u8 {
    NUM_DIRECT = 0,  // A row has one entry: C row = a * B row (already sorted)
    NUM_WAVE = 1,    // one wave per row, 128-entry LDS table per wave
    NUM_H1 = 2,      // workgroup hash, 512 entries, in-place rank sort
    NUM_H2 = 3,      // workgroup hash, 2048 entries, bitmap-rank sort
    NUM_H3 = 4,      // workgroup hash, 8192 entries, bitmap-rank sort
    NUM_D1 = 5,      // dense column-window accumulator, narrow column range
    NUM_D2 = 6,      // dense column-window accumulator, 16 Ki columns/window, multi-window
    NUM_G = 7,       // global-memory hash spill (heavy rows, wide column range)
    NUM_NONE = 0xFF
};

constexpr u32 kSymWaveCap = 128, kSymWaveMaxOps = 102;
constexpr u32 kSymH1Cap = 1024, kSymH1MaxOps = 819;
constexpr u32 kSymH2Cap = 8192, kSymH2MaxOps = 6553;
constexpr u32 kSymH3Cap = 32768, kSymH3MaxOps = 26214;
constexpr u32 kSymBm1Words = 8192;    // 32 KiB  -> 262144 columns
constexpr u32 kSymBm2Words = 32768;   // 128 KiB -> 1048576 columns per window

constexpr u32 kNumWaveCap = 128, kNumWaveMaxNnz = 85;
constexpr u32 kNumH1Cap = 512, kNumH1MaxNnz = 341;
constexpr u32 kNumH2Cap = 2048, kNumH2MaxNnz = 1365;
constexpr u32 kNumH3Cap = 8192, kNumH3MaxNnz = 5461;
constexpr u32 kNumD1Cols = 4096;
constexpr u32 kNumD2Cols = 16384;

// Tunables that travel to the classifying kernels.
struct ClassifyParams {
    u32 sym_bitmap_ratio;   // use a bitmap when range <= ratio * ops (and ops > wave limit)
    u32 num_dense_ratio;    // use D1 when range <= kNumD1Cols and range <= ratio * nnz
    u32 num_global_passes;  // use the global-hash spill when dense windows would exceed this
    u32 reserved;
};

__host__ __device__ inline u8 classify_symbolic(u32 len_a, u32 ops, u32 cmin, u32 cmax,
                                                const ClassifyParams& p)
{
    if (ops == 0 || len_a <= 1) return SYM_NONE;
    if (ops <= kSymWaveMaxOps) return SYM_WAVE;
    const u64 range = u64(cmax) - u64(cmin) + 1;
    const bool bitmap_ok = range <= u64(p.sym_bitmap_ratio) * ops;
    if (bitmap_ok && range <= u64(kSymBm1Words) * 32) return SYM_BM1;
    if (ops <= kSymH1MaxOps) return SYM_H1;
    if (ops <= kSymH2MaxOps) return SYM_H2;
    if (bitmap_ok) return SYM_BM2;
    if (ops <= kSymH3MaxOps) return SYM_H3;
    return SYM_BM2;
}

__host__ __device__ inline u8 classify_numeric(u32 len_a, u32 nnz, u32 cmin, u32 cmax,
                                               const ClassifyParams& p)
{
    if (nnz == 0) return NUM_NONE;
    if (len_a == 1) return NUM_DIRECT;
    if (nnz <= kNumWaveMaxNnz) return NUM_WAVE;
    const u64 range = u64(cmax) - u64(cmin) + 1;
    if (range <= kNumD1Cols && range <= u64(p.num_dense_ratio) * nnz) return NUM_D1;
    if (nnz <= kNumH1MaxNnz) return NUM_H1;
    if (nnz <= kNumH2MaxNnz) return NUM_H2;
    if (nnz <= kNumH3MaxNnz) return NUM_H3;
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

// Device-side statistics block; one per config, zeroed at the start of a call.
struct DeviceStats {
    u64 sum_products;
    u64 nnz_c;
    u32 max_row_ops;
    u32 max_row_nnz_c;
    u32 nnz_overflow;
    u32 capacity_miss;   // numeric kernels saw nnz_c > capacity of the reused C buffers
    u32 sym_count[8];
    u32 num_count[8];
    u32 sym_offset[9];
    u32 num_offset[9];
    u32 sym_cursor[8];
    u32 num_cursor[8];
    u64 sym_bytes[8];
    u64 num_bytes[8];
    u32 gmap_next;       // global-hash spill pool cursor
    u32 pad;
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

template <typename V>
__device__ __forceinline__ V wave_reduce_add(V v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ u32 wave_reduce_max(u32 v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, (u32)__shfl_xor((int)v, off, 64));
    return v;
}
__device__ __forceinline__ u32 wave_reduce_min(u32 v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, (u32)__shfl_xor((int)v, off, 64));
    return v;
}
// inclusive scan across the 64 lanes
__device__ __forceinline__ u32 wave_inclusive_scan(u32 v)
{
    const u32 lane = lane_id();
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        u32 t = (u32)__shfl_up((int)v, off, 64);
        if (lane >= (u32)off) v += t;
    }
    return v;
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

// log2 of the lane-group width that walks one B row: the smallest power of two
// >= the average B-row length of this A row, clamped to [min_shift, max_shift].
// (Role of the reference's getThreadShiftNew, include/common.cuh:509-555, re-derived
// for 64-lane waves: a group never spans waves.)
__device__ __forceinline__ u32 pick_group_shift(u32 ops, u32 len_a, u32 min_shift, u32 max_shift)
{
    const u32 avg = (ops + len_a - 1) / (len_a ? len_a : 1);
    u32 s = avg <= 1 ? 0 : 32 - __clz(avg - 1);
    return s < min_shift ? min_shift : (s > max_shift ? max_shift : s);
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
#endif  // __HIPCC__

}  // namespace speck
