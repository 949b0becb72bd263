// extras.hip -- the two side utilities the reference driver uses around the hot path:
//   * spECK::Compare   (reference source/GPU/Compare.cu:11-82) -- made strict: offsets and
//     column ids bit-exact, values by relative tolerance
//   * transpose for non-square A (reference source/GPU/Transpose.cu:10-117: a column histogram with global
//     atomics, a scan, an unordered placement and an O(n^2) rank per output row; the driver actually calls
//     cuSPARSE csr2csc, source/DataLoader.cpp:65-69).  Here: a STABLE least-significant-digit radix sort of
//     (column, position) pairs, hand-written for wave64 (8-bit digits, ranks from ballots, no atomics on the
//     ordered path) -- the CSR order is by row, so a stable sort by column IS the transpose, rows ascending
//     inside every output row, whatever the length of that row (hub columns included).  Outside the timed path.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string.h>

#include <cmath>
#include <cstdio>

#include "../../include/speck_c_api.h"
#include "device_common.hpp"

using namespace speck;

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            std::fprintf(stderr, "speck_amd: HIP error %s at %s:%d\n", hipGetErrorString(_e), \
                         __FILE__, __LINE__);                                             \
            return (_e == hipErrorOutOfMemory) ? SPECK_ERR_OOM : SPECK_ERR_HIP;           \
        }                                                                                 \
    } while (0)

namespace {

// one wave per row.  Structure: offsets and column ids bit-exact.  Values (compare_data): relative to
// max(|x|, |y|), or -- when `scale` is given -- |x - y| <= rel_tol * scale[j] with scale = sum |a*b| of
// the entry (the bound any summation order satisfies; `scale` has the pattern of `ref`).
template <typename T>
__global__ void compare_kernel(const u32* __restrict__ ro_a, const u32* __restrict__ col_a,
                               const T* __restrict__ val_a, const u32* __restrict__ ro_b,
                               const u32* __restrict__ col_b, const T* __restrict__ val_b,
                               const u32* __restrict__ ro_s, const T* __restrict__ val_s,
                               u32 rows, int compare_data, double rel_tol,
                               unsigned long long* __restrict__ mismatches /*[2]: structure, values*/)
{
    SPECK_POISON();
    const u32 lane = lane_id();
    const u64 wave = (u64(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const u64 nwaves = (u64(gridDim.x) * blockDim.x) >> 6;
    for (u64 row = wave; row < rows; row += nwaves) {
        // offsets relative to the first one: row-range views compare equal to their copies
        const u32 a0 = ro_a[row], a1 = ro_a[row + 1], b0 = ro_b[row], b1 = ro_b[row + 1];
        bool bad = (a1 - a0) != (b1 - b0) || (a0 - ro_a[0]) != (b0 - ro_b[0]);
        bool badv = false;
        if (!bad) {
            const u32 s0 = ro_s ? ro_s[row] : 0u;
            if (ro_s && ro_s[row + 1] - s0 != a1 - a0) bad = true;
            for (u32 j = lane; j < a1 - a0 && !bad; j += 64) {
                if (col_a[a0 + j] != col_b[b0 + j]) bad = true;
                if (compare_data) {
                    const double x = (double)val_a[a0 + j], y = (double)val_b[b0 + j];
                    const double scale = val_s ? fabs((double)val_s[s0 + j]) : fmax(fabs(x), fabs(y));
                    if (!(fabs(x - y) <= rel_tol * scale + 1e-300)) badv = true;
                }
            }
        }
        if (__ballot(bad) != 0 && lane == 0) atomicAdd(&mismatches[0], 1ull);
        if (__ballot(badv) != 0 && lane == 0) atomicAdd(&mismatches[1], 1ull);
    }
}

__global__ void expand_rows_kernel(const u32* __restrict__ ro, u32 rows, u32 base,
                                   u32* __restrict__ row_of)
{
    SPECK_POISON();
    const u32 lane = lane_id();
    const u64 wave = (u64(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const u64 nwaves = (u64(gridDim.x) * blockDim.x) >> 6;
    for (u64 row = wave; row < rows; row += nwaves) {
        const u32 a0 = ro[row] - base, a1 = ro[row + 1] - base;
        for (u32 j = a0 + lane; j < a1; j += 64) row_of[j] = (u32)row;
    }
}

__global__ void iota_kernel(u32* p, u32 n)
{
    SPECK_POISON();
    for (u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += u64(gridDim.x) * blockDim.x)
        p[i] = (u32)i;
}

template <typename T>
__global__ void transpose_gather_kernel(const u32* __restrict__ perm, const u32* __restrict__ row_of,
                                        const T* __restrict__ val, u32 nnz,
                                        u32* __restrict__ t_col, T* __restrict__ t_val)
{
    SPECK_POISON();
    for (u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x; i < nnz; i += u64(gridDim.x) * blockDim.x) {
        const u32 src = perm[i];
        t_col[i] = row_of[src];
        t_val[i] = val[src];
    }
}

// t_ro[c] = lower bound of c in the sorted column keys
__global__ void offsets_from_sorted_kernel(const u32* __restrict__ keys, u32 nnz, u32 cols,
                                           u32* __restrict__ t_ro)
{
    SPECK_POISON();
    for (u64 c = u64(blockIdx.x) * blockDim.x + threadIdx.x; c <= cols; c += u64(gridDim.x) * blockDim.x) {
        u32 lo = 0, hi = nnz;
        while (lo < hi) {
            const u32 mid = lo + ((hi - lo) >> 1);
            if (keys[mid] < (u32)c) lo = mid + 1; else hi = mid;
        }
        t_ro[c] = lo;
    }
}

// ---- stable LSD radix sort of (key, payload) pairs, 8 bits per pass ------------------------------------------
// kRadixBlocks workgroups each own a contiguous slice of the input, walked in tiles of 256 x kRadixItems elements;
// element order inside a tile: wave w owns elements [w, w+1) x 64 x kRadixItems, step s of a wave holds 64
// consecutive ones (lane = position) -- so "earlier in the input" = (lower wave, lower step, lower lane).
//   radix_hist_kernel     per workgroup: digit histogram of its slice            -> hist[digit][workgroup]
//   radix_scan_kernel     one workgroup: exclusive scan of hist in (digit, workgroup) order = where the keys of
//                         (digit, workgroup) start in the output
//   radix_scatter_kernel  per tile: the rank of a key among the equal digits before it = running count of its
//                         wave (LDS, advanced by one leader lane per digit and step) + lanes before it in the step
//                         (ballots: the lanes holding MY digit) + the counts of the waves before mine
constexpr int kRadixThreads = 256, kRadixItems = 8, kRadixBlocks = 1024;
constexpr u32 kRadixTile = kRadixThreads * kRadixItems;

__device__ __forceinline__ u32 slice_begin(u32 n, u32 b)
{
    // slices are whole tiles (the last one takes the remainder)
    const u32 tiles = (n + kRadixTile - 1) / kRadixTile;
    const u32 per = (tiles + kRadixBlocks - 1) / kRadixBlocks;
    const u64 t0 = u64(b) * per;
    return (u32)(t0 * kRadixTile < n ? t0 * kRadixTile : n);
}

__global__ __launch_bounds__(kRadixThreads) void radix_hist_kernel(const u32* __restrict__ keys, u32 n, u32 shift,
                                                                  u32* __restrict__ hist)
{
    SPECK_POISON();
    __shared__ u32 s_cnt[256];
    s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const u32 lo = slice_begin(n, blockIdx.x), hi = slice_begin(n, blockIdx.x + 1);
    for (u32 i = lo + threadIdx.x; i < hi; i += kRadixThreads) atomicAdd(&s_cnt[(keys[i] >> shift) & 255u], 1u);
    __syncthreads();
    hist[threadIdx.x * kRadixBlocks + blockIdx.x] = s_cnt[threadIdx.x];
}

__global__ __launch_bounds__(1024) void radix_scan_kernel(u32* __restrict__ hist)
{
    SPECK_POISON();
    __shared__ u32 s_scan[1024 / 64 + 1];
    constexpr u32 per = 256 * kRadixBlocks / 1024;
    u32* mine = hist + threadIdx.x * per;
    u32 sum = 0;
    for (u32 i = 0; i < per; ++i) sum += mine[i];
    u32 total;
    u32 run = block_exclusive_scan<1024>(sum, s_scan, &total);
    for (u32 i = 0; i < per; ++i) {
        const u32 v = mine[i];
        mine[i] = run;
        run += v;
    }
}

__global__ __launch_bounds__(kRadixThreads) void radix_scatter_kernel(const u32* __restrict__ keys_in,
                                                                     const u32* __restrict__ vals_in, u32 n, u32 shift,
                                                                     const u32* __restrict__ hist,
                                                                     u32* __restrict__ keys_out,
                                                                     u32* __restrict__ vals_out)
{
    SPECK_POISON();
    constexpr int NW = kRadixThreads / 64;
    __shared__ u32 s_base[256];        // where the next key of each digit goes (this workgroup's share of the output)
    __shared__ u32 s_wave[NW][256];    // keys of each digit seen by each wave in the current tile
    const u32 lane = lane_id(), wid = threadIdx.x >> 6;
    s_base[threadIdx.x] = hist[threadIdx.x * kRadixBlocks + blockIdx.x];
    const u32 lo = slice_begin(n, blockIdx.x), hi = slice_begin(n, blockIdx.x + 1);
    for (u32 t0 = lo; t0 < hi; t0 += kRadixTile) {
        for (int w = 0; w < NW; ++w) s_wave[w][threadIdx.x] = 0;
        __syncthreads();
        u32 key[kRadixItems], val[kRadixItems], rank[kRadixItems];
#pragma unroll
        for (int it = 0; it < kRadixItems; ++it) {
            const u32 i = t0 + (wid * kRadixItems + it) * 64 + lane;
            const bool ok = i < hi;
            key[it] = ok ? keys_in[i] : 0u;
            val[it] = ok ? vals_in[i] : 0u;
            const u32 d = (key[it] >> shift) & 255u;
            // the lanes of this step that hold my digit (lanes past the end match nobody)
            u64 peers = __ballot(ok);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const u64 m = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? m : ~m;
            }
            const u32 before = s_wave[wid][d];  // (LDS operations of one wave complete in order)
            rank[it] = before + (u32)__popcll(peers & lanemask_lt());
            if (ok && (peers & lanemask_lt()) == 0) s_wave[wid][d] = before + (u32)__popcll(peers);
            wave_lds_fence();
        }
        __syncthreads();
        // digit threadIdx.x: exclusive prefix over the waves, then the tile's total moves the base
        {
            u32 run = s_base[threadIdx.x];
            for (int w = 0; w < NW; ++w) {
                const u32 c = s_wave[w][threadIdx.x];
                s_wave[w][threadIdx.x] = run;
                run += c;
            }
            s_base[threadIdx.x] = run;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < kRadixItems; ++it) {
            const u32 i = t0 + (wid * kRadixItems + it) * 64 + lane;
            if (i < hi) {
                const u32 pos = s_wave[wid][(key[it] >> shift) & 255u] + rank[it];
                keys_out[pos] = key[it];
                vals_out[pos] = val[it];
            }
        }
        __syncthreads();
    }
}

// sorts (keys, vals) by the low `bits` bits of the keys; a/b are ping-pong buffers (the input is in a, the result
// in the returned index: 0 = a, 1 = b)
int radix_sort_pairs(u32* keys_a, u32* vals_a, u32* keys_b, u32* vals_b, u32 n, unsigned bits, u32* hist)
{
    int cur = 0;
    for (unsigned shift = 0; shift < bits; shift += 8) {
        const u32* ki = cur ? keys_b : keys_a;
        const u32* vi = cur ? vals_b : vals_a;
        u32* ko = cur ? keys_a : keys_b;
        u32* vo = cur ? vals_a : vals_b;
        hipLaunchKernelGGL(radix_hist_kernel, dim3(kRadixBlocks), dim3(kRadixThreads), 0, 0, ki, n, shift, hist);
        hipLaunchKernelGGL(radix_scan_kernel, dim3(1), dim3(1024), 0, 0, hist);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(kRadixBlocks), dim3(kRadixThreads), 0, 0, ki, vi, n, shift,
                           (const u32*)hist, ko, vo);
        cur ^= 1;
    }
    return cur;
}

template <typename T>
int compare_impl(const speck_dcsr* ref, const speck_dcsr* cmp, const speck_dcsr* scale, int compare_data,
                 double rel_tol, uint64_t* h_structure, uint64_t* h_values)
{
    if (!ref || !cmp || !h_structure) return SPECK_ERR_INVALID;
    if (h_values) *h_values = 0;
    if (ref->rows != cmp->rows || ref->cols != cmp->cols || ref->nnz != cmp->nnz ||
        (scale && (scale->rows != ref->rows || scale->nnz != ref->nnz))) {
        *h_structure = ref->rows ? ref->rows : 1;
        return SPECK_OK;
    }
    if (ref->rows == 0) {
        *h_structure = 0;
        return SPECK_OK;
    }
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), 16));
    HIP_TRY(hipMemset(d, 0, 16));
    const u32 rows = (u32)ref->rows;
    u32 blocks = (rows + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(compare_kernel<T>, dim3(blocks), dim3(256), 0, 0, ref->row_offsets, ref->col_ids,
                       static_cast<const T*>(ref->data), cmp->row_offsets, cmp->col_ids,
                       static_cast<const T*>(cmp->data), scale ? scale->row_offsets : nullptr,
                       scale ? static_cast<const T*>(scale->data) : nullptr, rows, compare_data, rel_tol, d);
    unsigned long long h[2] = {0, 0};
    HIP_TRY(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    *h_structure = h[0];
    if (h_values) *h_values = h[1];
    else *h_structure += h[1];
    return SPECK_OK;
}

template <typename T>
int transpose_impl(const speck_dcsr* A, speck_dcsr* At)
{
    if (!A || !At) return SPECK_ERR_INVALID;
    const u32 nnz = (u32)A->nnz, rows = (u32)A->rows, cols = (u32)A->cols;
    int rc = speck_dcsr_alloc(At, cols, rows, nnz, 1, sizeof(T));
    if (rc != SPECK_OK) return rc;
    if (nnz == 0) {
        HIP_TRY(hipMemset(At->row_offsets, 0, (size_t(cols) + 1) * 4));
        return SPECK_OK;
    }
    u32 base = 0;
    HIP_TRY(hipMemcpy(&base, A->row_offsets, 4, hipMemcpyDeviceToHost));
    // the six temporaries are ONE allocation, released on every way out (a failing HIP call used to leak them)
    struct Temp {
        void* p = nullptr;
        ~Temp() { if (p) (void)hipFree(p); }
    } temp;
    const size_t n4 = (size_t(nnz) * 4 + 255) & ~size_t(255);
    HIP_TRY(hipMalloc(&temp.p, 5 * n4 + size_t(256) * kRadixBlocks * 4));
    unsigned char* tb = static_cast<unsigned char*>(temp.p);
    u32* row_of = reinterpret_cast<u32*>(tb);
    u32* perm_a = reinterpret_cast<u32*>(tb + n4);
    u32* perm_b = reinterpret_cast<u32*>(tb + 2 * n4);
    u32* keys_a = reinterpret_cast<u32*>(tb + 3 * n4);
    u32* keys_b = reinterpret_cast<u32*>(tb + 4 * n4);
    u32* hist = reinterpret_cast<u32*>(tb + 5 * n4);
    u32 blocks = (rows + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(expand_rows_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, 0, A->row_offsets,
                       rows, base, row_of);
    hipLaunchKernelGGL(iota_kernel, dim3(2048), dim3(256), 0, 0, perm_a, nnz);
    HIP_TRY(hipMemcpyAsync(keys_a, A->col_ids + base, size_t(nnz) * 4, hipMemcpyDeviceToDevice, 0));
    unsigned end_bit = 1;
    while (end_bit < 32 && (1ull << end_bit) < (cols ? cols : 1)) ++end_bit;
    const int res = radix_sort_pairs(keys_a, perm_a, keys_b, perm_b, nnz, end_bit, hist);
    const u32* keys_out = res ? keys_b : keys_a;
    const u32* perm_out = res ? perm_b : perm_a;
    hipLaunchKernelGGL(transpose_gather_kernel<T>, dim3(2048), dim3(256), 0, 0, perm_out, row_of,
                       static_cast<const T*>(A->data) + base, nnz, At->col_ids, static_cast<T*>(At->data));
    hipLaunchKernelGGL(offsets_from_sorted_kernel, dim3(2048), dim3(256), 0, 0, keys_out, nnz, cols,
                       At->row_offsets);
    HIP_TRY(hipDeviceSynchronize());
    return SPECK_OK;
}

}  // namespace

extern "C" {

int speck_compare_f64(speck_config* /*cfg*/, const speck_dcsr* ref, const speck_dcsr* cmp,
                      int compare_data, double rel_tol, uint64_t* h_mismatches)
{
    return compare_impl<double>(ref, cmp, nullptr, compare_data, rel_tol, h_mismatches, nullptr);
}

int speck_compare_f32(speck_config* /*cfg*/, const speck_dcsr* ref, const speck_dcsr* cmp,
                      int compare_data, double rel_tol, uint64_t* h_mismatches)
{
    return compare_impl<float>(ref, cmp, nullptr, compare_data, rel_tol, h_mismatches, nullptr);
}

int speck_compare_bounded_f64(speck_config* /*cfg*/, const speck_dcsr* ref, const speck_dcsr* cmp,
                              const speck_dcsr* abs_products, double tol, uint64_t* h_structure_rows,
                              uint64_t* h_value_rows)
{
    if (!abs_products || !h_value_rows) return SPECK_ERR_INVALID;
    return compare_impl<double>(ref, cmp, abs_products, 1, tol, h_structure_rows, h_value_rows);
}

int speck_transpose_f64(speck_config* /*cfg*/, const speck_dcsr* A, speck_dcsr* At)
{
    return transpose_impl<double>(A, At);
}

int speck_transpose_f32(speck_config* /*cfg*/, const speck_dcsr* A, speck_dcsr* At)
{
    return transpose_impl<float>(A, At);
}

}  // extern "C"
