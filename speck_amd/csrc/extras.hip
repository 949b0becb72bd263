// extras.hip -- the two side utilities the reference driver uses around the hot path:
//   * spECK::Compare   (reference source/GPU/Compare.cu:11-82) -- made strict: offsets and
//     column ids bit-exact, values by relative tolerance
//   * transpose for non-square A (reference source/GPU/Transpose.cu:10-117; the driver
//     actually calls cuSPARSE csr2csc, source/DataLoader.cpp:65-69).  Order preserving:
//     a STABLE device radix sort of (column, position) pairs -- rocPRIM, the one library
//     call in this backend, used outside the timed path only.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

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
    for (u64 i = u64(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += u64(gridDim.x) * blockDim.x)
        p[i] = (u32)i;
}

template <typename T>
__global__ void transpose_gather_kernel(const u32* __restrict__ perm, const u32* __restrict__ row_of,
                                        const T* __restrict__ val, u32 nnz,
                                        u32* __restrict__ t_col, T* __restrict__ t_val)
{
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
    for (u64 c = u64(blockIdx.x) * blockDim.x + threadIdx.x; c <= cols; c += u64(gridDim.x) * blockDim.x) {
        u32 lo = 0, hi = nnz;
        while (lo < hi) {
            const u32 mid = lo + ((hi - lo) >> 1);
            if (keys[mid] < (u32)c) lo = mid + 1; else hi = mid;
        }
        t_ro[c] = lo;
    }
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
    u32 *row_of = nullptr, *perm_in = nullptr, *perm_out = nullptr, *keys_out = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&row_of), size_t(nnz) * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&perm_in), size_t(nnz) * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&perm_out), size_t(nnz) * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&keys_out), size_t(nnz) * 4));
    u32 blocks = (rows + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(expand_rows_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, 0, A->row_offsets,
                       rows, base, row_of);
    hipLaunchKernelGGL(iota_kernel, dim3(2048), dim3(256), 0, 0, perm_in, nnz);
    const u32* keys_in = A->col_ids + base;
    unsigned end_bit = 1;
    while (end_bit < 32 && (1ull << end_bit) < (cols ? cols : 1)) ++end_bit;
    size_t tmp_bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, keys_out, perm_in, perm_out, nnz, 0,
                                      end_bit, (hipStream_t)0));
    void* tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
    HIP_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, perm_in, perm_out, nnz, 0,
                                      end_bit, (hipStream_t)0));
    hipLaunchKernelGGL(transpose_gather_kernel<T>, dim3(2048), dim3(256), 0, 0, perm_out, row_of,
                       static_cast<const T*>(A->data) + base, nnz, At->col_ids, static_cast<T*>(At->data));
    hipLaunchKernelGGL(offsets_from_sorted_kernel, dim3(2048), dim3(256), 0, 0, keys_out, nnz, cols,
                       At->row_offsets);
    HIP_TRY(hipDeviceSynchronize());
    (void)hipFree(tmp);
    (void)hipFree(row_of);
    (void)hipFree(perm_in);
    (void)hipFree(perm_out);
    (void)hipFree(keys_out);
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
