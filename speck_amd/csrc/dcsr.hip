// dcsr.hip -- the device CSR container behind the C-ABI (include/speck_c_api.h): allocation, host <-> device
// conversion and the device-to-device copy.  Role of the reference's dCSR<T> (include/dCSR.h:9-40,
// source/dCSR.cpp:14-108).  Every function here works on the NULL stream and returns after the work is complete
// (blocking copies / a stream synchronise): a caller that fills `src` on its own non-blocking stream synchronises
// that stream first -- the same contract as the reference's cudaMemcpy-based convert().
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>

#include "../../include/speck_c_api.h"
#include "device_common.hpp"
#include "guards.hpp"

using namespace speck;

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            std::fprintf(stderr, "speck_amd: HIP error %s at %s:%d (%s)\n",                 \
                         hipGetErrorString(_e), __FILE__, __LINE__, #expr);                 \
            return (_e == hipErrorOutOfMemory) ? SPECK_ERR_OOM : SPECK_ERR_HIP;             \
        }                                                                                   \
    } while (0)

namespace {

// speck_dcsr_copy, entirely on the device: the source may be a row-range view whose offsets are absolute -- its first
// offset is read HERE, not on the host (no device-to-host copy anywhere in the conversion).
// words: col_ids as u32, values as u32 pairs / singles (vwords = value_size / 4)
__global__ __launch_bounds__(256) void dcsr_copy_kernel(const u32* __restrict__ s_ro, const u32* __restrict__ s_col,
                                                        const u32* __restrict__ s_val, u32* __restrict__ d_ro,
                                                        u32* __restrict__ d_col, u32* __restrict__ d_val, u64 rows, u64 nnz,
                                                        u32 vwords)
{
    const u32 base = rows ? s_ro[0] : 0u;
    const u64 tid = u64(blockIdx.x) * 256 + threadIdx.x, nth = u64(gridDim.x) * 256;
    for (u64 i = tid; i <= rows; i += nth) d_ro[i] = rows ? s_ro[i] - base : 0u;
    for (u64 i = tid; i < nnz; i += nth) d_col[i] = s_col[u64(base) + i];
    const u64 nv = nnz * vwords;
    const u32* sv = s_val + u64(base) * vwords;
    for (u64 i = tid; i < nv; i += nth) d_val[i] = sv[i];
}

}  // namespace

extern "C" {

int speck_dcsr_alloc(speck_dcsr* m, uint64_t rows, uint64_t cols, uint64_t nnz, int alloc_offsets,
                     size_t value_size)
{
    if (!m) return SPECK_ERR_INVALID;
    speck_dcsr_free(m);
    m->rows = rows;
    m->cols = cols;
    m->nnz = nnz;
    HIP_TRY(guarded_malloc(&m->data, std::max<size_t>(nnz, 1) * value_size));
    HIP_TRY(guarded_malloc(reinterpret_cast<void**>(&m->col_ids), std::max<size_t>(nnz, 1) * 4));
    if (alloc_offsets) HIP_TRY(guarded_malloc(reinterpret_cast<void**>(&m->row_offsets), (rows + 1) * 4));
    return SPECK_OK;
}

int speck_dcsr_free(speck_dcsr* m)
{
    if (!m) return SPECK_ERR_INVALID;
    if (m->col_ids) (void)guarded_free(m->col_ids);
    if (m->data) (void)guarded_free(m->data);
    if (m->row_offsets) (void)guarded_free(m->row_offsets);
    m->col_ids = nullptr;
    m->data = nullptr;
    m->row_offsets = nullptr;
    m->nnz = 0;
    m->rows = 0;
    return SPECK_OK;
}

int speck_dcsr_upload(speck_dcsr* dst, uint64_t rows, uint64_t cols, uint64_t nnz,
                      const uint32_t* h_row_offsets, const uint32_t* h_col_ids, const void* h_data,
                      size_t value_size)
{
    return speck_dcsr_upload_padded(dst, rows, cols, nnz, h_row_offsets, h_col_ids, h_data, value_size, 0u);
}

int speck_dcsr_upload_padded(speck_dcsr* dst, uint64_t rows, uint64_t cols, uint64_t nnz, const uint32_t* h_row_offsets,
                             const uint32_t* h_col_ids, const void* h_data, size_t value_size, uint32_t padding)
{
    // reference: dst.alloc(rows + padding, cols, nnz + 8 * padding), then rows / nnz of the source (dCSR.cpp:53-54)
    int rc = speck_dcsr_alloc(dst, rows + padding, cols, nnz + 8ull * padding, 1, value_size);
    if (rc != SPECK_OK) return rc;
    dst->rows = rows;
    dst->nnz = nnz;
    if (nnz) {
        HIP_TRY(hipMemcpy(dst->data, h_data, nnz * value_size, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dst->col_ids, h_col_ids, nnz * 4, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(dst->row_offsets, h_row_offsets, (rows + 1) * 4, hipMemcpyHostToDevice));
    if (padding) {  // dCSR.cpp:59-64
        HIP_TRY(hipMemset(static_cast<char*>(dst->data) + nnz * value_size, 0, 8ull * padding * value_size));
        HIP_TRY(hipMemset(dst->col_ids + nnz, 0, 8ull * padding * 4));
        HIP_TRY(hipMemset(dst->row_offsets + rows + 1, 0, size_t(padding) * 4));
    }
    return SPECK_OK;
}

int speck_dcsr_copy(speck_dcsr* dst, const speck_dcsr* src, size_t value_size, uint32_t padding)
{
    if (!dst || !src || dst == src) return SPECK_ERR_INVALID;
    if (src->rows && !src->row_offsets) return SPECK_ERR_INVALID;
    // alloc frees dst first (dCSR.cpp:28): a destination that shares ANY buffer with the source would free it
    if ((dst->data && dst->data == src->data) || (dst->col_ids && dst->col_ids == src->col_ids) ||
        (dst->row_offsets && dst->row_offsets == src->row_offsets))
        return SPECK_ERR_INVALID;
    const uint64_t rows = src->rows, nnz = src->nnz;
    int rc = speck_dcsr_alloc(dst, rows + padding, src->cols, nnz + 8ull * padding, 1, value_size);
    if (rc != SPECK_OK) return rc;
    dst->rows = rows;
    dst->nnz = nnz;
    const u64 work = std::max<u64>(rows + 1, nnz * (value_size / 4));
    hipLaunchKernelGGL(dcsr_copy_kernel, dim3((unsigned)std::min<u64>((work + 255) / 256, 8192)), dim3(256), 0, nullptr,
                       src->row_offsets, src->col_ids, static_cast<const u32*>(src->data), dst->row_offsets, dst->col_ids,
                       static_cast<u32*>(dst->data), rows, nnz, (u32)(value_size / 4));
    HIP_TRY(hipGetLastError());
    if (padding) {
        HIP_TRY(hipMemsetAsync(static_cast<char*>(dst->data) + nnz * value_size, 0, 8ull * padding * value_size, nullptr));
        HIP_TRY(hipMemsetAsync(dst->col_ids + nnz, 0, 8ull * padding * 4, nullptr));
        HIP_TRY(hipMemsetAsync(dst->row_offsets + rows + 1, 0, size_t(padding) * 4, nullptr));
    }
    HIP_TRY(hipStreamSynchronize(nullptr));
    return SPECK_OK;
}

int speck_dcsr_download(const speck_dcsr* src, uint32_t* h_row_offsets, uint32_t* h_col_ids,
                        void* h_data, size_t value_size)
{
    if (!src) return SPECK_ERR_INVALID;
    if (src->nnz) {
        if (h_data) HIP_TRY(hipMemcpy(h_data, src->data, src->nnz * value_size, hipMemcpyDeviceToHost));
        if (h_col_ids) HIP_TRY(hipMemcpy(h_col_ids, src->col_ids, src->nnz * 4, hipMemcpyDeviceToHost));
    }
    if (h_row_offsets && src->row_offsets)
        HIP_TRY(hipMemcpy(h_row_offsets, src->row_offsets, (src->rows + 1) * 4, hipMemcpyDeviceToHost));
    return SPECK_OK;
}

int speck_dcsr_update(speck_dcsr* dst, const uint32_t* h_row_offsets, const uint32_t* h_col_ids,
                      const void* h_data, size_t value_size)
{
    if (!dst) return SPECK_ERR_INVALID;
    if (dst->nnz) {
        if (h_data) HIP_TRY(hipMemcpy(dst->data, h_data, dst->nnz * value_size, hipMemcpyHostToDevice));
        if (h_col_ids) HIP_TRY(hipMemcpy(dst->col_ids, h_col_ids, dst->nnz * 4, hipMemcpyHostToDevice));
    }
    if (h_row_offsets && dst->row_offsets)
        HIP_TRY(hipMemcpy(dst->row_offsets, h_row_offsets, (dst->rows + 1) * 4, hipMemcpyHostToDevice));
    return SPECK_OK;
}

}  // extern "C"
