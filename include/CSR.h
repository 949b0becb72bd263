// CSR.h -- host CSR of the reference (include/CSR.h:57-65) + loaders, over libspeck_amd.so.
#pragma once

#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>

#include "dCSR.h"
#include "speck_c_api.h"

template <typename T>
struct CSR {
    size_t rows, cols, nnz;
    std::unique_ptr<T[]> data;
    std::unique_ptr<unsigned int[]> row_offsets;
    std::unique_ptr<unsigned int[]> col_ids;

    CSR() : rows(0), cols(0), nnz(0) {}
    void alloc(size_t r, size_t c, size_t n)
    {
        rows = r;
        cols = c;
        nnz = n;
        data = std::make_unique<T[]>(n);
        col_ids = std::make_unique<unsigned int[]>(n);
        row_offsets = std::make_unique<unsigned int[]>(r + 1);
    }
};

namespace speck_detail {
template <typename T>
inline CSR<T> from_handle(speck_host_csr* h)
{
    uint64_t r, c, n;
    speck_host_csr_dims(h, &r, &c, &n);
    CSR<T> m;
    m.alloc(r, c, n);
    std::unique_ptr<double[]> tmp(new double[n ? n : 1]);
    speck_host_csr_copy(h, m.row_offsets.get(), m.col_ids.get(), tmp.get());
    for (uint64_t i = 0; i < n; ++i) m.data[i] = static_cast<T>(tmp[i]);
    speck_host_csr_free(h);
    return m;
}
}  // namespace speck_detail

// loadCSR / storeCSR (.hicsr), reference source/CSR.cpp:88-137
template <typename T>
CSR<T> loadCSR(const char* file)
{
    speck_host_csr* h = nullptr;
    if (speck_load_hicsr(file, &h) != SPECK_OK) throw std::runtime_error(std::string("could not open \"") + file + "\"");
    return speck_detail::from_handle<T>(h);
}
template <typename T>
void storeCSR(const CSR<T>& mat, const char* file)
{
    std::unique_ptr<double[]> tmp(new double[mat.nnz ? mat.nnz : 1]);
    for (size_t i = 0; i < mat.nnz; ++i) tmp[i] = mat.data[i];
    speck_host_csr* h = nullptr;
    speck_host_csr_from_arrays(mat.rows, mat.cols, mat.nnz, mat.row_offsets.get(), mat.col_ids.get(), tmp.get(), &h);
    const int rc = speck_store_hicsr(h, file);
    speck_host_csr_free(h);
    if (rc != SPECK_OK) throw std::runtime_error(std::string("could not open \"") + file + "\"");
}
// loadMTX + convert(COO->CSR) in one step (reference source/COO.cpp:53-164 + source/CSR.cpp:173-212)
template <typename T>
CSR<T> loadMTXasCSR(const char* file)
{
    speck_host_csr* h = nullptr;
    if (speck_load_mtx(file, &h) != SPECK_OK) throw std::runtime_error(std::string("could not load mtx file: \"") + file + "\"");
    return speck_detail::from_handle<T>(h);
}

// ---- convert() (reference source/dCSR.cpp:51-115).  `padding`: buffers for rows + padding rows and nnz + 8 * padding
// entries, rows / nnz / cols of the source (dCSR.cpp:53-54, 70-71, 83-84, 94-95).
template <typename T>
void convert(dCSR<T>& dst, const CSR<T>& src, unsigned int padding)
{
    speck_dcsr d = dst.raw();
    speck_dcsr_upload_padded(&d, src.rows, src.cols, src.nnz, src.row_offsets.get(), src.col_ids.get(), src.data.get(),
                             sizeof(T), padding);
    dst.adopt(d);
}
template <typename T>
void convert(CSR<T>& dst, const dCSR<T>& src, unsigned int padding)
{
    dst.alloc(src.rows + padding, src.cols, src.nnz + 8 * size_t(padding));
    dst.rows = src.rows;
    dst.nnz = src.nnz;
    dst.cols = src.cols;
    speck_dcsr d = src.raw();
    speck_dcsr_download(&d, dst.row_offsets.get(), dst.col_ids.get(), dst.data.get(), sizeof(T));
}
// device to device: no host round trip (dCSR.cpp:81-89)
template <typename T>
void convert(dCSR<T>& dst, const dCSR<T>& src, unsigned int padding)
{
    speck_dcsr d = dst.raw(), s = src.raw();
    speck_dcsr_copy(&d, &s, sizeof(T), padding);
    dst.adopt(d);
}
template <typename T>
void convert(CSR<T>& dst, const CSR<T>& src, unsigned int padding)
{
    dst.alloc(src.rows + padding, src.cols, src.nnz + 8 * size_t(padding));
    dst.rows = src.rows;
    dst.nnz = src.nnz;
    dst.cols = src.cols;
    std::memcpy(dst.data.get(), src.data.get(), src.nnz * sizeof(T));
    std::memcpy(dst.col_ids.get(), src.col_ids.get(), src.nnz * sizeof(unsigned int));
    std::memcpy(dst.row_offsets.get(), src.row_offsets.get(), (src.rows + 1) * sizeof(unsigned int));
}
