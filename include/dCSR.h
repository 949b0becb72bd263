// dCSR.h -- device CSR of the reference (include/dCSR.h:9-47) over the C ABI of libspeck_amd.so.
// Same names, same field order (rows, cols, nnz, data, row_offsets, col_ids), same ownership:
// the destructor frees all three device buffers (reference source/dCSR.cpp:6-41).
#pragma once

#include <cstddef>
#include <cstring>
#include <type_traits>

#include "speck_c_api.h"

template <typename T>
struct CSR;

template <typename T>
struct dCSR {
    static_assert(std::is_same<T, double>::value || std::is_same<T, float>::value, "float or double");
    size_t rows, cols, nnz;

    T* data;
    unsigned int* row_offsets;
    unsigned int* col_ids;

    dCSR() : rows(0), cols(0), nnz(0), data(nullptr), row_offsets(nullptr), col_ids(nullptr) {}
    dCSR(const dCSR&) = delete;
    dCSR& operator=(const dCSR&) = delete;

    void alloc(size_t r, size_t c, size_t n, bool allocOffsets = true)
    {
        speck_dcsr d = raw();
        speck_dcsr_alloc(&d, r, c, n, allocOffsets ? 1 : 0, sizeof(T));
        adopt(d);
    }
    void reset()
    {
        speck_dcsr d = raw();
        speck_dcsr_free(&d);
        adopt(d);
    }
    virtual ~dCSR() { reset(); }

    // view for the C ABI (no ownership transfer)
    speck_dcsr raw() const { return speck_dcsr{rows, cols, nnz, data, row_offsets, col_ids}; }
    void adopt(const speck_dcsr& d)
    {
        rows = d.rows;
        cols = d.cols;
        nnz = d.nnz;
        data = static_cast<T*>(d.data);
        row_offsets = d.row_offsets;
        col_ids = d.col_ids;
    }
};

// Trivially-copyable view handed to kernels in the reference (include/dCSR.h:24-35).
template <typename T>
struct dCSRNoDealloc {
    size_t rows, cols, nnz;
    T* data;
    unsigned int* row_offsets;
    unsigned int* col_ids;
    dCSRNoDealloc(const dCSR<T>& a)
        : rows(a.rows), cols(a.cols), nnz(a.nnz), data(a.data), row_offsets(a.row_offsets), col_ids(a.col_ids)
    {
    }
    dCSRNoDealloc() = default;
};

// convert() overloads of the reference (include/dCSR.h:37-47, source/dCSR.cpp:51-99); defined in CSR.h
template <typename T>
void convert(dCSR<T>& dcsr, const CSR<T>& csr, unsigned int padding = 0);
template <typename T>
void convert(dCSR<T>& dcsr, const dCSR<T>& csr, unsigned int padding = 0);
template <typename T>
void convert(CSR<T>& csr, const dCSR<T>& dcsr, unsigned int padding = 0);
template <typename T>
void convert(CSR<T>& csr, const CSR<T>& other, unsigned int padding = 0);
