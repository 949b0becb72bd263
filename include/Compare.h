// Compare.h -- spECK::Compare of the reference (include/Compare.h:5-6, source/GPU/Compare.cu:65-84; float
// and double), stricter: row offsets and column ids bit-exact; values (when compare_data) within 1e-12
// relative (2e-5 for float).
#pragma once
#include "dCSR.h"

namespace spECK {
template <typename DataType>
bool Compare(const dCSR<DataType>& reference_mat, const dCSR<DataType>& compare_mat, bool compare_data)
{
    speck_dcsr a = reference_mat.raw(), b = compare_mat.raw();
    uint64_t bad = 1;
    const int rc = sizeof(DataType) == 8 ? speck_compare_f64(nullptr, &a, &b, compare_data ? 1 : 0, 1e-12, &bad)
                                         : speck_compare_f32(nullptr, &a, &b, compare_data ? 1 : 0, 2e-5, &bad);
    return rc == SPECK_OK && bad == 0;
}
}  // namespace spECK
