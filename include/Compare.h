// Compare.h -- spECK::Compare of the reference (include/Compare.h:5-6, source/GPU/Compare.cu:65-82),
// stricter: row offsets and column ids bit-exact; values (when compare_data) within 1e-12 relative.
#pragma once
#include "dCSR.h"

namespace spECK {
template <typename DataType>
bool Compare(const dCSR<DataType>& reference_mat, const dCSR<DataType>& compare_mat, bool compare_data)
{
    static_assert(sizeof(DataType) == 8, "Compare is provided for double (the reference driver's type)");
    speck_dcsr a = reference_mat.raw(), b = compare_mat.raw();
    uint64_t bad = 1;
    if (speck_compare_f64(nullptr, &a, &b, compare_data ? 1 : 0, 1e-12, &bad) != SPECK_OK) return false;
    return bad == 0;
}
}  // namespace spECK
