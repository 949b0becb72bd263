// Transpose.h -- spECK::Transpose of the reference (include/Transpose.h, source/GPU/Transpose.cu:10-117).
#pragma once
#include "dCSR.h"

namespace spECK {
template <typename DataType>
void Transpose(const dCSR<DataType>& matIn, dCSR<DataType>& matTransposeOut)
{
    static_assert(sizeof(DataType) == 8, "Transpose is provided for double (the reference driver's type)");
    speck_dcsr a = matIn.raw(), t = matTransposeOut.raw();
    speck_transpose_f64(nullptr, &a, &t);
    matTransposeOut.adopt(t);
}
}  // namespace spECK
