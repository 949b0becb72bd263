// Transpose.h -- spECK::Transpose of the reference (include/Transpose.h, source/GPU/Transpose.cu:10-117;
// instantiated for float and double, :116-117).
#pragma once
#include "dCSR.h"

namespace spECK {
template <typename DataType>
void Transpose(const dCSR<DataType>& matIn, dCSR<DataType>& matTransposeOut)
{
    speck_dcsr a = matIn.raw(), t = matTransposeOut.raw();
    if (sizeof(DataType) == 8) speck_transpose_f64(nullptr, &a, &t);
    else speck_transpose_f32(nullptr, &a, &t);
    matTransposeOut.adopt(t);
}
}  // namespace spECK
