// cxx_api.cpp -- the explicit instantiations the reference's library exports (source/GPU/Multiply.cu:1130-1131):
//   spECK::MultiplyspECK<float|double, 4, 1024, spECK_DYNAMIC_MEM_PER_BLOCK, spECK_STATIC_MEM_PER_BLOCK>
// so that a caller compiled against declarations only (SPECK_DECLARATIONS_ONLY) links like one of upstream.
#include "../../include/Multiply.h"

namespace spECK {
template void MultiplyspECK<float, 4, 1024, spECK_DYNAMIC_MEM_PER_BLOCK, spECK_STATIC_MEM_PER_BLOCK>(
    const dCSR<float>&, const dCSR<float>&, dCSR<float>&, spECKConfig&, Timings&);
template void MultiplyspECK<double, 4, 1024, spECK_DYNAMIC_MEM_PER_BLOCK, spECK_STATIC_MEM_PER_BLOCK>(
    const dCSR<double>&, const dCSR<double>&, dCSR<double>&, spECKConfig&, Timings&);
}  // namespace spECK
