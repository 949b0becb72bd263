// Multiply.h -- the reference's public entry point (include/Multiply.h:13-20) over the C ABI.
//   spECK::MultiplyspECK<T, BLOCKS_PER_SM, THREADS_PER_BLOCK, MAX_DYNAMIC_SHARED, MAX_STATIC_SHARED>
//       (A, B, matOut, config, timings)
// The four integer template arguments are accepted for source compatibility with callers such as
// the reference's Executor.cpp:48; this backend sizes its kernels for gfx950 (160 KiB LDS) itself.
// libspeck_amd.so exports the two explicit instantiations the reference's library exports
// (source/GPU/Multiply.cu:1130-1131): <float|double, 4, 1024, spECK_DYNAMIC_MEM_PER_BLOCK,
// spECK_STATIC_MEM_PER_BLOCK>.  A caller that defines SPECK_DECLARATIONS_ONLY sees declarations only -- as with
// the reference's header -- and links against those; otherwise the (thin) bodies are inline.
#pragma once

#include <cstdio>

#include "Timings.h"
#include "dCSR.h"
#include "spECKConfig.h"

static constexpr int spECK_STATIC_MEM_PER_BLOCK{65536};
static constexpr int spECK_DYNAMIC_MEM_PER_BLOCK{163840};

namespace spECK {
template <typename DataType, int BLOCKS_PER_SM, int THREADS_PER_BLOCK, int MAX_DYNAMIC_SHARED, int MAX_STATIC_SHARED>
void MultiplyspECK(const dCSR<DataType>& A, const dCSR<DataType>& B, dCSR<DataType>& matOut, spECKConfig& config,
                   Timings& timings);

template <typename DataType, int BLOCKS_PER_SM, int THREADS_PER_BLOCK, int MAX_DYNAMIC_SHARED, int MAX_STATIC_SHARED>
void MultiplyspECKImplementation(const dCSR<DataType>& A, const dCSR<DataType>& B, dCSR<DataType>& matOut,
                                 spECKConfig& config, Timings& timings);
// The reference declares the last argument as `Timings &timings = Timings()` (include/Multiply.h:19) -- a temporary
// bound to a non-const reference, which only its own compiler accepts.  The call it permits, without the timings
// argument, is this overload.
template <typename DataType, int BLOCKS_PER_SM, int THREADS_PER_BLOCK, int MAX_DYNAMIC_SHARED, int MAX_STATIC_SHARED>
inline void MultiplyspECKImplementation(const dCSR<DataType>& A, const dCSR<DataType>& B, dCSR<DataType>& matOut,
                                        spECKConfig& config)
{
    Timings timings;
    MultiplyspECKImplementation<DataType, BLOCKS_PER_SM, THREADS_PER_BLOCK, MAX_DYNAMIC_SHARED, MAX_STATIC_SHARED>(
        A, B, matOut, config, timings);
}

#ifndef SPECK_DECLARATIONS_ONLY
template <typename DataType, int BLOCKS_PER_SM, int THREADS_PER_BLOCK, int MAX_DYNAMIC_SHARED, int MAX_STATIC_SHARED>
void MultiplyspECKImplementation(const dCSR<DataType>& A, const dCSR<DataType>& B, dCSR<DataType>& matOut,
                                 spECKConfig& config, Timings& timings)
{
    speck_dcsr a = A.raw(), b = B.raw(), c = matOut.raw();
    speck_timings t = timings.to_c();
    const int rc = sizeof(DataType) == 8 ? speck_multiply_f64(config.handle, &a, &b, &c, &t)
                                         : speck_multiply_f32(config.handle, &a, &b, &c, &t);
    // adopt on every path: on a guard failure `c` comes back unmodified; a failure after C was
    // re-allocated leaves `c` owning the NEW buffers (the old ones are already freed)
    matOut.adopt(c);
    if (rc != SPECK_OK) {
        // the reference printf()s and returns (Multiply.cu:57-97)
        std::printf("ERROR: %s\n", speck_status_string(rc));
        return;
    }
    timings.from_c(t);
}

template <typename DataType, int BLOCKS_PER_SM, int THREADS_PER_BLOCK, int MAX_DYNAMIC_SHARED, int MAX_STATIC_SHARED>
void MultiplyspECK(const dCSR<DataType>& A, const dCSR<DataType>& B, dCSR<DataType>& matOut, spECKConfig& config,
                   Timings& timings)
{
    MultiplyspECKImplementation<DataType, BLOCKS_PER_SM, THREADS_PER_BLOCK, MAX_DYNAMIC_SHARED, MAX_STATIC_SHARED>(
        A, B, matOut, config, timings);
}
#endif  // SPECK_DECLARATIONS_ONLY

// exported by libspeck_amd.so (speck_amd/csrc/cxx_api.cpp)
extern template void MultiplyspECK<float, 4, 1024, spECK_DYNAMIC_MEM_PER_BLOCK, spECK_STATIC_MEM_PER_BLOCK>(
    const dCSR<float>&, const dCSR<float>&, dCSR<float>&, spECKConfig&, Timings&);
extern template void MultiplyspECK<double, 4, 1024, spECK_DYNAMIC_MEM_PER_BLOCK, spECK_STATIC_MEM_PER_BLOCK>(
    const dCSR<double>&, const dCSR<double>&, dCSR<double>&, spECKConfig&, Timings&);
}  // namespace spECK
