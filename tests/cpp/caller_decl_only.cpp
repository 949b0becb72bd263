// A caller that sees DECLARATIONS only (as with the reference's include/Multiply.h) and links the explicit
// instantiations libspeck_amd.so exports (reference: source/GPU/Multiply.cu:1130-1131).  Also touches the public
// stream / event fields of spECKConfig (reference include/spECKConfig.h:12-13).
//   g++ -std=c++17 -DSPECK_DECLARATIONS_ONLY -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include \
//       tests/cpp/caller_decl_only.cpp -Lspeck_amd -lspeck_amd -L/opt/rocm/lib -lamdhip64
#include <cstdio>
#include <vector>

#include "CSR.h"
#include "Multiply.h"

template <typename T>
static int one(spECK::spECKConfig& config)
{
    // A = [[1,2,0],[0,3,0],[4,0,5]]
    CSR<T> a;
    a.alloc(3, 3, 5);
    const unsigned ro[4] = {0, 2, 3, 5}, ci[5] = {0, 1, 1, 0, 2};
    const T v[5] = {1, 2, 3, 4, 5};
    for (int i = 0; i < 4; ++i) a.row_offsets[i] = ro[i];
    for (int i = 0; i < 5; ++i) a.col_ids[i] = ci[i], a.data[i] = v[i];
    dCSR<T> dA, dC;
    convert(dA, a, 0);
    Timings t;
    spECK::MultiplyspECK<T, 4, 1024, spECK_DYNAMIC_MEM_PER_BLOCK, spECK_STATIC_MEM_PER_BLOCK>(dA, dA, dC, config, t);
    CSR<T> c;
    convert(c, dC, 0);
    // A*A = [[1,8,0],[0,9,0],[24,8,25]]
    const unsigned want_ro[4] = {0, 2, 3, 6}, want_ci[6] = {0, 1, 1, 0, 1, 2};
    const T want_v[6] = {1, 8, 9, 24, 8, 25};
    if (c.nnz != 6) return 1;
    for (int i = 0; i < 4; ++i)
        if (c.row_offsets[i] != want_ro[i]) return 2;
    for (int i = 0; i < 6; ++i)
        if (c.col_ids[i] != want_ci[i] || c.data[i] != want_v[i]) return 3;
    return 0;
}

int main()
{
    auto config = spECK::spECKConfig::initialize(0);
    if (config.streams.size() != 6 || !config.completeStart || !config.completeEnd || !config.individualStart ||
        !config.individualEnd)
        return 10;
    for (auto s : config.streams)
        if (!s || hipStreamQuery(s) != hipSuccess) return 11;  // real, idle HIP streams
    const int rc = one<double>(config) * 10 + one<float>(config);
    config.cleanup();
    std::printf(rc == 0 ? "decl-only caller ok\n" : "decl-only caller FAILED %d\n", rc);
    return rc;
}
