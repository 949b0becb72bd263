// rocPRIM segmented radix sort of (u32 column, f64 value) pairs with the segment sizes of the
// NUM_G rows of the webbase-like input: is a sort-based spill path faster than L2 atomics?
#include <cstring>
#include <string.h>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
#include <random>
int main()
{
    std::mt19937 rng(1);
    const int segs = 1400;
    std::vector<unsigned> off(segs + 1, 0);
    // sizes: lognormal-ish between 5.5k and 113k, mean ~14k
    for (int i = 0; i < segs; ++i) {
        double u = std::generate_canonical<double, 32>(rng);
        unsigned sz = (unsigned)(5500.0 * std::pow(20.0, u * u * u));
        off[i + 1] = off[i] + sz;
    }
    const size_t n = off[segs];
    printf("segments %d, elements %zu\n", segs, n);
    std::vector<unsigned> keys(n);
    for (auto& k : keys) k = rng() & 0xFFFFF;
    unsigned *d_ki, *d_ko, *d_off;
    double *d_vi, *d_vo;
    hipMalloc(&d_ki, n * 4); hipMalloc(&d_ko, n * 4); hipMalloc(&d_vi, n * 8); hipMalloc(&d_vo, n * 8);
    hipMalloc(&d_off, (segs + 1) * 4);
    hipMemcpy(d_ki, keys.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(d_vi, 0, n * 8);
    hipMemcpy(d_off, off.data(), (segs + 1) * 4, hipMemcpyHostToDevice);
    size_t tmp_bytes = 0;
    rocprim::segmented_radix_sort_pairs(nullptr, tmp_bytes, d_ki, d_ko, d_vi, d_vo, n, segs, d_off, d_off + 1, 0, 20, 0);
    void* tmp;
    hipMalloc(&tmp, tmp_bytes);
    printf("temp %zu bytes\n", tmp_bytes);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(a, 0);
        rocprim::segmented_radix_sort_pairs(tmp, tmp_bytes, d_ki, d_ko, d_vi, d_vo, n, segs, d_off, d_off + 1, 0, 20, 0);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        printf("segmented_radix_sort_pairs: %.3f ms (%.1f M pairs/ms)\n", ms, n / ms * 1e-6);
    }
    // whole-array sort of (segment<<20 | key) as an alternative
    return 0;
}
