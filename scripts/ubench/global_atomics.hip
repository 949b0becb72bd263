// Throughput of random global-memory atomics on gfx950 as a function of table footprint and of
// the memory scope: what bounds the NUM_G (global hash spill) class.
//   hipcc --offload-arch=gfx950 -O3 global_atomics.hip -o global_atomics && ./global_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t rnd(uint32_t& s)
{
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;
    return s;
}

// MODE 0: non-returning u32 add, 1: returning CAS(empty->key), 2: f64 add, 3: plain load (gather),
//      4: plain store
template <int MODE, int SCOPE>
__global__ void k(uint32_t* tab, uint64_t slots, int iters, uint32_t* sink)
{
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it += 4) {
        uint64_t idx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) idx[u] = (uint64_t(rnd(s)) * slots) >> 32;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) __hip_atomic_fetch_add(&tab[idx[u] * 4], 1u, __ATOMIC_RELAXED, SCOPE);
            if (MODE == 1) {
                uint32_t e = 0xFFFFFFFFu;
                __hip_atomic_compare_exchange_strong(&tab[idx[u] * 4], &e, s, __ATOMIC_RELAXED, __ATOMIC_RELAXED, SCOPE);
                acc += e;
            }
            if (MODE == 2) __hip_atomic_fetch_add(reinterpret_cast<double*>(&tab[idx[u] * 4 + 2]), 1.0, __ATOMIC_RELAXED, SCOPE);
            if (MODE == 3) acc += tab[idx[u] * 4];
            if (MODE == 4) tab[idx[u] * 4] = s;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int SCOPE>
void run(const char* name, uint32_t* tab, uint64_t slots, uint32_t* sink)
{
    const int blocks = 256 * 8, threads = 256, iters = 256;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, SCOPE>), dim3(blocks), dim3(threads), 0, 0, tab, slots, 16, sink);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((k<MODE, SCOPE>), dim3(blocks), dim3(threads), 0, 0, tab, slots, iters, sink);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double ops = double(blocks) * threads * iters;
    printf("  %-28s %8.2f G ops/s  (%.3f ms)\n", name, ops / ms * 1e-6, ms);
}

int main()
{
    uint32_t* sink;
    hipMalloc(&sink, 64);
    for (uint64_t mb : {4ull, 64ull, 512ull, 2048ull}) {
        const uint64_t bytes = mb << 20, slots = bytes / 16;
        uint32_t* tab;
        hipMalloc(&tab, bytes);
        hipMemset(tab, 0xFF, bytes);
        printf("footprint %llu MiB (16-byte slots)\n", (unsigned long long)mb);
        run<3, __HIP_MEMORY_SCOPE_AGENT>("load", tab, slots, sink);
        run<4, __HIP_MEMORY_SCOPE_AGENT>("store", tab, slots, sink);
        run<0, __HIP_MEMORY_SCOPE_AGENT>("add u32, agent", tab, slots, sink);
        run<0, __HIP_MEMORY_SCOPE_WORKGROUP>("add u32, workgroup", tab, slots, sink);
        run<1, __HIP_MEMORY_SCOPE_AGENT>("cas rtn, agent", tab, slots, sink);
        run<1, __HIP_MEMORY_SCOPE_WORKGROUP>("cas rtn, workgroup", tab, slots, sink);
        run<2, __HIP_MEMORY_SCOPE_AGENT>("add f64, agent", tab, slots, sink);
        run<2, __HIP_MEMORY_SCOPE_WORKGROUP>("add f64, workgroup", tab, slots, sink);
        hipFree(tab);
    }
    return 0;
}
