// chain_bench.hip -- the look-back chain of speck_amd/csrc/chain.hpp alone: nb workgroups publish an aggregate each and
// take the exclusive prefix; kernel duration with and without the chain, result checked.
//   hipcc --offload-arch=gfx950 -O3 -Ispeck_amd/csrc scripts/ubench/chain_bench.hip -o scripts/ubench/chain_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "chain.hpp"
#ifndef VARIANT
#define VARIANT 0
#endif
using namespace speck;

template <bool CHAIN>
__global__ __launch_bounds__(256) void k(Chain ch, u32 nb, u64* out, u32 spin, u64* times)
{
    __shared__ u32 s_mine[kChainWords];
    __shared__ u64 s_pref[kChainWords], s_tmp[2 * kChainWords + 2];
    // some work first, as long for every workgroup (the analysis walks ~17 us)
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < spin) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x < kChainWords) s_mine[threadIdx.x] = blockIdx.x + threadIdx.x;
    __syncthreads();
    if (CHAIN) {
        const u64 tA = __builtin_amdgcn_s_memrealtime();
        chain_publish_own(ch, blockIdx.x, s_mine);
        chain_exclusive(ch, blockIdx.x, nb, s_mine, s_pref, s_tmp);
        const u64 tB = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) { times[2 * blockIdx.x] = tA; times[2 * blockIdx.x + 1] = tB; }
        if (threadIdx.x < kChainPfxWords) out[size_t(blockIdx.x) * kChainWords + threadIdx.x] = s_pref[threadIdx.x];
    }
}

int main()
{
    void* buf;
    hipMalloc(&buf, kChainBytes);
    hipMemset(buf, 0, kChainBytes);
    Chain ch;
    ch.agg = (u64*)buf;
    ch.sup = ch.agg + kChainAggWords;
    ch.error = (u32*)(ch.sup + kChainSupWords);
    u32 launches = 0;
    u64* out; u64* times; hipMalloc(&times, 4096 * 16);
    hipMalloc(&out, 4096 * kChainWords * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    
    for (u32 nb : {64u, 334u, 668u, 1022u, 4085u}) {
        for (int chain = 0; chain < 2; ++chain) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0, 0);
                ch.tag = ++launches;
                if (chain) hipLaunchKernelGGL(k<true>, dim3(nb), dim3(256), 0, 0, ch, nb, out, 1500u, times);
                else hipLaunchKernelGGL(k<false>, dim3(nb), dim3(256), 0, 0, ch, nb, out, 1500u, times);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("nb %4u  %s  %.1f us\n", nb, chain ? "chain   " : "no chain", best * 1e3f);
        }
        { std::vector<u64> tt(2 * nb); hipMemcpy(tt.data(), times, tt.size() * 8, hipMemcpyDeviceToHost);
          u64 tmin = ~0ull, tmax = 0; double sum = 0, mx = 0; for (u32 b = 0; b < nb; ++b) { tmin = tt[2*b] < tmin ? tt[2*b] : tmin; tmax = tt[2*b+1] > tmax ? tt[2*b+1] : tmax; double d = (tt[2*b+1] - tt[2*b]) * 0.01; sum += d; mx = d > mx ? d : mx; }
          u64 pmax = 0; for (u32 b = 0; b < nb; ++b) pmax = tt[2*b] > pmax ? tt[2*b] : pmax;
          printf("  in-kernel: chain avg %.2f us, max %.2f us per workgroup; first publish -> last done %.2f us; publish spread %.2f us\n", sum / nb, mx, (tmax - tmin) * 0.01, (pmax - tmin) * 0.01); }
        std::vector<u64> h(size_t(nb) * kChainWords);
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        bool ok = true;
        for (u32 b = 0; b < nb && ok; ++b)
            for (u32 w = 0; w < kChainPfxWords; ++w) {
                u64 want = 0;
                for (u32 j = 0; j < b; ++j) want += w == 14 ? ((u64(j + 15) << 32) + j + 14) : (w == 15 ? 0 : j + w);
                if (w == 14 ? (h[size_t(b) * kChainWords + 14] + (h[size_t(b) * kChainWords + 15] << 32)) != want : (w != 15 && h[size_t(b) * kChainWords + w] != want)) { ok = false; printf("  MISMATCH block %u word %u: %llu != %llu\n", b, w, (unsigned long long)h[size_t(b) * kChainWords + w], (unsigned long long)want); break; }
            }
        u32 err; hipMemcpy(&err, ch.error, 4, hipMemcpyDeviceToHost);
        printf("  prefixes %s, error word %u\n", ok ? "ok" : "WRONG", err);
    }
    return 0;
}
