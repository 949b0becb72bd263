// chain_probe.hip -- what costs time in a look-back chain on MI355X: stores only / stores + one poll, 1 or 24 words,
// transposed or block-major descriptors.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64; typedef uint32_t u32;

__global__ __launch_bounds__(256) void k(u64* buf, u32 tag, int words, int layout, int poll, int scope_sys)
{
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 1500) __builtin_amdgcn_s_sleep(4);
    const u32 b = blockIdx.x, t = threadIdx.x, sb = b / 64, j = b % 64;
    auto idx = [&](u32 blk, u32 w) -> size_t {
        return layout == 0 ? (size_t(blk / 64) * 24 + w) * 64 + blk % 64 : size_t(blk) * 32 + w;   // block-major: 256 B per block
    };
    if ((int)t < words) {
        if (scope_sys) __hip_atomic_store(buf + idx(b, t), (u64(tag) << 32) | b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else __hip_atomic_store(buf + idx(b, t), (u64(tag) << 32) | b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (poll && t < 64 && t < j) {
        // poll == 1: word 0 of every predecessor; poll == 16: ALL `words` words of every predecessor
        const int nw = poll > 1 ? words : 1;
        for (int w = 0; w < nw; ++w) {
            const u64* p = buf + idx(sb * 64 + t, w);
            u32 guard = 0;
            while ((u32)(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) != tag && ++guard < (1u << 20)) __builtin_amdgcn_s_sleep(1);
        }
    }
}

int main()
{
    u64* buf;
    hipMalloc(&buf, 64 << 20);
    hipMemset(buf, 0, 64 << 20);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    u32 tag = 0;
    for (u32 nb : {64u, 668u}) {
        for (int words : {1, 8, 24})
            for (int layout : {0, 1})
                for (int poll : {0, 1, 16})
                    for (int sys : {0}) {
                        float best = 1e9f;
                        for (int rep = 0; rep < 5; ++rep) {
                            ++tag;
                            hipEventRecord(e0, 0);
                            hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, buf, tag, words, layout, poll, sys);
                            hipEventRecord(e1, 0);
                            hipEventSynchronize(e1);
                            float ms; hipEventElapsedTime(&ms, e0, e1);
                            if (ms < best) best = ms;
                        }
                        printf("nb %4u words %2d layout %s poll(sleep) %2d store-scope %s: %.1f us\n", nb, words, layout ? "block-major" : "transposed ", poll, sys ? "system" : "agent ", best * 1e3f);
                    }
    }
    return 0;
}
