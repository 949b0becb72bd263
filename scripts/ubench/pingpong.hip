// pingpong.hip -- one-way latency of a word handed from one workgroup to another through memory, the primitive of the
// look-back chain (speck_amd/csrc/chain.hpp): agent-scope relaxed atomics (sc1) between workgroups on different XCDs
// (blocks b and b + 1), on the same XCD (b and b + 8), and system scope.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/pingpong.hip -o scripts/ubench/pingpong && scripts/ubench/pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int SCOPE>
__global__ void pingpong(uint64_t* a, uint64_t* b, int peer, int iters, uint64_t* ticks)
{
    if (threadIdx.x != 0) return;
    const int me = blockIdx.x;
    if (me != 0 && me != peer) return;
    uint64_t* mine = me == 0 ? a : b;
    uint64_t* theirs = me == 0 ? b : a;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 1; i <= iters; ++i) {
        if (me == 0) {
            __hip_atomic_store(theirs, (uint64_t)i, __ATOMIC_RELAXED, SCOPE);
            while (__hip_atomic_load(mine, __ATOMIC_RELAXED, SCOPE) != (uint64_t)i) __builtin_amdgcn_s_sleep(1);
        } else {
            while (__hip_atomic_load(mine, __ATOMIC_RELAXED, SCOPE) != (uint64_t)i) __builtin_amdgcn_s_sleep(1);
            __hip_atomic_store(theirs, (uint64_t)i, __ATOMIC_RELAXED, SCOPE);
        }
    }
    if (me == 0) *ticks = __builtin_amdgcn_s_memrealtime() - t0;
}

// fan-in: block 0 waits for a word from each of `n` blocks (the look-back's shape): time from "everybody starts" to "all seen"
__global__ void fanin(uint64_t* words, int n, uint64_t tag, uint64_t* ticks)
{
    if (blockIdx.x != 0) {
        if (threadIdx.x == 0 && (int)blockIdx.x <= n)
            __hip_atomic_store(words + blockIdx.x - 1, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        while (__hip_atomic_load(words + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag) __builtin_amdgcn_s_sleep(1);
    __syncthreads();
    if (threadIdx.x == 0) *ticks = __builtin_amdgcn_s_memrealtime() - t0;
}

int main()
{
    uint64_t *d, *t;
    hipMalloc(&d, 1 << 20);
    hipMemset(d, 0, 1 << 20);
    hipHostMalloc(&t, 64);
    const int iters = 2000;
    for (int peer : {1, 8, 9}) {
        hipMemset(d, 0, 4096);
        hipLaunchKernelGGL(pingpong<__HIP_MEMORY_SCOPE_AGENT>, dim3(16), dim3(64), 0, 0, d, d + 64, peer, iters, t);
        hipDeviceSynchronize();
        printf("agent scope, blocks 0 <-> %d: %.0f ns per one-way hop\n", peer, *t * 10.0 / (2.0 * iters));
        hipMemset(d, 0, 4096);
        hipLaunchKernelGGL(pingpong<__HIP_MEMORY_SCOPE_SYSTEM>, dim3(16), dim3(64), 0, 0, d, d + 64, peer, iters, t);
        hipDeviceSynchronize();
        printf("system scope, blocks 0 <-> %d: %.0f ns per one-way hop\n", peer, *t * 10.0 / (2.0 * iters));
    }
    for (int n : {63, 255, 667}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(fanin, dim3(n + 1), dim3(256), 0, 0, d + 1024, n, (uint64_t)(rep + 1 + 10 * n), t);
            hipDeviceSynchronize();
        }
        printf("fan-in of %d workgroups' words into block 0: %.0f ns after block 0 started\n", n, *t * 10.0);
    }
    return 0;
}
