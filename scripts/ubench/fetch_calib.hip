// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the SpGEMM
// kernels use (4 B and 8 B per lane, coalesced), against a known byte count (1 GiB > L3).
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./fetch_calib ; same with WRITE_SIZE
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename V>
__global__ void rd(const V* __restrict__ p, size_t n, V* sink)
{
    V acc = V(0);
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) acc += p[i];
    if (acc == V(12345)) sink[0] = acc;
}
template <typename V>
__global__ void wr(V* __restrict__ p, size_t n)
{
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) p[i] = V(i);
}
__global__ void rd16(const float4* __restrict__ p, size_t n, float4* sink)
{
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        float4 v = p[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x == 12345.f) sink[0] = acc;
}
// gather patterns of the SpGEMM kernels: single random elements, and short runs (a B row of ~6 entries)
// at random places -- which request sizes does the L2 send to the fabric for them?
template <typename V, int RUN>
__global__ void gather(const V* __restrict__ p, size_t n, size_t gathers, V* sink)
{
    V acc = V(0);
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < gathers; i += size_t(gridDim.x) * blockDim.x) {
        unsigned long long h = (i / RUN) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        h *= 0xBF58476D1CE4E5B9ull;
        h ^= h >> 32;
        acc += p[(h % (n - RUN)) + i % RUN];
    }
    if (acc == V(12345)) sink[0] = acc;
}
int main()
{
    const size_t bytes = size_t(1) << 30;
    void *a, *b;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes);
    hipMemset(b, 0, bytes);
    hipLaunchKernelGGL(rd<unsigned>, dim3(4096), dim3(256), 0, 0, (const unsigned*)a, bytes / 4, (unsigned*)b);
    hipLaunchKernelGGL(rd<double>, dim3(4096), dim3(256), 0, 0, (const double*)a, bytes / 8, (double*)b);
    hipLaunchKernelGGL(rd16, dim3(4096), dim3(256), 0, 0, (const float4*)a, bytes / 16, (float4*)b);
    hipLaunchKernelGGL(wr<unsigned>, dim3(4096), dim3(256), 0, 0, (unsigned*)b, bytes / 4);
    hipLaunchKernelGGL(wr<double>, dim3(4096), dim3(256), 0, 0, (double*)b, bytes / 8);
    const size_t g = size_t(1) << 24;  // 16 Mi gathers over 1 GiB: practically every one a miss
    hipLaunchKernelGGL((gather<unsigned, 1>), dim3(4096), dim3(256), 0, 0, (const unsigned*)a, bytes / 4, g, (unsigned*)b);
    hipLaunchKernelGGL((gather<double, 1>), dim3(4096), dim3(256), 0, 0, (const double*)a, bytes / 8, g, (double*)b);
    hipLaunchKernelGGL((gather<unsigned, 6>), dim3(4096), dim3(256), 0, 0, (const unsigned*)a, bytes / 4, g, (unsigned*)b);
    hipLaunchKernelGGL((gather<double, 6>), dim3(4096), dim3(256), 0, 0, (const double*)a, bytes / 8, g, (double*)b);
    hipDeviceSynchronize();
    std::printf("each streaming kernel moves %zu bytes; each gather kernel issues %zu element loads\n", bytes, g);
    return 0;
}
