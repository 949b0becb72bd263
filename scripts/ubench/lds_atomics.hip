// Microbenchmark: LDS atomic throughput on gfx950 (ops per cycle per CU), to size the SpGEMM
// accumulators.  hipcc --offload-arch=gfx950 -O3 lds_atomics.hip -o lds_atomics && ./lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

enum Op { OR_B32, ADD_U32_RTN, CAS_B32, ADD_F32, ADD_F64, ADD_F64_RTN, MAX_U32, ADD_U64,
          // the numeric hash kernels' own mix: K returning compare-and-swaps (u32 keys) per fp64 add (round 6: the ceiling
          // bench.py prices SQ_INSTS_LDS_ATOMIC against is the mix's, not the slower op's)
          MIX_1CAS_1ADD, MIX_3CAS_1ADD, MIX_7CAS_1ADD };
enum Pat { LINEAR, RANDOM, SAME, RANDOM_SMALL };

template <int OP, int PAT>
__global__ __launch_bounds__(256) void k(int iters, unsigned* sink)
{
    __shared__ __attribute__((aligned(16))) double lds_d[4096];
    unsigned* lds_u = reinterpret_cast<unsigned*>(lds_d);
    float* lds_f = reinterpret_cast<float*>(lds_d);
    unsigned long long* lds_ull = reinterpret_cast<unsigned long long*>(lds_d);
    for (int i = threadIdx.x; i < 4096; i += 256) lds_d[i] = 0.0;
    __syncthreads();
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            unsigned idx;
            if (PAT == LINEAR) idx = (threadIdx.x + (it * 8 + u) * 256) & 4095;
            else if (PAT == SAME) idx = (it * 8 + u) & 4095;
            else {
                x = x * 1664525u + 1013904223u;
                idx = (x >> 12) & (PAT == RANDOM ? 4095 : 255);
            }
            if (OP == OR_B32) atomicOr(&lds_u[idx], 1u << (x & 31));
            if (OP == ADD_U32_RTN) acc += atomicAdd(&lds_u[idx], 1u);
            if (OP == CAS_B32) acc += atomicCAS(&lds_u[idx], 0u, x | 1u);
            if (OP == ADD_F32) atomicAdd(&lds_f[idx], 1.0f);
            if (OP == ADD_F64) atomicAdd(&lds_d[idx], 1.0);
            if (OP == ADD_F64_RTN) acc += (unsigned)atomicAdd(&lds_d[idx], 1.0);
            if (OP == MAX_U32) atomicMax(&lds_u[idx], x);
            if (OP == ADD_U64) atomicAdd(&lds_ull[idx], 1ull);
            if (OP == MIX_1CAS_1ADD || OP == MIX_3CAS_1ADD || OP == MIX_7CAS_1ADD) {
                const int period = OP == MIX_1CAS_1ADD ? 2 : (OP == MIX_3CAS_1ADD ? 4 : 8);
                // (keys in the upper half of the array, values in the lower: as the tables of numeric.hip)
                if (u % period == period - 1) atomicAdd(&lds_d[idx & 2047], 1.0);
                else acc += atomicCAS(&lds_u[4096 + idx], 0u, x | 1u);
            }
        }
    }
    __syncthreads();
    if (acc == 0x12345678u || lds_u[threadIdx.x] == 0xdeadbeefu) sink[0] = acc;
}

template <int OP, int PAT>
void run(const char* name, unsigned* sink)
{
    const int blocks = 256 * 8, iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((k<OP, PAT>), dim3(blocks), dim3(256), 0, 0, 10, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<OP, PAT>), dim3(blocks), dim3(256), 0, 0, iters, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double ops = double(blocks) * 256 * iters * 8;
    double per_cu_per_cycle = ops / (ms * 1e-3) / 256 / 2.4e9;
    std::printf("%-28s %8.3f ms  %8.1f Gops/s  %6.3f lanes/cycle/CU (@2.4GHz)  %6.1f cyc/wave-instr/CU\n", name, ms,
                ops / ms / 1e6, per_cu_per_cycle, 64.0 / per_cu_per_cycle);
}

int main()
{
    unsigned* sink;
    hipMalloc(&sink, 4);
#define R(OP, PAT) run<OP, PAT>(#OP " " #PAT, sink)
    R(OR_B32, LINEAR); R(OR_B32, RANDOM); R(OR_B32, RANDOM_SMALL); R(OR_B32, SAME);
    R(ADD_U32_RTN, LINEAR); R(ADD_U32_RTN, RANDOM);
    R(CAS_B32, LINEAR); R(CAS_B32, RANDOM); R(CAS_B32, RANDOM_SMALL);
    R(MAX_U32, RANDOM);
    R(ADD_F32, LINEAR); R(ADD_F32, RANDOM);
    R(ADD_F64, LINEAR); R(ADD_F64, RANDOM); R(ADD_F64, RANDOM_SMALL); R(ADD_F64, SAME);
    R(ADD_F64_RTN, RANDOM);
    R(ADD_U64, LINEAR); R(ADD_U64, RANDOM);
    R(MIX_1CAS_1ADD, RANDOM); R(MIX_3CAS_1ADD, RANDOM); R(MIX_7CAS_1ADD, RANDOM);
    return 0;
}
