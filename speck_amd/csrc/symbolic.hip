// symbolic.hip -- symbolic phase for gfx950: nnz of every C row (distinct columns
// reached by the row; structural, values are never read).
// Role of the reference's spGEMMCountLauncher / denseSpGEMMCount
// (include/GPU/spECK_HashSpGEMM.cuh:1797-1853, 1681-1711) and HashMapNoValue
// (include/HashMap.cuh:136-229); designed for 64-lane waves and 160 KiB of LDS:
//   SYM_WAVE : one wave per row, private 128-key LDS set, no workgroup barrier
//   SYM_H1-3 : one workgroup per row, power-of-two key set (4 / 32 / 128 KiB)
//   SYM_BM1-2: column bitmap (1 bit per column) -- one ds_or per product, no probing;
//              128 KiB of LDS cover 1 Mi columns per window, so the heaviest rows
//              never need a global-memory spill in this phase.
// Algorithmic bytes per row: 8 + 12*lenA + 4*ops + 4 (device_common.hpp).
#include "device_common.hpp"
#include "launch.hpp"

namespace speck {

template <u32 CAP>
__device__ __forceinline__ u32 set_insert(u32* tab, u32 key)
{
    u32 slot = hash_slot<CAP>(key);
    while (true) {
        const u32 old = atomicCAS(&tab[slot], kEmptyKey, key);
        if (old == kEmptyKey) return 1;
        if (old == key) return 0;
        slot = (slot + 1) & (CAP - 1);
    }
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void sym_wave_kernel(const u32* __restrict__ a_ro,
                                                           const u32* __restrict__ a_col,
                                                           const u32* __restrict__ b_ro,
                                                           const u32* __restrict__ b_col, RowWork w,
                                                           u32* __restrict__ counts)
{
    constexpr int NW = THREADS / 64;
    __shared__ u32 s_tab[NW][kSymWaveCap];
    const u32 lane = lane_id(), wid = threadIdx.x >> 6;
    u32* tab = s_tab[wid];
    const u32 off = w.st->sym_offset[SYM_WAVE], count = w.st->sym_count[SYM_WAVE];
    const u32 nwaves = gridDim.x * NW;
    for (u32 idx = blockIdx.x * NW + wid; idx < count; idx += nwaves) {
        const u32 row = w.bin_rows[off + idx];
        tab[lane] = kEmptyKey;
        tab[lane + 64] = kEmptyKey;
        const u32 a0 = a_ro[row], a1 = a_ro[row + 1];
        const u32 shift = pick_group_shift(w.row_ops[row], a1 - a0, 0, 6);
        const u32 G = 1u << shift, gl = lane & (G - 1), gsub = lane >> shift, ngroups = 64u >> shift;
        wave_lds_fence();
        u32 cnt = 0;
        for (u32 ia = a0 + gsub; ia < a1; ia += ngroups) {
            const u32 k = a_col[ia];
            const u32 bs = b_ro[k], be = b_ro[k + 1];
            for (u32 ib = bs + gl; ib < be; ib += G) cnt += set_insert<kSymWaveCap>(tab, b_col[ib]);
        }
        cnt = wave_reduce_add(cnt);
        if (lane == 0) counts[row] = cnt;
        wave_lds_fence();
    }
}

template <u32 CAP, int THREADS>
__global__ __launch_bounds__(THREADS) void sym_hash_kernel(const u32* __restrict__ a_ro,
                                                           const u32* __restrict__ a_col,
                                                           const u32* __restrict__ b_ro,
                                                           const u32* __restrict__ b_col, RowWork w,
                                                           u32* __restrict__ counts, int cls)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32* tab = reinterpret_cast<u32*>(smem);
    u32* s_cnt = tab + CAP;
    constexpr u32 kLog2Threads = __builtin_ctz((u32)THREADS);
    const u32 off = w.st->sym_offset[cls], count = w.st->sym_count[cls];
    for (u32 idx = blockIdx.x; idx < count; idx += gridDim.x) {
        const u32 row = w.bin_rows[off + idx];
        uint4* tab4 = reinterpret_cast<uint4*>(tab);
        for (u32 i = threadIdx.x; i < CAP / 4; i += THREADS)
            tab4[i] = make_uint4(kEmptyKey, kEmptyKey, kEmptyKey, kEmptyKey);
        if (threadIdx.x == 0) *s_cnt = 0;
        __syncthreads();
        const u32 a0 = a_ro[row], a1 = a_ro[row + 1];
        const u32 shift = pick_group_shift(w.row_ops[row], a1 - a0, 0, kLog2Threads);
        const u32 G = 1u << shift, gl = threadIdx.x & (G - 1), gsub = threadIdx.x >> shift,
                  ngroups = (u32)THREADS >> shift;
        u32 cnt = 0;
        for (u32 ia = a0 + gsub; ia < a1; ia += ngroups) {
            const u32 k = a_col[ia];
            const u32 bs = b_ro[k], be = b_ro[k + 1];
            for (u32 ib = bs + gl; ib < be; ib += G) cnt += set_insert<CAP>(tab, b_col[ib]);
        }
        cnt = wave_reduce_add(cnt);
        if (lane_id() == 0 && cnt) atomicAdd(s_cnt, cnt);
        __syncthreads();
        if (threadIdx.x == 0) counts[row] = *s_cnt;
        __syncthreads();
    }
}

template <u32 WORDS, int THREADS>
__global__ __launch_bounds__(THREADS) void sym_bitmap_kernel(const u32* __restrict__ a_ro,
                                                             const u32* __restrict__ a_col,
                                                             const u32* __restrict__ b_ro,
                                                             const u32* __restrict__ b_col,
                                                             RowWork w, u32* __restrict__ counts,
                                                             int cls)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32* bm = reinterpret_cast<u32*>(smem);
    u32* s_cnt = bm + WORDS;
    constexpr u32 kLog2Threads = __builtin_ctz((u32)THREADS);
    constexpr u64 kWindowCols = u64(WORDS) * 32;
    const u32 off = w.st->sym_offset[cls], count = w.st->sym_count[cls];
    for (u32 idx = blockIdx.x; idx < count; idx += gridDim.x) {
        const u32 row = w.bin_rows[off + idx];
        const u32 a0 = a_ro[row], a1 = a_ro[row + 1];
        const u32 cmin = w.row_col_min[row], cmax = w.row_col_max[row];
        const u32 shift = pick_group_shift(w.row_ops[row], a1 - a0, 0, kLog2Threads);
        const u32 G = 1u << shift, gl = threadIdx.x & (G - 1), gsub = threadIdx.x >> shift,
                  ngroups = (u32)THREADS >> shift;
        if (threadIdx.x == 0) *s_cnt = 0;
        u32 total = 0;
        for (u64 w0 = cmin; w0 <= cmax; w0 += kWindowCols) {
            const u64 left = u64(cmax) - w0 + 1;
            const u32 ncols = left < kWindowCols ? (u32)left : (u32)kWindowCols;
            const u32 nwords = (ncols + 31) >> 5;
            for (u32 i = threadIdx.x; i < nwords; i += THREADS) bm[i] = 0;
            __syncthreads();
            const u32 base = (u32)w0;
            for (u32 ia = a0 + gsub; ia < a1; ia += ngroups) {
                const u32 k = a_col[ia];
                const u32 bs = b_ro[k], be = b_ro[k + 1];
                for (u32 ib = bs + gl; ib < be; ib += G) {
                    const u32 d = b_col[ib] - base;  // wraps to a huge value when left of the window
                    if (d < ncols) atomicOr(&bm[d >> 5], 1u << (d & 31));
                }
            }
            __syncthreads();
            for (u32 i = threadIdx.x; i < nwords; i += THREADS) total += __popc(bm[i]);
            __syncthreads();
        }
        total = wave_reduce_add(total);
        if (lane_id() == 0 && total) atomicAdd(s_cnt, total);
        __syncthreads();
        if (threadIdx.x == 0) counts[row] = *s_cnt;
        __syncthreads();
    }
}

u32 symbolic_lds_bytes(int cls)
{
    switch (cls) {
        case SYM_WAVE: return 4 * kSymWaveCap * 4;
        case SYM_H1: return kSymH1Cap * 4 + 16;
        case SYM_H2: return kSymH2Cap * 4 + 16;
        case SYM_H3: return kSymH3Cap * 4 + 16;
        case SYM_BM1: return kSymBm1Words * 4 + 16;
        case SYM_BM2: return kSymBm2Words * 4 + 16;
    }
    return 0;
}

template <typename K>
static void set_dyn_lds(K kernel, u32 bytes)
{
    // kernels above 64 KiB of LDS need the opt-in (a workgroup may own all 160 KiB on gfx950)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

static u32 grid_for(u32 count, u32 lds, int cu_count, u32 rows_per_block)
{
    u32 per_cu = lds ? (160u * 1024u) / lds : 8;
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    const u64 cap = u64(cu_count) * per_cu * 16;  // 16 rounds of resident workgroups, then stride
    u64 need = (u64(count) + rows_per_block - 1) / rows_per_block;
    if (need > cap) need = cap;
    return need ? (u32)need : 1u;
}

void launch_symbolic(hipStream_t s, int cls, u32 count, const u32* a_ro, const u32* a_col,
                     const u32* b_ro, const u32* b_col, const RowWork& w, u32* counts, int cu_count)
{
    if (count == 0) return;
    const u32 lds = symbolic_lds_bytes(cls);
    switch (cls) {
        case SYM_WAVE: {
            constexpr int T = 256;
            hipLaunchKernelGGL(sym_wave_kernel<T>, dim3(grid_for(count, lds, cu_count, T / 64)),
                               dim3(T), 0, s, a_ro, a_col, b_ro, b_col, w, counts);
            break;
        }
        case SYM_H1: {
            auto k = sym_hash_kernel<kSymH1Cap, 256>;
            hipLaunchKernelGGL(k, dim3(grid_for(count, lds, cu_count, 1)), dim3(256), lds, s, a_ro,
                               a_col, b_ro, b_col, w, counts, cls);
            break;
        }
        case SYM_H2: {
            auto k = sym_hash_kernel<kSymH2Cap, 512>;
            hipLaunchKernelGGL(k, dim3(grid_for(count, lds, cu_count, 1)), dim3(512), lds, s, a_ro,
                               a_col, b_ro, b_col, w, counts, cls);
            break;
        }
        case SYM_H3: {
            auto k = sym_hash_kernel<kSymH3Cap, 1024>;
            set_dyn_lds(k, lds);
            hipLaunchKernelGGL(k, dim3(grid_for(count, lds, cu_count, 1)), dim3(1024), lds, s, a_ro,
                               a_col, b_ro, b_col, w, counts, cls);
            break;
        }
        case SYM_BM1: {
            auto k = sym_bitmap_kernel<kSymBm1Words, 256>;
            hipLaunchKernelGGL(k, dim3(grid_for(count, lds, cu_count, 1)), dim3(256), lds, s, a_ro,
                               a_col, b_ro, b_col, w, counts, cls);
            break;
        }
        case SYM_BM2: {
            auto k = sym_bitmap_kernel<kSymBm2Words, 1024>;
            set_dyn_lds(k, lds);
            hipLaunchKernelGGL(k, dim3(grid_for(count, lds, cu_count, 1)), dim3(1024), lds, s, a_ro,
                               a_col, b_ro, b_col, w, counts, cls);
            break;
        }
    }
}

}  // namespace speck
