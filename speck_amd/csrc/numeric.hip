// numeric.hip -- numeric phase for gfx950: accumulate the products of every C
// row, compact, sort ascending by column and write C.col_ids / C.data.
// Role of the reference's spGEMMNumericLauncher / denseSpGEMMNumeric /
// hashSpGEMMSortingKernel (include/GPU/spECK_HashSpGEMM.cuh:1714-1794, 1439-1472,
// 1856-1925) and HashMap (include/HashMap.cuh:23-110).  Designed for wave64 + 160 KiB LDS:
//   NUM_DIRECT: A row with one entry -> scaled copy of a (sorted) B row
//   NUM_WAVE  : one wave per row, private 128-entry table, ballot compaction,
//               in-wave rank sort (no workgroup barrier anywhere)
//   NUM_H1    : workgroup per row, 512-entry table, in-place rank sort
//   NUM_H2/H3 : workgroup per row, 2048/8192-entry table; sorted output positions come
//               from a column BITMAP of the (distinct) keys: rank = prefix popcount.
//               O(n + range/32) instead of a comparison sort.
//   NUM_D1/D2 : dense column-window accumulator (fp64 per column + presence bitmap):
//               one ds_add_f64 + one ds_or per product, no probing, output sorted for free.
// The product a*b is rounded first and then added with an LDS atomic (ds_add_f64 /
// ds_add_f32), as the reference does (spECK_HashSpGEMM.cuh:157-165) -- no FMA across the add.
// Algorithmic bytes per row: 8 + 20*lenA + 12*ops + 4 + 12*nnz for fp64 (device_common.hpp).
#include "device_common.hpp"
#include "launch.hpp"

namespace speck {

template <u32 CAP, typename T>
__device__ __forceinline__ void table_accumulate(u32* keys, T* vals, u32 key, T prod)
{
    u32 slot = hash_slot<CAP>(key);
    while (true) {
        const u32 old = atomicCAS(&keys[slot], kEmptyKey, key);
        if (old == kEmptyKey || old == key) break;
        slot = (slot + 1) & (CAP - 1);
    }
    atomicAdd(&vals[slot], prod);
}

// ------------------------------------------------------------------ NUM_DIRECT
template <typename T, int THREADS>
__global__ __launch_bounds__(THREADS) void num_direct_kernel(CsrView<T> A, CsrView<T> B, RowWork w,
                                                             const u32* __restrict__ c_ro,
                                                             u32* __restrict__ c_col,
                                                             T* __restrict__ c_val)
{
    constexpr int NW = THREADS / 64;
    const u32 lane = lane_id(), wid = threadIdx.x >> 6;
    const u32 off = w.st->num_offset[NUM_DIRECT], count = w.st->num_count[NUM_DIRECT];
    const u32 nwaves = gridDim.x * NW;
    for (u32 idx = blockIdx.x * NW + wid; idx < count; idx += nwaves) {
        const u32 row = w.bin_rows[off + idx];
        const u32 ia = A.row_offsets[row];
        const u32 k = A.col_ids[ia];
        const T av = A.data[ia];
        const u32 bs = B.row_offsets[k], be = B.row_offsets[k + 1];
        const u32 base = c_ro[row];
        for (u32 j = lane; j < be - bs; j += 64) {
            c_col[base + j] = B.col_ids[bs + j];
            c_val[base + j] = av * B.data[bs + j];
        }
    }
}

// ------------------------------------------------------------------ NUM_WAVE
template <typename T, int THREADS>
__global__ __launch_bounds__(THREADS) void num_wave_kernel(CsrView<T> A, CsrView<T> B, RowWork w,
                                                           const u32* __restrict__ c_ro,
                                                           u32* __restrict__ c_col,
                                                           T* __restrict__ c_val)
{
    constexpr int NW = THREADS / 64;
    constexpr u32 CAP = kNumWaveCap;
    __shared__ __attribute__((aligned(16))) T s_vals[NW][CAP];
    __shared__ __attribute__((aligned(16))) u32 s_keys[NW][CAP];
    __shared__ __attribute__((aligned(16))) u32 s_ckeys[NW][CAP];
    const u32 lane = lane_id(), wid = threadIdx.x >> 6;
    u32* keys = s_keys[wid];
    T* vals = s_vals[wid];
    u32* ckeys = s_ckeys[wid];
    const u32 off = w.st->num_offset[NUM_WAVE], count = w.st->num_count[NUM_WAVE];
    const u32 nwaves = gridDim.x * NW;
    for (u32 idx = blockIdx.x * NW + wid; idx < count; idx += nwaves) {
        const u32 row = w.bin_rows[off + idx];
        keys[lane] = kEmptyKey;
        keys[lane + 64] = kEmptyKey;
        ckeys[lane] = kEmptyKey;
        ckeys[lane + 64] = kEmptyKey;
        vals[lane] = T(0);
        vals[lane + 64] = T(0);
        const u32 a0 = A.row_offsets[row], a1 = A.row_offsets[row + 1];
        const u32 shift = pick_group_shift(w.row_ops[row], a1 - a0, 0, 6);
        const u32 G = 1u << shift, gl = lane & (G - 1), gsub = lane >> shift, ngroups = 64u >> shift;
        wave_lds_fence();
        for (u32 ia = a0 + gsub; ia < a1; ia += ngroups) {
            const u32 k = A.col_ids[ia];
            const T av = A.data[ia];
            const u32 bs = B.row_offsets[k], be = B.row_offsets[k + 1];
            for (u32 ib = bs + gl; ib < be; ib += G)
                table_accumulate<CAP>(keys, vals, B.col_ids[ib], av * B.data[ib]);
        }
        wave_lds_fence();
        // compaction of the keys (ballot), then rank of each key among the row's keys
        const u32 k0 = keys[lane], k1 = keys[lane + 64];
        const T v0 = vals[lane], v1 = vals[lane + 64];
        const u64 m0 = __ballot(k0 != kEmptyKey), m1 = __ballot(k1 != kEmptyKey);
        const u32 n0 = __popcll(m0), n = n0 + __popcll(m1);
        if (k0 != kEmptyKey) ckeys[__popcll(m0 & lanemask_lt())] = k0;
        if (k1 != kEmptyKey) ckeys[n0 + __popcll(m1 & lanemask_lt())] = k1;
        wave_lds_fence();
        u32 r0 = 0, r1 = 0;
        const uint4* ck4 = reinterpret_cast<const uint4*>(ckeys);
        for (u32 j = 0; j < (n + 3) / 4; ++j) {
            const uint4 q = ck4[j];  // same address in every lane: LDS broadcast
            r0 += (q.x < k0) + (q.y < k0) + (q.z < k0) + (q.w < k0);
            r1 += (q.x < k1) + (q.y < k1) + (q.z < k1) + (q.w < k1);
        }
        const u32 base = c_ro[row];
        if (k0 != kEmptyKey) {
            c_col[base + r0] = k0;
            c_val[base + r0] = v0;
        }
        if (k1 != kEmptyKey) {
            c_col[base + r1] = k1;
            c_val[base + r1] = v1;
        }
        wave_lds_fence();
    }
}

// Shared product loop of the workgroup-per-row kernels.
template <int THREADS, typename T, typename F>
__device__ __forceinline__ void for_each_product(const CsrView<T>& A, const CsrView<T>& B, u32 row,
                                                 u32 ops, F&& f)
{
    constexpr u32 kLog2Threads = __builtin_ctz((u32)THREADS);
    const u32 a0 = A.row_offsets[row], a1 = A.row_offsets[row + 1];
    const u32 shift = pick_group_shift(ops, a1 - a0, 0, kLog2Threads);
    const u32 G = 1u << shift, gl = threadIdx.x & (G - 1), gsub = threadIdx.x >> shift,
              ngroups = (u32)THREADS >> shift;
    for (u32 ia = a0 + gsub; ia < a1; ia += ngroups) {
        const u32 k = A.col_ids[ia];
        const T av = A.data[ia];
        const u32 bs = B.row_offsets[k], be = B.row_offsets[k + 1];
        for (u32 ib = bs + gl; ib < be; ib += G) f(B.col_ids[ib], av * B.data[ib]);
    }
}

// Exclusive prefix of popcounts over bm[0..nwords) into pref[]; returns the total.
// Every thread owns a contiguous run of words. scratch: THREADS/64 + 1 u32.
template <int THREADS>
__device__ __forceinline__ u32 bitmap_prefix(const u32* bm, u32* pref, u32 nwords, u32* scratch)
{
    const u32 wpt = (nwords + THREADS - 1) / THREADS;
    const u32 w_begin = min(threadIdx.x * wpt, nwords), w_end = min(w_begin + wpt, nwords);
    u32 local = 0;
    for (u32 i = w_begin; i < w_end; ++i) local += __popc(bm[i]);
    u32 total;
    u32 run = block_exclusive_scan<THREADS>(local, scratch, &total);
    for (u32 i = w_begin; i < w_end; ++i) {
        pref[i] = run;
        run += __popc(bm[i]);
    }
    __syncthreads();
    return total;
}

// ------------------------------------------------------------------ NUM_H1/H2/H3
// BMW = bitmap words for the bitmap-rank sort (0 -> in-place rank sort).
template <typename T, u32 CAP, int THREADS, u32 BMW>
__global__ __launch_bounds__(THREADS) void num_hash_kernel(CsrView<T> A, CsrView<T> B, RowWork w,
                                                           const u32* __restrict__ c_ro,
                                                           u32* __restrict__ c_col,
                                                           T* __restrict__ c_val, int cls)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* vals = reinterpret_cast<T*>(smem);
    u32* keys = reinterpret_cast<u32*>(vals + CAP);
    u32* bm = keys + CAP;
    u32* pref = bm + BMW;
    u32* scratch = pref + BMW;  // THREADS/64 + 1
    const u32 off = w.st->num_offset[cls], count = w.st->num_count[cls];
    for (u32 idx = blockIdx.x; idx < count; idx += gridDim.x) {
        const u32 row = w.bin_rows[off + idx];
        uint4* keys4 = reinterpret_cast<uint4*>(keys);
        for (u32 i = threadIdx.x; i < CAP / 4; i += THREADS)
            keys4[i] = make_uint4(kEmptyKey, kEmptyKey, kEmptyKey, kEmptyKey);
        for (u32 i = threadIdx.x; i < CAP; i += THREADS) vals[i] = T(0);
        __syncthreads();
        for_each_product<THREADS>(A, B, row, w.row_ops[row],
                                  [&](u32 c, T p) { table_accumulate<CAP>(keys, vals, c, p); });
        __syncthreads();
        const u32 base = c_ro[row];
        if constexpr (BMW == 0) {
            // in-place rank sort: rank = number of table keys smaller than mine
            // (empty slots hold 0xFFFFFFFF and never count)
            const uint4* k4 = reinterpret_cast<const uint4*>(keys);
            for (u32 s = threadIdx.x; s < CAP; s += THREADS) {
                const u32 k = keys[s];
                if (__ballot(k != kEmptyKey) == 0) continue;  // wave-uniform skip
                u32 r = 0;
                for (u32 j = 0; j < CAP / 4; ++j) {
                    const uint4 q = k4[j];
                    r += (q.x < k) + (q.y < k) + (q.z < k) + (q.w < k);
                }
                if (k != kEmptyKey) {
                    c_col[base + r] = k;
                    c_val[base + r] = vals[s];
                }
            }
            __syncthreads();
        } else {
            constexpr u64 kWindowCols = u64(BMW) * 32;
            const u32 cmin = w.row_col_min[row], cmax = w.row_col_max[row];
            u32 emitted = 0;
            for (u64 w0 = cmin; w0 <= cmax; w0 += kWindowCols) {
                const u64 left = u64(cmax) - w0 + 1;
                const u32 ncols = left < kWindowCols ? (u32)left : (u32)kWindowCols;
                const u32 nwords = (ncols + 31) >> 5;
                const u32 wbase = (u32)w0;
                for (u32 i = threadIdx.x; i < nwords; i += THREADS) bm[i] = 0;
                __syncthreads();
                for (u32 s = threadIdx.x; s < CAP; s += THREADS) {
                    const u32 k = keys[s];
                    const u32 d = k - wbase;
                    if (k != kEmptyKey && d < ncols) atomicOr(&bm[d >> 5], 1u << (d & 31));
                }
                __syncthreads();
                const u32 total = bitmap_prefix<THREADS>(bm, pref, nwords, scratch);
                for (u32 s = threadIdx.x; s < CAP; s += THREADS) {
                    const u32 k = keys[s];
                    const u32 d = k - wbase;
                    if (k != kEmptyKey && d < ncols) {
                        const u32 r =
                            emitted + pref[d >> 5] + __popc(bm[d >> 5] & ((1u << (d & 31)) - 1u));
                        c_col[base + r] = k;
                        c_val[base + r] = vals[s];
                    }
                }
                emitted += total;
                __syncthreads();
            }
        }
    }
}

// ------------------------------------------------------------------ NUM_D1/D2
template <typename T, u32 WCOLS, int THREADS>
__global__ __launch_bounds__(THREADS) void num_dense_kernel(CsrView<T> A, CsrView<T> B, RowWork w,
                                                            const u32* __restrict__ c_ro,
                                                            u32* __restrict__ c_col,
                                                            T* __restrict__ c_val, int cls)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr u32 WORDS = WCOLS / 32;
    T* vals = reinterpret_cast<T*>(smem);
    u32* bm = reinterpret_cast<u32*>(vals + WCOLS);
    u32* pref = bm + WORDS;
    u32* scratch = pref + WORDS;
    const u32 off = w.st->num_offset[cls], count = w.st->num_count[cls];
    for (u32 idx = blockIdx.x; idx < count; idx += gridDim.x) {
        const u32 row = w.bin_rows[off + idx];
        const u32 cmin = w.row_col_min[row], cmax = w.row_col_max[row];
        const u32 ops = w.row_ops[row];
        const u32 base = c_ro[row];
        u32 emitted = 0;
        for (u64 w0 = cmin; w0 <= cmax; w0 += WCOLS) {
            const u64 left = u64(cmax) - w0 + 1;
            const u32 ncols = left < WCOLS ? (u32)left : WCOLS;
            const u32 nwords = (ncols + 31) >> 5;
            const u32 wbase = (u32)w0;
            for (u32 i = threadIdx.x; i < ncols; i += THREADS) vals[i] = T(0);
            for (u32 i = threadIdx.x; i < nwords; i += THREADS) bm[i] = 0;
            __syncthreads();
            for_each_product<THREADS>(A, B, row, ops, [&](u32 c, T p) {
                const u32 d = c - wbase;
                if (d < ncols) {
                    atomicAdd(&vals[d], p);
                    atomicOr(&bm[d >> 5], 1u << (d & 31));
                }
            });
            __syncthreads();
            const u32 total = bitmap_prefix<THREADS>(bm, pref, nwords, scratch);
            for (u32 d = threadIdx.x; d < ncols; d += THREADS) {
                const u32 word = bm[d >> 5];
                if (word & (1u << (d & 31))) {
                    const u32 r = emitted + pref[d >> 5] + __popc(word & ((1u << (d & 31)) - 1u));
                    c_col[base + r] = wbase + d;
                    c_val[base + r] = vals[d];
                }
            }
            emitted += total;
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------ launchers
constexpr u32 kNumH2BmWords = 2048;  // 64 Ki columns per sort window
constexpr u32 kNumH3BmWords = 4096;  // 128 Ki columns per sort window

u32 numeric_lds_bytes(int cls, u32 vsize)
{
    switch (cls) {
        case NUM_DIRECT: return 0;
        case NUM_WAVE: return 4 * kNumWaveCap * (vsize + 8);
        case NUM_H1: return kNumH1Cap * (vsize + 4) + 64;
        case NUM_H2: return kNumH2Cap * (vsize + 4) + kNumH2BmWords * 8 + 64;
        case NUM_H3: return kNumH3Cap * (vsize + 4) + kNumH3BmWords * 8 + 128;
        case NUM_D1: return kNumD1Cols * vsize + (kNumD1Cols / 32) * 8 + 64;
        case NUM_D2: return kNumD2Cols * vsize + (kNumD2Cols / 32) * 8 + 128;
    }
    return 0;
}

template <typename K>
static void set_dyn_lds(K kernel, u32 bytes)
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

static u32 grid_for(u32 count, u32 lds, int cu_count, u32 rows_per_block)
{
    u32 per_cu = lds ? (160u * 1024u) / lds : 8;
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    const u64 cap = u64(cu_count) * per_cu * 16;
    u64 need = (u64(count) + rows_per_block - 1) / rows_per_block;
    if (need > cap) need = cap;
    return need ? (u32)need : 1u;
}

template <typename T>
void launch_numeric(hipStream_t s, int cls, u32 count, const CsrView<T>& A, const CsrView<T>& B,
                    const RowWork& w, const u32* c_ro, u32* c_col, T* c_val, u64 /*c_capacity*/,
                    DeviceStats* /*st_mut*/, int cu_count)
{
    if (count == 0) return;
    const u32 lds = numeric_lds_bytes(cls, sizeof(T));
    switch (cls) {
        case NUM_DIRECT: {
            constexpr int TH = 256;
            hipLaunchKernelGGL((num_direct_kernel<T, TH>), dim3(grid_for(count, 0, cu_count, TH / 64)),
                               dim3(TH), 0, s, A, B, w, c_ro, c_col, c_val);
            break;
        }
        case NUM_WAVE: {
            constexpr int TH = 256;
            hipLaunchKernelGGL((num_wave_kernel<T, TH>), dim3(grid_for(count, lds, cu_count, TH / 64)),
                               dim3(TH), 0, s, A, B, w, c_ro, c_col, c_val);
            break;
        }
        case NUM_H1: {
            auto k = num_hash_kernel<T, kNumH1Cap, 256, 0>;
            hipLaunchKernelGGL(k, dim3(grid_for(count, lds, cu_count, 1)), dim3(256), lds, s, A, B, w,
                               c_ro, c_col, c_val, cls);
            break;
        }
        case NUM_H2: {
            auto k = num_hash_kernel<T, kNumH2Cap, 256, kNumH2BmWords>;
            hipLaunchKernelGGL(k, dim3(grid_for(count, lds, cu_count, 1)), dim3(256), lds, s, A, B, w,
                               c_ro, c_col, c_val, cls);
            break;
        }
        case NUM_H3: {
            auto k = num_hash_kernel<T, kNumH3Cap, 512, kNumH3BmWords>;
            set_dyn_lds(k, lds);
            hipLaunchKernelGGL(k, dim3(grid_for(count, lds, cu_count, 1)), dim3(512), lds, s, A, B, w,
                               c_ro, c_col, c_val, cls);
            break;
        }
        case NUM_D1: {
            auto k = num_dense_kernel<T, kNumD1Cols, 256>;
            hipLaunchKernelGGL(k, dim3(grid_for(count, lds, cu_count, 1)), dim3(256), lds, s, A, B, w,
                               c_ro, c_col, c_val, cls);
            break;
        }
        case NUM_D2: {
            auto k = num_dense_kernel<T, kNumD2Cols, 1024>;
            set_dyn_lds(k, lds);
            hipLaunchKernelGGL(k, dim3(grid_for(count, lds, cu_count, 1)), dim3(1024), lds, s, A, B, w,
                               c_ro, c_col, c_val, cls);
            break;
        }
    }
}

template void launch_numeric<double>(hipStream_t, int, u32, const CsrView<double>&,
                                     const CsrView<double>&, const RowWork&, const u32*, u32*,
                                     double*, u64, DeviceStats*, int);
template void launch_numeric<float>(hipStream_t, int, u32, const CsrView<float>&,
                                    const CsrView<float>&, const RowWork&, const u32*, u32*, float*,
                                    u64, DeviceStats*, int);

}  // namespace speck
