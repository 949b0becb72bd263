// numeric.hip -- numeric phase for gfx950: accumulate the products of every C row, then
// write the row's column ids ascending with their values.
// Role of the reference's spGEMMNumericLauncher / denseSpGEMMNumeric / hashSpGEMMSortingKernel
// (include/GPU/spECK_HashSpGEMM.cuh:1714-1794, 1439-1472, 1856-1925) and HashMap
// (include/HashMap.cuh:23-110).  Designed for wave64 + 160 KiB LDS:
//   NUM_DIRECT : A row with one entry -> scaled copy of a (sorted) B row, 16 lanes per row
//   NUM_G16    : 16 lanes per row (4 rows per wave), 64-entry table, rank sort
//   NUM_W128   : one wave per row, 128-entry table, ballot compaction + rank sort
//   NUM_W512   : one wave per row, 512-entry table, two-level bitmap sort
//   NUM_B2K/B8K: one workgroup per row, 2048/8192-entry table, two-level bitmap sort
//   NUM_D1/D2  : dense column-window accumulator (value per column + presence bitmap):
//                one ds_add_f64 + one ds_or per product, no probing, output sorted for free.
// Two-level bitmap sort: the keys of a row are DISTINCT, so their sorted position is a prefix
// popcount.  Level 1 marks the occupied 32-column buckets (range/32 bits), its prefix ranks
// the occupied buckets; level 2 holds one 32-bit mask per OCCUPIED bucket (<= nnz words).
// O(nnz + range/1024) LDS operations instead of a comparison sort (the reference uses an
// O(nnz^2) rank sort below 500 entries and cub::BlockRadixSort above, :813-865, 1856-1925).
// The sort scratch aliases the hash table: by then every lane holds its slots in registers.
// The product a*b is rounded first and then added with an LDS atomic (ds_add_f64), as the
// reference does (spECK_HashSpGEMM.cuh:157-165) -- no FMA across the add.
// Algorithmic bytes per row: 8 + 20*lenA + 12*ops + 4 + 12*nnz for fp64 (device_common.hpp).
#include "device_common.hpp"
#include "launch.hpp"
#include "row_groups.hpp"

namespace speck {

// ------------------------------------------------------------------ NUM_DIRECT
template <typename T, int THREADS>
__global__ __launch_bounds__(THREADS) void num_direct_kernel(ProductSrc<T> src, const u32* a_ro, RowWork w,
                                                             u32* __restrict__ c_col,
                                                             T* __restrict__ c_val)
{
    constexpr u32 L = 16, NG = THREADS / L;
    if (w.st->capacity_miss) return;
    src.rebase(a_ro);
    const u32 lane = threadIdx.x & (L - 1), gid = threadIdx.x / L;
    const u32 count = w.st->num.count[NUM_DIRECT];
    const RowRec* recs = w.recs + w.st->num.offset[NUM_DIRECT];
    for (u32 idx = blockIdx.x * NG + gid; idx < count; idx += gridDim.x * NG) {
        const RowRec rec = recs[idx];
        const T av = src.a_val[rec.a0];
        const u32 bs = src.b_start[rec.a0];
        for (u32 j = lane; j < rec.nnz; j += L) {
            c_col[rec.base + j] = src.b_col[bs + j];
            c_val[rec.base + j] = av * src.b_val[bs + j];
        }
    }
}

// ------------------------------------------------------------------ sorting back-ends
// Rank sort for tiny tables: every lane owns OWN = CAP/SIZE slots (registers).
// `ckeys` may alias the table: all slots are in registers before the first write.
template <class G, typename T, u32 CAP>
__device__ __forceinline__ void emit_rank_sorted(const G& g, const u32* keys, const T* vals,
                                                 u32* ckeys, u32 base, u32* __restrict__ c_col,
                                                 T* __restrict__ c_val)
{
    constexpr u32 OWN = CAP / G::SIZE;
    u32 k[OWN];
    T v[OWN];
#pragma unroll
    for (u32 j = 0; j < OWN; ++j) {
        k[j] = keys[j * G::SIZE + g.lane];
        v[j] = vals[j * G::SIZE + g.lane];
    }
    g.sync();
    u32 run = 0;
    const u64 lt = (1ull << g.lane) - 1ull;
#pragma unroll
    for (u32 j = 0; j < OWN; ++j) {
        const u64 mask = g.ballot(k[j] != kEmptyKey);
        if (k[j] != kEmptyKey) ckeys[run + __popcll(mask & lt)] = k[j];
        run += __popcll(mask);
    }
    if (g.lane < 4) ckeys[run + g.lane] = kEmptyKey;  // pad the last uint4
    g.sync();
    u32 r[OWN];
#pragma unroll
    for (u32 j = 0; j < OWN; ++j) r[j] = 0;
    const uint4* ck4 = reinterpret_cast<const uint4*>(ckeys);
    for (u32 q = 0; q < (run + 3) / 4; ++q) {
        const uint4 x = ck4[q];  // same address for the whole group: LDS broadcast
#pragma unroll
        for (u32 j = 0; j < OWN; ++j) r[j] += (x.x < k[j]) + (x.y < k[j]) + (x.z < k[j]) + (x.w < k[j]);
    }
#pragma unroll
    for (u32 j = 0; j < OWN; ++j)
        if (k[j] != kEmptyKey) {
            c_col[base + r[j]] = k[j];
            c_val[base + r[j]] = v[j];
        }
}

// Two-level bitmap sort (see the header comment).  S: LDS scratch of max(2*W1, 2*NMAX) words;
// it may alias the table (slots are loaded into registers first).
template <class G, typename T, u32 CAP, u32 W1, u32 NMAX>
__device__ __forceinline__ void emit_bitmap_sorted(const G& g, const u32* keys, const T* vals, u32* S,
                                                   u32* scan_scratch, u32 cmin, u32 cmax, u32 base,
                                                   u32* __restrict__ c_col, T* __restrict__ c_val)
{
    constexpr u32 OWN = CAP / G::SIZE;
    constexpr u64 kWindowCols = u64(W1) * 1024;
    u32 k[OWN], brank[OWN];
    T v[OWN];
#pragma unroll
    for (u32 j = 0; j < OWN; ++j) {
        k[j] = keys[j * G::SIZE + g.lane];
        v[j] = vals[j * G::SIZE + g.lane];
        brank[j] = 0;
    }
    g.sync();
    u32* l1 = S;
    u32* l1pref = S + W1;
    u32* masks = S;
    u32* mpref = S + NMAX;
    u32 emitted = 0;
    for (u64 w0 = cmin; w0 <= cmax; w0 += kWindowCols) {
        const u64 left = u64(cmax) - w0 + 1;
        const u32 ncols = left < kWindowCols ? (u32)left : (u32)kWindowCols;
        const u32 nw1 = (((ncols + 31) >> 5) + 31) >> 5;
        const u32 wbase = (u32)w0;
        for (u32 i = g.lane; i < nw1; i += G::SIZE) l1[i] = 0;
        g.sync();
#pragma unroll
        for (u32 j = 0; j < OWN; ++j) {
            const u32 d = k[j] - wbase;
            if (k[j] != kEmptyKey && d < ncols) atomicOr(&l1[d >> 10], 1u << ((d >> 5) & 31));
        }
        g.sync();
        const u32 nocc = bitmap_prefix(g, l1, l1pref, nw1, scan_scratch);
#pragma unroll
        for (u32 j = 0; j < OWN; ++j) {
            const u32 d = k[j] - wbase;
            if (k[j] != kEmptyKey && d < ncols)
                brank[j] = l1pref[d >> 10] + __popc(l1[d >> 10] & ((1u << ((d >> 5) & 31)) - 1u));
        }
        g.sync();  // level-1 arrays are dead from here: the masks alias them
        for (u32 i = g.lane; i < nocc; i += G::SIZE) masks[i] = 0;
        g.sync();
#pragma unroll
        for (u32 j = 0; j < OWN; ++j) {
            const u32 d = k[j] - wbase;
            if (k[j] != kEmptyKey && d < ncols) atomicOr(&masks[brank[j]], 1u << (d & 31));
        }
        g.sync();
        const u32 total = bitmap_prefix(g, masks, mpref, nocc, scan_scratch);
#pragma unroll
        for (u32 j = 0; j < OWN; ++j) {
            const u32 d = k[j] - wbase;
            if (k[j] != kEmptyKey && d < ncols) {
                const u32 r = emitted + mpref[brank[j]] +
                              __popc(masks[brank[j]] & ((1u << (d & 31)) - 1u));
                c_col[base + r] = k[j];
                c_val[base + r] = v[j];
            }
        }
        emitted += total;
        g.sync();
    }
}

// ------------------------------------------------------------------ hash kernels
enum SortMode { SORT_RANK = 0, SORT_BITMAP = 1 };

template <class G, int THREADS>
constexpr u32 scan_scratch_words()
{
    return G::kIsBlock ? (THREADS / 64 + 2) : 0;
}
// LDS bytes of one group, 16-byte granular: table (values, keys) | a_ik | prefix | offsets | scan
template <class G, typename T, u32 CAP, int THREADS>
constexpr u32 num_group_lds()
{
    const u32 words = 2 * G::SIZE + scan_scratch_words<G, THREADS>();
    return CAP * ((u32)sizeof(T) + 4u) + G::SIZE * (u32)sizeof(T) + (words + 3u) / 4u * 16u;
}

template <class G, typename T, u32 CAP, u32 W1, u32 NMAX, int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void num_hash_kernel(ProductSrc<T> src, const u32* a_ro, RowWork w,
                                                           u32* __restrict__ c_col,
                                                           T* __restrict__ c_val, int cls)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr u32 NG = THREADS / G::SIZE;
    constexpr u32 kGroupBytes = num_group_lds<G, T, CAP, THREADS>();
    // the sort scratch (rank: NMAX+8 words, bitmap: max(2*W1, 2*NMAX) words) fits in the table
    static_assert((MODE == SORT_RANK ? NMAX + 8 : (2 * W1 > 2 * NMAX ? 2 * W1 : 2 * NMAX)) * 4 <=
                      CAP * (sizeof(T) + 4),
                  "sort scratch must fit in the table it aliases");
    const G g;
    const u32 gid = G::kIsBlock ? 0u : threadIdx.x / G::SIZE;
    unsigned char* mine = smem + gid * kGroupBytes;
    T* vals = reinterpret_cast<T*>(mine);
    u32* keys = reinterpret_cast<u32*>(vals + CAP);
    T* m_av = reinterpret_cast<T*>(keys + CAP);
    u32* m_incl = reinterpret_cast<u32*>(m_av + G::SIZE);
    RowMeta<T> meta{m_incl, m_incl + G::SIZE, m_av};
    u32* scan_scratch = m_incl + 2 * G::SIZE;
    u32* S = reinterpret_cast<u32*>(mine);
    if (w.st->capacity_miss) return;
    src.rebase(a_ro);
    const u32 count = w.st->num.count[cls];
    const RowRec* recs = w.recs + w.st->num.offset[cls];
    u32 idx = blockIdx.x * NG + gid;
    const u32 stride = gridDim.x * NG;
    RowRec next{};
    if (idx < count) next = recs[idx];
    while (idx < count) {
        const RowRec rec = next;  // fetched while the previous row was being processed
        if (idx + stride < count) next = recs[idx + stride];
        for (u32 i = g.lane; i < CAP; i += G::SIZE) {
            keys[i] = kEmptyKey;
            vals[i] = T(0);
        }
        g.sync();
        for_each_product<true>(g, src, rec.a0, rec.a1, meta, scan_scratch,
                               [&](u32 c, T p) { table_accumulate<CAP>(keys, vals, c, p); });
        if constexpr (MODE == SORT_RANK) {
            emit_rank_sorted<G, T, CAP>(g, keys, vals, S, rec.base, c_col, c_val);
        } else {
            emit_bitmap_sorted<G, T, CAP, W1, NMAX>(g, keys, vals, S, scan_scratch, rec.cmin, rec.cmax,
                                                    rec.base, c_col, c_val);
        }
        g.sync();
        idx += stride;
    }
}

// ------------------------------------------------------------------ NUM_D1/D2
template <typename T, u32 WCOLS, int THREADS>
constexpr u32 num_dense_lds()
{
    return (WCOLS + THREADS) * (u32)sizeof(T) + (2 * (WCOLS / 32) + 2 * THREADS + THREADS / 64 + 2 + 3) / 4 * 16;
}

template <typename T, u32 WCOLS, int THREADS>
__global__ __launch_bounds__(THREADS) void num_dense_kernel(ProductSrc<T> src, const u32* a_ro, RowWork w,
                                                            u32* __restrict__ c_col,
                                                            T* __restrict__ c_val, int cls)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr u32 WORDS = WCOLS / 32;
    using G = Block<THREADS>;
    const G g;
    T* vals = reinterpret_cast<T*>(smem);
    T* m_av = vals + WCOLS;
    u32* bm = reinterpret_cast<u32*>(m_av + THREADS);
    u32* pref = bm + WORDS;
    RowMeta<T> meta{pref + WORDS, pref + WORDS + THREADS, m_av};
    u32* scratch = pref + WORDS + 2 * THREADS;
    if (w.st->capacity_miss) return;
    src.rebase(a_ro);
    const u32 count = w.st->num.count[cls];
    const RowRec* recs = w.recs + w.st->num.offset[cls];
    RowRec next{};
    if (blockIdx.x < count) next = recs[blockIdx.x];
    for (u32 idx = blockIdx.x; idx < count; idx += gridDim.x) {
        const RowRec rec = next;  // fetched while the previous row was being processed
        if (idx + gridDim.x < count) next = recs[idx + gridDim.x];
        u32 emitted = 0;
        for (u64 w0 = rec.cmin; w0 <= rec.cmax; w0 += WCOLS) {
            const u64 left = u64(rec.cmax) - w0 + 1;
            const u32 ncols = left < WCOLS ? (u32)left : WCOLS;
            const u32 nwords = (ncols + 31) >> 5;
            const u32 wbase = (u32)w0;
            for (u32 i = threadIdx.x; i < ncols; i += THREADS) vals[i] = T(0);
            for (u32 i = threadIdx.x; i < nwords; i += THREADS) bm[i] = 0;
            __syncthreads();
            for_each_product<true>(g, src, rec.a0, rec.a1, meta, scratch, [&](u32 c, T p) {
                const u32 d = c - wbase;
                if (d < ncols) {
                    atomicAdd(&vals[d], p);
                    atomicOr(&bm[d >> 5], 1u << (d & 31));
                }
            });
            const u32 total = bitmap_prefix(g, bm, pref, nwords, scratch);
            for (u32 d = threadIdx.x; d < ncols; d += THREADS) {
                const u32 word = bm[d >> 5];
                if (word & (1u << (d & 31))) {
                    const u32 r = emitted + pref[d >> 5] + __popc(word & ((1u << (d & 31)) - 1u));
                    c_col[rec.base + r] = wbase + d;
                    c_val[rec.base + r] = vals[d];
                }
            }
            emitted += total;
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------ launchers
constexpr u32 kW512W1 = 256;   // 256 Ki columns per sort window
constexpr u32 kB2KW1 = 512;    // 512 Ki columns per sort window
constexpr u32 kB8KW1 = 512;

template <typename T>
u32 numeric_lds_bytes_t(int cls)
{
    switch (cls) {
        case NUM_DIRECT: return 0;
        case NUM_G16: return 16 * num_group_lds<SubWave<16>, T, kNumG16Cap, 256>();
        case NUM_W128: return 4 * num_group_lds<SubWave<64>, T, kNumW128Cap, 256>();
        case NUM_W512: return 4 * num_group_lds<SubWave<64>, T, kNumW512Cap, 256>();
        case NUM_B2K: return num_group_lds<Block<256>, T, kNumB2KCap, 256>();
        case NUM_B8K: return num_group_lds<Block<512>, T, kNumB8KCap, 512>();
        case NUM_D1: return num_dense_lds<T, kNumD1Cols, 256>();
        case NUM_D2: return num_dense_lds<T, kNumD2Cols, 1024>();
    }
    return 0;
}

u32 numeric_lds_bytes(int cls, u32 vsize)
{
    return vsize == 8 ? numeric_lds_bytes_t<double>(cls) : numeric_lds_bytes_t<float>(cls);
}

template <typename K>
static void set_dyn_lds(K kernel, u32 bytes)
{
    if (bytes > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <class G, typename T, u32 CAP, u32 W1, u32 NMAX, int MODE, int THREADS>
static void launch_num_hash(hipStream_t s, int cls, u32 count, const ProductSrc<T>& A, const u32* B,
                            const RowWork& w, u32* c_col, T* c_val, int cu_count)
{
    auto k = num_hash_kernel<G, T, CAP, W1, NMAX, MODE, THREADS>;
    const u32 lds = numeric_lds_bytes_t<T>(cls);
    set_dyn_lds(k, lds);
    hipLaunchKernelGGL(k, dim3(grid_for(count, lds, THREADS, cu_count, THREADS / G::SIZE)),
                       dim3(THREADS), lds, s, A, B, w, c_col, c_val, cls);
}

template <typename T>
void launch_numeric(hipStream_t s, int cls, u32 count, const CsrView<T>& Av, const CsrView<T>& Bv,
                    const RowWork& w, u32* c_col, T* c_val, int cu_count)
{
    if (count == 0) return;
    // (A, B) below = (product source, A.row_offsets): the kernels rebase the per-entry arrays
    const ProductSrc<T> A{w.b_start, w.b_len, Av.data, Bv.col_ids, Bv.data};
    const u32* B = Av.row_offsets;
    const u32 lds = numeric_lds_bytes_t<T>(cls);
    switch (cls) {
        case NUM_DIRECT: {
            constexpr int TH = 256;
            hipLaunchKernelGGL((num_direct_kernel<T, TH>), dim3(grid_for(count, 0, TH, cu_count, TH / 16)),
                               dim3(TH), 0, s, A, B, w, c_col, c_val);
            break;
        }
        case NUM_G16:
            launch_num_hash<SubWave<16>, T, kNumG16Cap, 0, kNumG16MaxNnz, SORT_RANK, 256>(
                s, cls, count, A, B, w, c_col, c_val, cu_count);
            break;
        case NUM_W128:
            launch_num_hash<SubWave<64>, T, kNumW128Cap, 0, kNumW128MaxNnz, SORT_RANK, 256>(
                s, cls, count, A, B, w, c_col, c_val, cu_count);
            break;
        case NUM_W512:
            launch_num_hash<SubWave<64>, T, kNumW512Cap, kW512W1, kNumW512MaxNnz, SORT_BITMAP, 256>(
                s, cls, count, A, B, w, c_col, c_val, cu_count);
            break;
        case NUM_B2K:
            launch_num_hash<Block<256>, T, kNumB2KCap, kB2KW1, kNumB2KMaxNnz, SORT_BITMAP, 256>(
                s, cls, count, A, B, w, c_col, c_val, cu_count);
            break;
        case NUM_B8K:
            launch_num_hash<Block<512>, T, kNumB8KCap, kB8KW1, kNumB8KMaxNnz, SORT_BITMAP, 512>(
                s, cls, count, A, B, w, c_col, c_val, cu_count);
            break;
        case NUM_D1: {
            auto k = num_dense_kernel<T, kNumD1Cols, 256>;
            set_dyn_lds(k, lds);
            hipLaunchKernelGGL(k, dim3(grid_for(count, lds, 256, cu_count, 1)), dim3(256), lds, s, A, B, w,
                               c_col, c_val, cls);
            break;
        }
        case NUM_D2: {
            auto k = num_dense_kernel<T, kNumD2Cols, 1024>;
            set_dyn_lds(k, lds);
            hipLaunchKernelGGL(k, dim3(grid_for(count, lds, 1024, cu_count, 1)), dim3(1024), lds, s, A, B,
                               w, c_col, c_val, cls);
            break;
        }
    }
}

template void launch_numeric<double>(hipStream_t, int, u32, const CsrView<double>&,
                                     const CsrView<double>&, const RowWork&, u32*, double*, int);
template void launch_numeric<float>(hipStream_t, int, u32, const CsrView<float>&,
                                    const CsrView<float>&, const RowWork&, u32*, float*, int);

}  // namespace speck
