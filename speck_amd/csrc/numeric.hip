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
#include <type_traits>

#include "device_common.hpp"
#include "launch.hpp"
#include "row_groups.hpp"

namespace speck {

// ------------------------------------------------------------------ NUM_DIRECT
// A row with one entry: C row = a * B row (already sorted).  These rows are short (a handful of
// products) and there are many of them, so a workgroup takes a CHUNK of THREADS such rows and walks
// the flattened product space of the chunk exactly like the products of one long row: product p
// belongs to the row s with incl[s] > p (window_owners), lanes of a wave read consecutive B entries
// and write consecutive C entries, and no lane idles on a short row.
template <typename T, int THREADS>
constexpr u32 num_direct_lds()
{
    // per row: incl | B source index rebased | C destination rebased | a   + scan scratch + windows
    return THREADS * (12u + (u32)sizeof(T)) + (THREADS / 64 + 2 + win_words<Block<THREADS>>() + 3) / 4 * 16;
}

template <typename T, int THREADS>
__device__ __forceinline__ void num_direct_body(unsigned char* smem, const ProductSrc<T>& src, const RowWork& w,
                                                u32* __restrict__ c_col, T* __restrict__ c_val, u32 bidx,
                                                u32 nblk)
{
    using G = Block<THREADS>;
    const G g;
    T* m_av = reinterpret_cast<T*>(smem);
    u32* m_incl = reinterpret_cast<u32*>(m_av + THREADS);
    u32* m_src = m_incl + THREADS;
    u32* m_dst = m_src + THREADS;
    u32* scratch = m_dst + THREADS;
    u32* win_all = scratch + THREADS / 64 + 2;
    const u32 count = w.st->num.count[NUM_DIRECT];
    const RowRec* recs = w.recs + w.st->num.offset[NUM_DIRECT];
    const u32 l = lane_id();
    u32* win = win_all + (threadIdx.x >> 6) * kWinWords;
    for (u32 first = bidx * THREADS; first < count; first += nblk * THREADS) {
        const u32 cnt = min((u32)THREADS, count - first);
        u32 len = 0, bs = 0, base = 0;
        T av = T(0);
        if (threadIdx.x < cnt) {
            const RowRec rec = recs[first + threadIdx.x];
            len = rec.nnz;
            base = rec.base;
            av = src.a_val[rec.a0];
            bs = src.b_start[rec.a0];
        }
        u32 total;
        const u32 incl = g.inclusive_scan(len, &total, scratch);
        if (threadIdx.x < cnt) {
            m_incl[threadIdx.x] = incl;
            m_src[threadIdx.x] = bs - (incl - len);
            m_dst[threadIdx.x] = base - (incl - len);
            m_av[threadIdx.x] = av;
        }
        g.sync();
        u32 p, step, end;
        g.product_range(total, p, step, end);
        u32 pbase = (u32)__builtin_amdgcn_readfirstlane((int)(p - l));
        const u32 wend = (u32)__builtin_amdgcn_readfirstlane((int)end);
        u32 s0 = 0;
        if (pbase < wend) s0 = uniform_owner(m_incl, cnt, pbase);
        while (pbase < wend) {
            u32 own[kBatch];
            window_owners(m_incl, win, cnt, pbase, s0, own);
            u32 col[kBatch], dst[kBatch];
            T bv[kBatch], a[kBatch];
            bool ok[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const u32 pu = pbase + u * 64 + l;
                ok[u] = pu < wend;
                if (ok[u]) {
                    const u32 ib = m_src[own[u]] + pu;
                    dst[u] = m_dst[own[u]] + pu;
                    a[u] = m_av[own[u]];
                    col[u] = src.b_col[ib];
                    bv[u] = src.b_val[ib];
                }
            }
#pragma unroll
            for (int u = 0; u < kBatch; ++u)
                if (ok[u]) {
                    c_col[dst[u]] = col[u];
                    c_val[dst[u]] = a[u] * bv[u];
                }
            pbase += kWinProducts;
        }
        g.sync();
    }
}

// LDS accumulator cell.  fp32 rows accumulate in fp64 cells too: ds_add_f32 is ~10x slower than
// ds_add_f64 on gfx950 (192 vs 20.6 cycles per wave instruction, scripts/ubench/lds_atomics.hip).
// The product a*b is still rounded to T first (as the reference does); the sum is rounded to T
// once, when the row is written.
template <typename T>
using Acc = double;

// ------------------------------------------------------------------ sorting back-ends
// Rank sort for tiny tables: every lane owns OWN = CAP/SIZE slots (registers).
// `ckeys` may alias the table: all slots are in registers before the first write.
// `cap_row` (a power of two, SIZE <= cap_row <= CAP) slots of the table are in use.
template <class G, typename T, u32 CAP>
__device__ __forceinline__ void emit_rank_sorted(const G& g, const u32* keys, const Acc<T>* vals,
                                                 u32* ckeys, u32 cap_row, u32 base,
                                                 u32* __restrict__ c_col, T* __restrict__ c_val)
{
    constexpr u32 OWN = CAP / G::SIZE;
    u32 k[OWN];
    Acc<T> v[OWN];
#pragma unroll
    for (u32 j = 0; j < OWN; ++j) {
        k[j] = kEmptyKey;
        v[j] = 0;
        if (j * G::SIZE < cap_row) {
            k[j] = keys[j * G::SIZE + g.lane];
            v[j] = vals[j * G::SIZE + g.lane];
        }
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
    // most rows of these classes are far smaller than the class limit: when every group of the
    // wave uses a single slot per lane, only that slot is ranked
    if (__ballot(cap_row > (u32)G::SIZE) == 0) {
        for (u32 q = 0; q < (run + 3) / 4; ++q) {
            const uint4 x = ck4[q];  // same address for the whole group: LDS broadcast
            r[0] += (x.x < k[0]) + (x.y < k[0]) + (x.z < k[0]) + (x.w < k[0]);
        }
    } else {
        for (u32 q = 0; q < (run + 3) / 4; ++q) {
            const uint4 x = ck4[q];
#pragma unroll
            for (u32 j = 0; j < OWN; ++j) r[j] += (x.x < k[j]) + (x.y < k[j]) + (x.z < k[j]) + (x.w < k[j]);
        }
    }
#pragma unroll
    for (u32 j = 0; j < OWN; ++j)
        if (k[j] != kEmptyKey) {
            c_col[base + r[j]] = k[j];
            c_val[base + r[j]] = (T)v[j];
        }
}

// Two-level bitmap sort (see the header comment).  S: LDS scratch of max(2*W1, 2*NMAX) words;
// it may alias the table (slots are loaded into registers first).
template <class G, typename T, u32 CAP, u32 W1, u32 NMAX>
__device__ __forceinline__ void emit_bitmap_sorted(const G& g, const u32* keys, const Acc<T>* vals, u32* S,
                                                   u32* scan_scratch, u32 cap_row, u32 cmin, u32 cmax,
                                                   u32 base, u32* __restrict__ c_col,
                                                   T* __restrict__ c_val, int cls = 0)
{
    constexpr u32 OWN = CAP / G::SIZE;
    constexpr u64 kWindowCols = u64(W1) * 1024;
    PHASE_BEGIN(cls);
    u32 k[OWN], brank[OWN];
    Acc<T> v[OWN];
    // slots j*SIZE + lane: only the first cap_row / SIZE of them exist for this row (the guards
    // below are uniform for the group, whole iterations are skipped)
#pragma unroll
    for (u32 j = 0; j < OWN; ++j) {
        k[j] = kEmptyKey;
        v[j] = 0;
        brank[j] = 0;
        if (j * G::SIZE < cap_row) {
            k[j] = keys[j * G::SIZE + g.lane];
            v[j] = vals[j * G::SIZE + g.lane];
        }
    }
    g.sync();
    PHASE_MARK(3);
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
            if (j * G::SIZE >= cap_row) continue;
            const u32 d = k[j] - wbase;
            if (k[j] != kEmptyKey && d < ncols) atomicOr(&l1[d >> 10], 1u << ((d >> 5) & 31));
        }
        g.sync();
        PHASE_MARK(4);
        const u32 nocc = bitmap_prefix(g, l1, l1pref, nw1, scan_scratch);
        PHASE_MARK(5);
#pragma unroll
        for (u32 j = 0; j < OWN; ++j) {
            if (j * G::SIZE >= cap_row) continue;
            const u32 d = k[j] - wbase;
            if (k[j] != kEmptyKey && d < ncols)
                brank[j] = l1pref[d >> 10] + __popc(l1[d >> 10] & ((1u << ((d >> 5) & 31)) - 1u));
        }
        g.sync();  // level-1 arrays are dead from here: the masks alias them
        PHASE_MARK(6);
        for (u32 i = g.lane; i < nocc; i += G::SIZE) masks[i] = 0;
        g.sync();
#pragma unroll
        for (u32 j = 0; j < OWN; ++j) {
            if (j * G::SIZE >= cap_row) continue;
            const u32 d = k[j] - wbase;
            if (k[j] != kEmptyKey && d < ncols) atomicOr(&masks[brank[j]], 1u << (d & 31));
        }
        g.sync();
        PHASE_MARK(7);
        const u32 total = bitmap_prefix(g, masks, mpref, nocc, scan_scratch);
        PHASE_MARK(8);
#pragma unroll
        for (u32 j = 0; j < OWN; ++j) {
            if (j * G::SIZE >= cap_row) continue;
            const u32 d = k[j] - wbase;
            if (k[j] != kEmptyKey && d < ncols) {
                const u32 r = emitted + mpref[brank[j]] +
                              __popc(masks[brank[j]] & ((1u << (d & 31)) - 1u));
                c_col[base + r] = k[j];
                c_val[base + r] = (T)v[j];
            }
        }
        emitted += total;
        g.sync();
        PHASE_MARK(9);
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
    const u32 words = 2 * G::SIZE + scan_scratch_words<G, THREADS>() + win_words<G>();
    return CAP * ((u32)sizeof(Acc<T>) + 4u) + G::SIZE * (u32)sizeof(Acc<T>) + (words + 3u) / 4u * 16u;
}

template <class G, typename T, u32 CAP, u32 W1, u32 NMAX, int MODE, int THREADS>
__device__ __forceinline__ void num_hash_body(unsigned char* smem, const ProductSrc<T>& src, const RowWork& w,
                                              u32* __restrict__ c_col, T* __restrict__ c_val, int cls,
                                              u32 bidx, u32 nblk)
{
    constexpr u32 NG = THREADS / G::SIZE;
    constexpr u32 kGroupBytes = num_group_lds<G, T, CAP, THREADS>();
    // the sort scratch (rank: NMAX+8 words, bitmap: max(2*W1, 2*NMAX) words) fits in the table
    static_assert((MODE == SORT_RANK ? NMAX + 8 : (2 * W1 > 2 * NMAX ? 2 * W1 : 2 * NMAX)) * 4 <=
                      CAP * (sizeof(Acc<T>) + 4),
                  "sort scratch must fit in the table it aliases");
    const G g;
    const u32 gid = G::kIsBlock ? 0u : threadIdx.x / G::SIZE;
    unsigned char* mine = smem + gid * kGroupBytes;
    Acc<T>* vals = reinterpret_cast<Acc<T>*>(mine);
    u32* keys = reinterpret_cast<u32*>(vals + CAP);
    T* m_av = reinterpret_cast<T*>(keys + CAP);
    u32* m_incl = reinterpret_cast<u32*>(m_av + G::SIZE);
    u32* scan_scratch = m_incl + 2 * G::SIZE;
    RowMeta<T> meta{m_incl, m_incl + G::SIZE, m_av, scan_scratch + scan_scratch_words<G, THREADS>()};
    u32* S = reinterpret_cast<u32*>(mine);
    const u32 count = w.st->num.count[cls];
    const RowRec* recs = w.recs + w.st->num.offset[cls];
    u32 idx = bidx * NG + gid;
    const u32 stride = nblk * NG;
    RowRec next{};
    if (idx < count) next = recs[idx];
    while (idx < count) {
        PHASE_BEGIN(cls);
        const RowRec rec = next;  // fetched while the previous row was being processed
        if (idx + stride < count) next = recs[idx + stride];
        // table of this row: the smallest power of two >= 1.5 nnz (load <= 2/3), at least one slot
        // per lane; the class limit guarantees it fits (nnz <= 2/3 CAP)
        u32 bits = 32u - (u32)__clz((int)max(rec.nnz + (rec.nnz >> 1), 2u) - 1);
        bits = min(max(bits, (u32)__builtin_ctz(G::SIZE)), (u32)__builtin_ctz(CAP));
        if constexpr (G::SIZE >= 64) bits = (u32)__builtin_amdgcn_readfirstlane((int)bits);
        const u32 cap_row = 1u << bits;
        for (u32 i = g.lane; i < cap_row; i += G::SIZE) {
            keys[i] = kEmptyKey;
            vals[i] = 0;
        }
        g.sync();
        PHASE_MARK(0);
        for_each_product<true>(g, src, rec.a0, rec.a1, meta, scan_scratch,
                               [&](const u32(&c)[kBatch], const T(&p)[kBatch], u32 n) {
                                   Acc<T> pa[kBatch];
#pragma unroll
                                   for (int u = 0; u < kBatch; ++u) pa[u] = p[u];
                                   table_accumulate_batch(keys, vals, bits, c, pa, n);
                               }, cls);
        PHASE_MARK(1);
        if constexpr (MODE == SORT_RANK) {
            emit_rank_sorted<G, T, CAP>(g, keys, vals, S, cap_row, rec.base, c_col, c_val);
        } else {
            emit_bitmap_sorted<G, T, CAP, W1, NMAX>(g, keys, vals, S, scan_scratch, cap_row, rec.cmin,
                                                    rec.cmax, rec.base, c_col, c_val, cls);
        }
        g.sync();
        PHASE_MARK(2);
        idx += stride;
    }
}

// ------------------------------------------------------------------ NUM_D1/D2
template <typename T, u32 WCOLS, int THREADS>
constexpr u32 num_dense_lds()
{
    return (WCOLS + THREADS) * (u32)sizeof(Acc<T>) +
           (2 * (WCOLS / 32) + 2 * THREADS + THREADS / 64 + 2 + win_words<Block<THREADS>>() + 3) / 4 * 16;
}

template <typename T, u32 WCOLS, int THREADS>
__device__ __forceinline__ void num_dense_body(unsigned char* smem, const ProductSrc<T>& src, const RowWork& w,
                                               u32* __restrict__ c_col, T* __restrict__ c_val, int cls,
                                               u32 bidx, u32 nblk)
{
    constexpr u32 WORDS = WCOLS / 32;
    using G = Block<THREADS>;
    const G g;
    Acc<T>* vals = reinterpret_cast<Acc<T>*>(smem);
    T* m_av = reinterpret_cast<T*>(vals + WCOLS);
    u32* bm = reinterpret_cast<u32*>(m_av + THREADS);
    u32* pref = bm + WORDS;
    u32* scratch = pref + WORDS + 2 * THREADS;
    RowMeta<T> meta{pref + WORDS, pref + WORDS + THREADS, m_av, scratch + THREADS / 64 + 2};
    const u32 count = w.st->num.count[cls];
    const RowRec* recs = w.recs + w.st->num.offset[cls];
    RowRec next{};
    if (bidx < count) next = recs[bidx];
    for (u32 idx = bidx; idx < count; idx += nblk) {
        const RowRec rec = next;  // fetched while the previous row was being processed
        if (idx + nblk < count) next = recs[idx + nblk];
        u32 emitted = 0;
        for (u64 w0 = rec.cmin; w0 <= rec.cmax; w0 += WCOLS) {
            const u64 left = u64(rec.cmax) - w0 + 1;
            const u32 ncols = left < WCOLS ? (u32)left : WCOLS;
            const u32 nwords = (ncols + 31) >> 5;
            const u32 wbase = (u32)w0;
            for (u32 i = threadIdx.x; i < ncols; i += THREADS) vals[i] = 0;
            for (u32 i = threadIdx.x; i < nwords; i += THREADS) bm[i] = 0;
            __syncthreads();
            for_each_product<true>(g, src, rec.a0, rec.a1, meta, scratch,
                                   [&](const u32(&c)[kBatch], const T(&p)[kBatch], u32 n) {
#pragma unroll
                                       for (int u = 0; u < kBatch; ++u) {
                                           const u32 d = c[u] - wbase;
                                           if ((u32)u < n && d < ncols) {
                                               atomicAdd(&vals[d], (Acc<T>)p[u]);
                                               atomicOr(&bm[d >> 5], 1u << (d & 31));
                                           }
                                       }
                                   });
            const u32 total = bitmap_prefix(g, bm, pref, nwords, scratch);
            for (u32 d = threadIdx.x; d < ncols; d += THREADS) {
                const u32 word = bm[d >> 5];
                if (word & (1u << (d & 31))) {
                    const u32 r = emitted + pref[d >> 5] + __popc(word & ((1u << (d & 31)) - 1u));
                    c_col[rec.base + r] = wbase + d;
                    c_val[rec.base + r] = (T)vals[d];
                }
            }
            emitted += total;
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------ kernels
// Stand-alone kernels (one class per launch) and the merged "light" kernel: all classes whose
// workgroups are 256 threads wide and need <= ~40 KiB of LDS share ONE launch -- block ranges map
// to classes (ClassGrid), heaviest class first.  One launch instead of up to six removes the
// cross-queue fork/join hand-offs (~15 us each on MI355X) and lets the dispatcher interleave
// workgroups of different classes on a CU.
template <typename T, int THREADS>
__global__ __launch_bounds__(THREADS) void num_direct_kernel(ProductSrc<T> src, const u32* a_ro, RowWork w,
                                                             u32* __restrict__ c_col,
                                                             T* __restrict__ c_val)
{
    if (w.st->capacity_miss) return;
    src.rebase(a_ro);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    num_direct_body<T, THREADS>(smem, src, w, c_col, c_val, blockIdx.x, gridDim.x);
}

template <class G, typename T, u32 CAP, u32 W1, u32 NMAX, int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void num_hash_kernel(ProductSrc<T> src, const u32* a_ro, RowWork w,
                                                           u32* __restrict__ c_col,
                                                           T* __restrict__ c_val, int cls)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (w.st->capacity_miss) return;
    src.rebase(a_ro);
    num_hash_body<G, T, CAP, W1, NMAX, MODE, THREADS>(smem, src, w, c_col, c_val, cls, blockIdx.x, gridDim.x);
}

template <typename T, u32 WCOLS, int THREADS>
__global__ __launch_bounds__(THREADS) void num_dense_kernel(ProductSrc<T> src, const u32* a_ro, RowWork w,
                                                            u32* __restrict__ c_col,
                                                            T* __restrict__ c_val, int cls)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (w.st->capacity_miss) return;
    src.rebase(a_ro);
    num_dense_body<T, WCOLS, THREADS>(smem, src, w, c_col, c_val, cls, blockIdx.x, gridDim.x);
}

constexpr u32 kW512W1 = 256;   // 256 Ki columns per sort window
constexpr u32 kB2KW1 = 512;    // 512 Ki columns per sort window
constexpr u32 kB8KW1 = 512;

template <typename T>
__global__ __launch_bounds__(256) void num_light_kernel(ProductSrc<T> src, const u32* a_ro, RowWork w,
                                                        u32* __restrict__ c_col, T* __restrict__ c_val,
                                                        ClassGrid cg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (w.st->capacity_miss) return;
    src.rebase(a_ro);
    const u32 b = blockIdx.x;
    // launch order (ClassGrid slots): D1, B2K, W512, W128, G16, DIRECT
    if (b < cg.first[1])
        num_dense_body<T, kNumD1Cols, 256>(smem, src, w, c_col, c_val, NUM_D1, b - cg.first[0], cg.first[1] - cg.first[0]);
    else if (b < cg.first[2])
        num_hash_body<Block<256>, T, kNumB2KCap, kB2KW1, kNumB2KMaxNnz, SORT_BITMAP, 256>(
            smem, src, w, c_col, c_val, NUM_B2K, b - cg.first[1], cg.first[2] - cg.first[1]);
    else if (b < cg.first[3])
        num_hash_body<SubWave<64>, T, kNumW512Cap, kW512W1, kNumW512MaxNnz, SORT_BITMAP, 256>(
            smem, src, w, c_col, c_val, NUM_W512, b - cg.first[2], cg.first[3] - cg.first[2]);
    else if (b < cg.first[4])
        num_hash_body<SubWave<64>, T, kNumW128Cap, 0, kNumW128MaxNnz, SORT_RANK, 256>(
            smem, src, w, c_col, c_val, NUM_W128, b - cg.first[3], cg.first[4] - cg.first[3]);
    else if (b < cg.first[5])
        num_hash_body<SubWave<16>, T, kNumG16Cap, 0, kNumG16MaxNnz, SORT_RANK, 256>(
            smem, src, w, c_col, c_val, NUM_G16, b - cg.first[4], cg.first[5] - cg.first[4]);
    else
        num_direct_body<T, 256>(smem, src, w, c_col, c_val, b - cg.first[5], cg.first[6] - cg.first[5]);
}

// The three smallest classes alone: the merged kernel above takes the register count of its
// hungriest body (5 waves per SIMD); these bodies need 70 VGPRs, and a launch of their own
// reaches 7 waves per SIMD -- what rows that are one short chain of dependent loads need.
template <typename T>
__global__ __launch_bounds__(256) void num_tiny_kernel(ProductSrc<T> src, const u32* a_ro, RowWork w,
                                                       u32* __restrict__ c_col, T* __restrict__ c_val,
                                                       ClassGrid cg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (w.st->capacity_miss) return;
    src.rebase(a_ro);
    const u32 b = blockIdx.x;
    if (b < cg.first[4])
        num_hash_body<SubWave<64>, T, kNumW128Cap, 0, kNumW128MaxNnz, SORT_RANK, 256>(
            smem, src, w, c_col, c_val, NUM_W128, b - cg.first[3], cg.first[4] - cg.first[3]);
    else if (b < cg.first[5])
        num_hash_body<SubWave<16>, T, kNumG16Cap, 0, kNumG16MaxNnz, SORT_RANK, 256>(
            smem, src, w, c_col, c_val, NUM_G16, b - cg.first[4], cg.first[5] - cg.first[4]);
    else
        num_direct_body<T, 256>(smem, src, w, c_col, c_val, b - cg.first[5], cg.first[6] - cg.first[5]);
}

// ------------------------------------------------------------------ NUM_G
// Global-memory spill for heavy rows whose column range would need many dense windows
// (role of the reference's global hash maps, include/HashMap.cuh:112-134 and
// spECK_HashSpGEMM.cuh:25-36; the reference hard-disables the numeric one, Multiply.cu:699-700,
// and falls back to multi-window dense rows instead).
// The table of row i lives in a pool sized 2*nnz(C): slots [2*base_i, 2*base_i + 2*nnz_i), load
// factor 1/2 from the EXACT nnz, so it can never fill.  One workgroup owns a row from
// initialisation to emission, so no inter-workgroup protocol is needed: all table traffic during
// accumulation is L2 atomics (global_atomic_cmpswap + global_atomic_add_f64), and the emission
// reads it back with agent-scope loads that bypass this CU's L1.
// Sorted output: the distinct keys are ranked with an LDS column bitmap (512 Ki columns per
// window) -- prefix popcount, no comparisons.
template <typename V>
__device__ __forceinline__ V load_l2(const V* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T, int THREADS, u32 BMW>
constexpr u32 num_global_lds()
{
    return THREADS * (u32)sizeof(T) +
           (2 * BMW + 2 * THREADS + THREADS / 64 + 2 + win_words<Block<THREADS>>() + 3) / 4 * 16;
}

template <typename T, int THREADS, u32 BMW>
__global__ __launch_bounds__(THREADS) void num_global_kernel(ProductSrc<T> src, const u32* a_ro, RowWork w,
                                                             u32* __restrict__ c_col,
                                                             T* __restrict__ c_val, int cls)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using G = Block<THREADS>;
    using Bits = typename std::conditional<sizeof(T) == 8, unsigned long long, unsigned int>::type;
    const G g;
    T* m_av = reinterpret_cast<T*>(smem);
    u32* bm = reinterpret_cast<u32*>(m_av + THREADS);
    u32* pref = bm + BMW;
    u32* scratch = pref + BMW + 2 * THREADS;
    RowMeta<T> meta{pref + BMW, pref + BMW + THREADS, m_av, scratch + THREADS / 64 + 2};
    if (w.st->capacity_miss) return;
    src.rebase(a_ro);
    u32* gkeys = w.gkeys;
    T* gvals = static_cast<T*>(w.gvals);
    constexpr u64 kWindowCols = u64(BMW) * 32;
    const u32 count = w.st->num.count[cls];
    const RowRec* recs = w.recs + w.st->num.offset[cls];
    RowRec next{};
    if (blockIdx.x < count) next = recs[blockIdx.x];
    for (u32 idx = blockIdx.x; idx < count; idx += gridDim.x) {
        PHASE_BEGIN(cls);
        const RowRec rec = next;
        if (idx + gridDim.x < count) next = recs[idx + gridDim.x];
        const u32 cap = 2u * rec.nnz;
        const size_t t0 = 2 * size_t(rec.base);
        for (u32 i = threadIdx.x; i < cap; i += THREADS) {
            gkeys[t0 + i] = kEmptyKey;
            gvals[t0 + i] = T(0);
        }
        __syncthreads();  // the table is initialised (stores are acknowledged by L2) before any atomic
        PHASE_MARK(0);
        for_each_product<true>(g, src, rec.a0, rec.a1, meta, scratch,
                               [&](const u32(&c)[kBatch], const T(&p)[kBatch], u32 n) {
                                   u32 slot[kBatch], old[kBatch];
#pragma unroll
                                   for (int u = 0; u < kBatch; ++u) {  // kBatch L2 atomics in flight
                                       slot[u] = __umulhi(c[u] * 0x9E3779B1u, cap);
                                       old[u] = kEmptyKey;
                                       if ((u32)u < n) old[u] = atomicCAS(&gkeys[t0 + slot[u]], kEmptyKey, c[u]);
                                   }
#pragma unroll
                                   for (int u = 0; u < kBatch; ++u) {
                                       if ((u32)u >= n) continue;
                                       while (old[u] != kEmptyKey && old[u] != c[u]) {
                                           slot[u] = slot[u] + 1 == cap ? 0u : slot[u] + 1;
                                           old[u] = atomicCAS(&gkeys[t0 + slot[u]], kEmptyKey, c[u]);
                                       }
                                       unsafeAtomicAdd(&gvals[t0 + slot[u]], p[u]);
                                   }
                               }, cls);
        __syncthreads();
        PHASE_MARK(1);
        u32 emitted = 0;
        for (u64 w0 = rec.cmin; w0 <= rec.cmax; w0 += kWindowCols) {
            const u64 left = u64(rec.cmax) - w0 + 1;
            const u32 ncols = left < kWindowCols ? (u32)left : (u32)kWindowCols;
            const u32 nwords = (ncols + 31) >> 5;
            const u32 wbase = (u32)w0;
            for (u32 i = threadIdx.x; i < nwords; i += THREADS) bm[i] = 0;
            __syncthreads();
            constexpr int kScan = 8;  // independent L2 loads in flight per thread
            for (u32 i0 = threadIdx.x; i0 < cap; i0 += THREADS * kScan) {
                u32 k[kScan];
#pragma unroll
                for (int u = 0; u < kScan; ++u) {
                    const u32 i = i0 + u * THREADS;
                    k[u] = i < cap ? load_l2(&gkeys[t0 + i]) : kEmptyKey;
                }
#pragma unroll
                for (int u = 0; u < kScan; ++u) {
                    const u32 d = k[u] - wbase;
                    if (k[u] != kEmptyKey && d < ncols) atomicOr(&bm[d >> 5], 1u << (d & 31));
                }
            }
            __syncthreads();
            const u32 total = bitmap_prefix(g, bm, pref, nwords, scratch);
            for (u32 i0 = threadIdx.x; i0 < cap; i0 += THREADS * kScan) {
                u32 k[kScan];
                Bits raw[kScan];
#pragma unroll
                for (int u = 0; u < kScan; ++u) {
                    const u32 i = i0 + u * THREADS;
                    k[u] = i < cap ? load_l2(&gkeys[t0 + i]) : kEmptyKey;
                    raw[u] = i < cap ? load_l2(reinterpret_cast<const Bits*>(&gvals[t0 + i])) : Bits(0);
                }
#pragma unroll
                for (int u = 0; u < kScan; ++u) {
                    const u32 d = k[u] - wbase;
                    if (k[u] != kEmptyKey && d < ncols) {
                        const u32 r = emitted + pref[d >> 5] + __popc(bm[d >> 5] & ((1u << (d & 31)) - 1u));
                        T val;
                        __builtin_memcpy(&val, &raw[u], sizeof(T));
                        c_col[rec.base + r] = k[u];
                        c_val[rec.base + r] = val;
                    }
                }
            }
            emitted += total;
            __syncthreads();
        }
        PHASE_MARK(2);
    }
}

// ------------------------------------------------------------------ launchers
constexpr u32 kNumGBmWords = 16384;  // 512 Ki columns per sort window of the global-spill class

template <typename T>
u32 numeric_lds_bytes_t(int cls)
{
    switch (cls) {
        case NUM_DIRECT: return num_direct_lds<T, 256>();
        case NUM_G16: return 16 * num_group_lds<SubWave<16>, T, kNumG16Cap, 256>();
        case NUM_W128: return 4 * num_group_lds<SubWave<64>, T, kNumW128Cap, 256>();
        case NUM_W512: return 4 * num_group_lds<SubWave<64>, T, kNumW512Cap, 256>();
        case NUM_W1K: return 4 * num_group_lds<SubWave<64>, T, kNumW1KCap, 256>();
        case NUM_B2K: return num_group_lds<Block<256>, T, kNumB2KCap, 256>();
        case NUM_B8K: return num_group_lds<Block<512>, T, kNumB8KCap, 512>();
        case NUM_D1: return num_dense_lds<T, kNumD1Cols, 256>();
        case NUM_D2: return num_dense_lds<T, kNumD2Cols, 1024>();
        case NUM_G: return num_global_lds<T, 1024, kNumGBmWords>();
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
void launch_numeric_light(hipStream_t s, const u32* counts_hint, u32 mask, const CsrView<T>& Av,
                          const CsrView<T>& Bv, const RowWork& w, u32* c_col, T* c_val, int cu_count)
{
    static const int slots[6] = {NUM_D1, NUM_B2K, NUM_W512, NUM_W128, NUM_G16, NUM_DIRECT};
    static const u32 rows_per_block[6] = {1, 1, 4, 4, 16, 256};
    u32 lds = 0;
    for (int k = 0; k < 6; ++k)
        if (mask >> slots[k] & 1u) lds = lds > numeric_lds_bytes_t<T>(slots[k]) ? lds : numeric_lds_bytes_t<T>(slots[k]);
    ClassGrid cg{};
    for (int k = 0; k < 6; ++k) {
        const bool on = (mask >> slots[k] & 1u) && counts_hint[slots[k]];
        cg.first[k + 1] = cg.first[k] + (on ? grid_for(counts_hint[slots[k]], lds, 256, cu_count, rows_per_block[k]) : 0u);
    }
    if (cg.first[6] == 0) return;
    const ProductSrc<T> src{w.b_start, w.b_len, Av.data, Bv.col_ids, Bv.data};
    if (cg.first[3] == 0)
        hipLaunchKernelGGL((num_tiny_kernel<T>), dim3(cg.first[6]), dim3(256), lds, s, src, Av.row_offsets, w,
                           c_col, c_val, cg);
    else
        hipLaunchKernelGGL((num_light_kernel<T>), dim3(cg.first[6]), dim3(256), lds, s, src, Av.row_offsets, w,
                           c_col, c_val, cg);
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
            hipLaunchKernelGGL((num_direct_kernel<T, TH>), dim3(grid_for(count, lds, TH, cu_count, TH)),
                               dim3(TH), lds, s, A, B, w, c_col, c_val);
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
        case NUM_W1K:
            launch_num_hash<SubWave<64>, T, kNumW1KCap, kW512W1, kNumW1KMaxNnz, SORT_BITMAP, 256>(
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
        case NUM_G: {
            auto k = num_global_kernel<T, 1024, kNumGBmWords>;
            set_dyn_lds(k, lds);
            hipLaunchKernelGGL(k, dim3(grid_for(count, lds, 1024, cu_count, 1)), dim3(1024), lds, s, A, B,
                               w, c_col, c_val, cls);
            break;
        }
    }
}

}  // namespace speck
#ifdef SPECK_PHASE_CLOCKS
// out: kMaxClasses*16 u64 (class-major); the device counters are reset.
extern "C" int speck_debug_phase_clocks(unsigned long long* out)
{
    constexpr int n = speck::kMaxClasses * 16;
    static unsigned long long all[speck::kPhaseSlots * n];
    if (hipMemcpyFromSymbol(all, HIP_SYMBOL(speck::g_phase_clk), sizeof(all)) != hipSuccess) return 3;
    for (int i = 0; i < n; ++i) out[i] = 0;
    for (int s = 0; s < speck::kPhaseSlots; ++s)
        for (int i = 0; i < n; ++i) out[i] += all[s * n + i];
    for (auto& x : all) x = 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(speck::g_phase_clk), all, sizeof(all)) == hipSuccess ? 0 : 3;
}
#endif
namespace speck {

template void launch_numeric_light<double>(hipStream_t, const u32*, u32, const CsrView<double>&,
                                           const CsrView<double>&, const RowWork&, u32*, double*, int);
template void launch_numeric_light<float>(hipStream_t, const u32*, u32, const CsrView<float>&,
                                          const CsrView<float>&, const RowWork&, u32*, float*, int);
template void launch_numeric<double>(hipStream_t, int, u32, const CsrView<double>&,
                                     const CsrView<double>&, const RowWork&, u32*, double*, int);
template void launch_numeric<float>(hipStream_t, int, u32, const CsrView<float>&,
                                    const CsrView<float>&, const RowWork&, u32*, float*, int);

}  // namespace speck
