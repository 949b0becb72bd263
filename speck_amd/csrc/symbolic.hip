// symbolic.hip -- symbolic phase for gfx950: nnz of every C row (distinct columns reached by
// the row; structural, values are never read).
// Role of the reference's spGEMMCountLauncher / denseSpGEMMCount
// (include/GPU/spECK_HashSpGEMM.cuh:1797-1853, 1681-1711) and HashMapNoValue
// (include/HashMap.cuh:136-229); designed for 64-lane waves and 160 KiB of LDS:
//   SYM_G8 / G16   : 8 / 16 lanes per row (8 / 4 rows per wave), 32 / 64-key set, no barrier
//   SYM_W128/W256  : 16 / 32 lanes per row, 128 / 256-key set
//   SYM_W1K        : one wave per row, 1024-key set, no workgroup barrier
//   SYM_B4K/16K/32K: one workgroup per row, 16 / 64 / 128 KiB key set
//   SYM_BM1/BM2    : column bitmap (1 bit per column): one ds_or per product, no probing;
//                    128 KiB of LDS cover 1 Mi columns per window (per-entry cursors between windows)
//   SYM_GH         : key set in global memory for rows that are wide AND sparse (few products per bitmap
//                    window) -- role of the reference's global hash maps for the symbolic phase
//                    (include/GPU/spECK_HashSpGEMM.cuh:89-126, 1025-1057; include/HashMap.cuh:112-134)
// Algorithmic bytes per row: 8 + 12*lenA + 4*ops + 4 (device_common.hpp).
#include "device_common.hpp"
#include "launch.hpp"
#include "row_groups.hpp"
#include "esc.hpp"
#include "esc_rows.hpp"
#include "esc_wide.hpp"

namespace speck {

template <class G, int THREADS>
constexpr u32 sym_scratch_words()
{
    return G::kIsBlock ? (THREADS / 64 + 2) : 0;
}

// LDS per group: key set + A-row staging (+ scan scratch for workgroup groups)
template <class G, u32 CAP, int THREADS>
constexpr u32 sym_group_lds()
{
    return (CAP + 2 * G::SIZE + sym_scratch_words<G, THREADS>() + win_words<G>() + 3u) / 4u * 16u;
}

template <class G, u32 CAP, int THREADS>
__device__ __forceinline__ void sym_hash_body(unsigned char* smem, const ProductSrc<float>& src,
                                              const RowWork& w, u32* __restrict__ counts, int cls, u32 bidx,
                                              u32 nblk, u32 hint = kNoCount)
{
    constexpr u32 NG = THREADS / G::SIZE;
    constexpr u32 kGroupBytes = sym_group_lds<G, CAP, THREADS>();
    const G g;
    const u32 gid = G::kIsBlock ? 0u : threadIdx.x / G::SIZE;
    u32* tab = reinterpret_cast<u32*>(smem + gid * kGroupBytes);
    u32* scratch = tab + CAP + 2 * G::SIZE;
    RowMeta<float> meta{tab + CAP, tab + CAP + G::SIZE, nullptr, scratch + sym_scratch_words<G, THREADS>()};
    RowCursor cur = open_list<true>(w, cls, hint, bidx, nblk, NG, gid, (w.xcd_aware & (G::kIsBlock ? 4u : 1u)) != 0);
    // (a replayed sequence that an earlier kernel has declared void walks nothing)
    if (group_void(g, cur.miss)) return;
    while (cur.more()) {
        const RowRec rec = cur.take();  // (its successor's record is requested now: RowCursor)
        static_assert(CAP % (4 * G::SIZE) == 0, "the key set is cleared 16 bytes per lane and step");
        for (u32 q = g.lane; q < CAP / 4; q += G::SIZE)
            reinterpret_cast<uint4*>(tab)[q] = make_uint4(kEmptyKey, kEmptyKey, kEmptyKey, kEmptyKey);
        g.sync();
        u32 cnt = 0;
        for_each_product<false>(g, src, rec.a0, rec.a1, meta, scratch,
                                [&](const u32(&c)[kBatch], const float(&)[kBatch], u32 n) {
                                    cnt += set_insert_batch<CAP>(tab, c, n);
                                });
        cnt = g.reduce_add(cnt, scratch);
        if (g.lane == 0) store_row_count(w, counts, rec.row, cnt);
        g.sync();
    }
}

// SYM_G8 / SYM_G16: rows with at most 4 L products from at most L entries of A (esc.hpp; L = 8 / 16 lanes per row)
// -- the products' columns are sorted in the registers of the row's lanes, the row's nnz is the number of places
// where the column changes.  No key set, no compare-and-swap, no probing loop.  LDS per group: the B-row offsets
// of the (non-empty) entries.
template <u32 L>
constexpr u32 sym_esc_group_lds() { return L * 4u; }

template <u32 L, int THREADS>
__device__ __forceinline__ void sym_esc_body(unsigned char* smem, const ProductSrc<float>& src, const RowWork& w,
                                             u32* __restrict__ counts, int cls, u32 bidx, u32 nblk,
                                             u32 hint = kNoCount)
{
    using G = SubWave<L>;
    constexpr u32 NG = THREADS / L, PER = kEscPerLane;
    const G g;
    const u32 gid = threadIdx.x / L;
    u32* s_off = reinterpret_cast<u32*>(smem + gid * sym_esc_group_lds<L>());
    RowCursor cur = open_list<true>(w, cls, hint, bidx, nblk, NG, gid, (w.xcd_aware & 8u) != 0);
    // (a replayed sequence that an earlier kernel has declared void walks nothing)
    if (wave_void(cur.miss)) return;
    const u32 gl = g.lane;
    while (cur.more()) {
        const RowRec rec = cur.take();  // (its successor's record is requested now: RowCursor)
        uint2 sl = make_uint2(0u, 0u);
        if (rec.a0 + gl < rec.a1) sl = src.b_sl[rec.a0 + gl];
        u32 total;
        const u32 incl = g.inclusive_scan(sl.y, &total, nullptr);
        const bool nonempty = sl.y != 0;
        const u32 before = (u32)__popcll(g.ballot(nonempty) & ((1ull << gl) - 1ull));
        if (nonempty) s_off[before] = sl.x - (incl - sl.y);
        u64 ends;
        if constexpr (L == 8) {
            ends = esc_group_or((nonempty && incl < 32u) ? (1u << incl) : 0u);
        } else {
            const u64 bit = (nonempty && incl < 64u) ? (1ull << incl) : 0ull;
            ends = (u64(esc_row_or((u32)(bit >> 32))) << 32) | esc_row_or((u32)bit);
        }
        wave_lds_fence();
        u32 col[PER];
#pragma unroll
        for (u32 u = 0; u < PER; ++u) {
            const u32 p = u * L + gl;
            col[u] = kEscInvalid;
            if (p < total) col[u] = src.b_col[s_off[(u32)__popcll(ends & ((2ull << p) - 1ull))] + p];
        }
        esc_sort<L>(col, gl);
        const u32 prev_col = dpp_move<kDppRowShr + 1>(kEscInvalid, col[PER - 1]);
        u32 heads = 0;
#pragma unroll
        for (u32 r = 0; r < PER; ++r) {
            const u32 before_col = r ? col[r - 1] : (gl ? prev_col : kEscInvalid);
            heads += (col[r] != kEscInvalid && col[r] != before_col) ? 1u : 0u;
        }
        heads = g.reduce_add(heads, nullptr);
        if (gl == 0) store_row_count(w, counts, rec.row, heads);
        wave_lds_fence();  // the next row overwrites the offsets
    }
}

template <u32 WORDS, int THREADS>
__device__ __forceinline__ void sym_bitmap_body(unsigned char* smem, const ProductSrc<float>& src,
                                                const RowWork& w, u32* __restrict__ counts, int cls, u32 bidx,
                                                u32 nblk, u32 hint = kNoCount)
{
    using G = Block<THREADS>;
    const G g;
    u32* bm = reinterpret_cast<u32*>(smem);
    u32* scratch = bm + WORDS + 2 * THREADS;
    RowMeta<float> meta{bm + WORDS, bm + WORDS + THREADS, nullptr, scratch + THREADS / 64 + 2};
    constexpr u64 kWindowCols = u64(WORDS) * 32;
    RowCursor cur = open_list<true>(w, cls, hint, bidx, nblk, 1u, 0u, (w.xcd_aware & 2u) != 0);
    if (block_void(cur.miss)) return;
    while (cur.more()) {
        const RowRec rec = cur.take();  // (its successor's record is requested now: RowCursor)
        u32 total = 0;
        // a row wider than one window: per-entry cursors, every B entry is read once (WindowCursors)
        const bool multi = u64(rec.cmax) - rec.cmin + 1 > kWindowCols;
        const WindowCursors<THREADS> cur{src.b_sl, src.b_col, src.w_sl, rec.a0, rec.a1};
        ProductSrc<float> wsrc = src;
        if (multi) {
            wsrc.b_sl = src.w_sl;
            cur.reset();
        }
        u32 base = rec.cmin;
        while (true) {
            if (multi) {
                base = cur.next_window((u32)kWindowCols, scratch);
                if (base == 0xFFFFFFFFu) break;
            }
            const u64 left = u64(rec.cmax) - base + 1;
            const u32 ncols = left < kWindowCols ? (u32)left : (u32)kWindowCols;
            const u32 nwords = (ncols + 31) >> 5;
            for (u32 i = threadIdx.x; i < nwords; i += THREADS) bm[i] = 0;
            __syncthreads();
            for_each_product<false>(g, wsrc, rec.a0, rec.a1, meta, scratch,
                                    [&](const u32(&c)[kBatch], const float(&)[kBatch], u32 n) {
#pragma unroll
                                        for (int u = 0; u < kBatch; ++u) {
                                            const u32 d = c[u] - base;  // wraps when left of the window
                                            bitmap_or_runs(bm, ((u32)u < n && d < ncols) ? d >> 5 : 0xFFFFFFFFu, 1u << (d & 31));
                                        }
                                    });
            for (u32 i = threadIdx.x; i < nwords; i += THREADS) total += __popc(bm[i]);
            __syncthreads();
            if (!multi) break;
        }
        total = g.reduce_add(total, scratch);
        if (threadIdx.x == 0) store_row_count(w, counts, rec.row, total);
    }
}

// SYM_GH: one workgroup per row, the key set is the row's slot of the scratch pool (sized by the analysis
// pass from the product count, a power of two at load <= 1/2).  Every compare-and-swap of a row comes from
// ONE workgroup, i.e. one XCD and one L2: workgroup-scope atomics (performed in that L2, no trip to the
// memory side) are enough.  The batched form keeps 4 x 1024 independent atomics in flight; only a collision
// enters the probing loop.
__device__ __forceinline__ u32 gh_cas(u32* p, u32 key)
{
    u32 expected = kEmptyKey;
    __hip_atomic_compare_exchange_strong(p, &expected, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_WORKGROUP);
    return expected;  // previous content
}
__device__ __forceinline__ u32 gh_insert_batch(u32* __restrict__ tab, u32 shift, u32 mask, const u32 (&key)[kBatch],
                                               u32 nvalid)
{
    u32 slot[kBatch], old[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
        slot[u] = (key[u] * 0x9E3779B1u) >> shift;
        old[u] = kEmptyKey;
        if ((u32)u < nvalid) old[u] = gh_cas(&tab[slot[u]], key[u]);
    }
    // collisions: the next probes of ALL pending keys of the lane go out together -- the number of L2 round
    // trips is the longest chain among the lane's keys, not the sum of the chains
    u32 added = 0;
    bool pending[kBatch];
    bool any = false;
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
        added += ((u32)u < nvalid && old[u] == kEmptyKey) ? 1u : 0u;
        pending[u] = (u32)u < nvalid && old[u] != kEmptyKey && old[u] != key[u];
        any |= pending[u];
    }
    while (any) {
#pragma unroll
        for (int u = 0; u < kBatch; ++u)
            if (pending[u]) {
                slot[u] = (slot[u] + 1u) & mask;
                old[u] = gh_cas(&tab[slot[u]], key[u]);
            }
        any = false;
#pragma unroll
        for (int u = 0; u < kBatch; ++u)
            if (pending[u]) {
                added += old[u] == kEmptyKey ? 1u : 0u;
                pending[u] = old[u] != kEmptyKey && old[u] != key[u];
                any |= pending[u];
            }
    }
    return added;
}

constexpr int kGhThreads = 1024;
__global__ __launch_bounds__(kGhThreads) void sym_global_hash_kernel(ProductSrc<float> src, const u32* a_ro, RowWork w,
                                                                      u32* __restrict__ counts)
{
    SPECK_POISON();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // a replayed sequence whose scratch pool no longer holds the key sets (sym_scatter_kernel raised the flag)
    // must not clear or probe them: the eager path re-runs with a pool of the right size
    if (block_void(w.st->capacity_miss)) return;
    src.rebase(a_ro);
    using G = Block<kGhThreads>;
    const G g;
    u32* lds = reinterpret_cast<u32*>(smem);
    u32* scratch = lds + 2 * kGhThreads;
    RowMeta<float> meta{lds, lds + kGhThreads, nullptr, scratch + kGhThreads / 64 + 2};
    const u32 count = min(w.st->sym.count[SYM_GH], w.m);
    for (u32 idx = blockIdx.x; idx < count; idx += gridDim.x) {
        const RowRec rec = *class_rec_at(w.recs, w.m, SYM_GH, idx);
        const u32 slots = gh_table_slots(rec.ops);
        const u32 shift = 32u - (u32)__builtin_ctz(slots);
        const u64 slot0 = w.nf_off[rec.row];
        if (slot0 + slots > w.nf_cap) {  // never past the end of the pool, whatever the flag said
            if (threadIdx.x == 0) const_cast<DeviceStats*>(w.st)->capacity_miss = 1;
            continue;
        }
        u32* tab = w.nf_col + slot0;
        for (u32 i = threadIdx.x; i < slots; i += kGhThreads) tab[i] = kEmptyKey;
        __threadfence();  // the stores are in the L2 before any wave of this workgroup sends an atomic there
        __syncthreads();
        u32 cnt = 0;
        for_each_product<false>(g, src, rec.a0, rec.a1, meta, scratch,
                                [&](const u32(&c)[kBatch], const float(&)[kBatch], u32 n) {
                                    cnt += gh_insert_batch(tab, shift, slots - 1u, c, n);
                                });
        cnt = g.reduce_add(cnt, scratch);
        if (threadIdx.x == 0) store_row_count(w, counts, rec.row, cnt);
        __syncthreads();
    }
}

// ------------------------------------------------------------------ kernels
// Stand-alone kernels (one class per launch) and the merged "light" kernel (see numeric.hip):
// every 256-thread class shares one launch, block ranges map to classes, heaviest first.
template <class G, u32 CAP, int THREADS>
__global__ __launch_bounds__(THREADS) void sym_hash_kernel(ProductSrc<float> src, const u32* a_ro,
                                                           RowWork w, u32* __restrict__ counts,
                                                           int cls)
{
    SPECK_POISON();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    src.rebase(a_ro);
    sym_hash_body<G, CAP, THREADS>(smem, src, w, counts, cls, blockIdx.x, gridDim.x);
}

template <u32 WORDS, int THREADS>
__global__ __launch_bounds__(THREADS) void sym_bitmap_kernel(ProductSrc<float> src, const u32* a_ro,
                                                             RowWork w, u32* __restrict__ counts,
                                                             int cls)
{
    SPECK_POISON();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    src.rebase(a_ro);
    sym_bitmap_body<WORDS, THREADS>(smem, src, w, counts, cls, blockIdx.x, gridDim.x);
}

template <u32 L>
__global__ __launch_bounds__(256) void sym_escw_kernel(ProductSrc<float> src, const u32* a_ro, RowWork w,
                                                       u32* __restrict__ counts, int cls)
{
    SPECK_POISON();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    src.rebase(a_ro);
    sym_escw_body<L, 256>(smem, src, w, counts, cls, blockIdx.x, gridDim.x);
}

template <u32 L>
__global__ __launch_bounds__(256) void sym_esc_kernel(ProductSrc<float> src, const u32* a_ro, RowWork w,
                                                      u32* __restrict__ counts, int cls)
{
    SPECK_POISON();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    src.rebase(a_ro);
    sym_esc_body<L, 256>(smem, src, w, counts, cls, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256) void sym_light_kernel(ProductSrc<float> src, const u32* a_ro, RowWork w,
                                                        u32* __restrict__ counts, ClassGrid cg)
{
    SPECK_POISON();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    src.rebase(a_ro);
    const u32 b = blockIdx.x;
    // launch order (ClassGrid slots): BM1, B4K, W1K, W256, R64, R32, W128, G16, G8
    if (b < cg.first[1])
        sym_bitmap_body<kSymBm1Words, 256>(smem, src, w, counts, SYM_BM1, b - cg.first[0], cg.first[1] - cg.first[0], cg.cnt[0]);
    else if (b < cg.first[2])
        sym_hash_body<Block<256>, kSymB4KCap, 256>(smem, src, w, counts, SYM_B4K, b - cg.first[1], cg.first[2] - cg.first[1], cg.cnt[1]);
    else if (b < cg.first[3])
        sym_hash_body<SubWave<64>, kSymW1KCap, 256>(smem, src, w, counts, SYM_W1K, b - cg.first[2], cg.first[3] - cg.first[2], cg.cnt[2]);
    else if (b < cg.first[4])
        sym_hash_body<SubWave<32>, kSymW256Cap, 256>(smem, src, w, counts, SYM_W256, b - cg.first[3], cg.first[4] - cg.first[3], cg.cnt[3]);
    else if (b < cg.first[5])
        sym_escw_body<64, 256>(smem, src, w, counts, SYM_R64, b - cg.first[4], cg.first[5] - cg.first[4], cg.cnt[4]);
    else if (b < cg.first[6])
        sym_escw_body<32, 256>(smem, src, w, counts, SYM_R32, b - cg.first[5], cg.first[6] - cg.first[5], cg.cnt[5]);
    else if (b < cg.first[7])
        sym_hash_body<SubWave<16>, kSymW128Cap, 256>(smem, src, w, counts, SYM_W128, b - cg.first[6], cg.first[7] - cg.first[6], cg.cnt[6]);
    else if (b < cg.first[8])
        sym_esc_body<16, 256>(smem, src, w, counts, SYM_G16, b - cg.first[7], cg.first[8] - cg.first[7], cg.cnt[7]);
    else
        sym_esc_body<8, 256>(smem, src, w, counts, SYM_G8, b - cg.first[8], cg.first[9] - cg.first[8], cg.cnt[8]);
}

// The light launch of a sequence that REUSES the placement of the previous identical call: the rows of the
// register classes are finished here -- products expanded, sorted, summed and written to the place that call gave the row
// in C (num_esc_body, FUSED) -- instead of being counted now and walked again in the numeric phase.  The other classes
// count as above.
template <typename T>
__global__ __launch_bounds__(256) void sym_light_fused_kernel(ProductSrc<T> nsrc, const u32* a_ro, RowWork w,
                                                              u32* __restrict__ counts, ClassGrid cg)
{
    SPECK_POISON();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    nsrc.rebase(a_ro);
    const ProductSrc<float> src{nsrc.b_sl, nullptr, nsrc.b_col, nullptr, nsrc.w_sl};
    const u32 b = blockIdx.x;
    if (b < cg.first[1])
        sym_bitmap_body<kSymBm1Words, 256>(smem, src, w, counts, SYM_BM1, b - cg.first[0], cg.first[1] - cg.first[0], cg.cnt[0]);
    else if (b < cg.first[2])
        sym_hash_body<Block<256>, kSymB4KCap, 256>(smem, src, w, counts, SYM_B4K, b - cg.first[1], cg.first[2] - cg.first[1], cg.cnt[1]);
    else if (b < cg.first[3])
        sym_hash_body<SubWave<64>, kSymW1KCap, 256>(smem, src, w, counts, SYM_W1K, b - cg.first[2], cg.first[3] - cg.first[2], cg.cnt[2]);
    else if (b < cg.first[4])
        sym_hash_body<SubWave<32>, kSymW256Cap, 256>(smem, src, w, counts, SYM_W256, b - cg.first[3], cg.first[4] - cg.first[3], cg.cnt[3]);
    else if (b < cg.first[5])
        num_escw_body<T, 64, 256, true>(smem, nsrc, w, w.nf_direct_col, static_cast<T*>(w.nf_direct_val), SYM_R64,
                                        b - cg.first[4], cg.first[5] - cg.first[4], cg.cnt[4], counts);
    else if (b < cg.first[6])
        num_escw_body<T, 32, 256, true>(smem, nsrc, w, w.nf_direct_col, static_cast<T*>(w.nf_direct_val), SYM_R32,
                                        b - cg.first[5], cg.first[6] - cg.first[5], cg.cnt[5], counts);
    else if (b < cg.first[7])
        sym_hash_body<SubWave<16>, kSymW128Cap, 256>(smem, src, w, counts, SYM_W128, b - cg.first[6], cg.first[7] - cg.first[6], cg.cnt[6]);
    else if (b < cg.first[8])
        num_esc_body<T, 16, 256, true>(smem, nsrc, w, w.nf_direct_col, static_cast<T*>(w.nf_direct_val), SYM_G16,
                                       b - cg.first[7], cg.first[8] - cg.first[7], cg.cnt[7], counts);
    else
        num_esc_body<T, 8, 256, true>(smem, nsrc, w, w.nf_direct_col, static_cast<T*>(w.nf_direct_val), SYM_G8,
                                      b - cg.first[8], cg.first[9] - cg.first[8], cg.cnt[8], counts);
}

u32 symbolic_lds_bytes(int cls)
{
    switch (cls) {
        case SYM_G8: return 32 * sym_esc_group_lds<8>();
        case SYM_G16: return 16 * sym_esc_group_lds<16>();
        case SYM_R32: return 8 * sym_escw_group_lds<32>();
        case SYM_R64: return 4 * sym_escw_group_lds<64>();
        case SYM_W128: return 16 * sym_group_lds<SubWave<16>, kSymW128Cap, 256>();
        case SYM_W256: return 8 * sym_group_lds<SubWave<32>, kSymW256Cap, 256>();
        case SYM_W1K: return 4 * sym_group_lds<SubWave<64>, kSymW1KCap, 256>();
        case SYM_B4K: return sym_group_lds<Block<256>, kSymB4KCap, 256>();
        case SYM_B16K: return sym_group_lds<Block<512>, kSymB16KCap, 512>();
        case SYM_B32K: return sym_group_lds<Block<1024>, kSymB32KCap, 1024>();
        case SYM_BM1: return (kSymBm1Words + 2 * 256 + 8 + win_words<Block<256>>()) * 4;
        case SYM_BM2: return (kSymBm2Words + 2 * 1024 + 24 + win_words<Block<1024>>()) * 4;
        case SYM_GH: return (2 * kGhThreads + kGhThreads / 64 + 8 + win_words<Block<kGhThreads>>()) * 4;
    }
    return 0;
}

template <typename K>
static void set_dyn_lds(K kernel, u32 bytes)
{
    // kernels above 64 KiB of LDS need the opt-in (a workgroup may own all 160 KiB on gfx950)
    if (bytes > 48 * 1024)
        note_launch_status(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes),
                           "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
}

static u32 g_grid_rounds_block = 1, g_grid_rounds_sub = 4;
void set_grid_rounds(u32 block_classes, u32 subwave_classes)
{
    if (block_classes) g_grid_rounds_block = block_classes;
    if (subwave_classes) g_grid_rounds_sub = subwave_classes;
}

u32 grid_for(u32 count, u32 lds, int threads, int cu_count, u32 rows_per_block)
{
    u32 per_cu = lds ? (160u * 1024u) / lds : 8;
    const u32 by_threads = 2048u / (u32)threads;
    if (per_cu > by_threads) per_cu = by_threads;
    if (per_cu < 1) per_cu = 1;
    // sub-wave classes: 4 rounds of resident workgroups, then a static stride;
    // workgroup-per-row classes (rows_per_block == 1): the resident set, rows come from a queue
    // Workgroups that fit several times into a CU are cheap to dispatch one row at a time (the
    // hardware dispatcher balances rows of very different cost); a workgroup that owns most of
    // the LDS would starve behind smaller ones, so those classes keep a resident set and stride.
    u32 rounds = rows_per_block > 1 ? g_grid_rounds_sub : g_grid_rounds_block;
    if (rows_per_block == 1 && g_grid_rounds_block == 1 && lds <= 48u * 1024u) rounds = 16;
    const u64 cap = u64(cu_count) * per_cu * rounds;
    u64 need = (u64(count) + rows_per_block - 1) / rows_per_block;
    if (need > cap) need = cap;
    return need ? (u32)need : 1u;
}

template <class G, u32 CAP, int THREADS>
static void launch_sym_hash(hipStream_t s, int cls, u32 count, const ProductSrc<float>& A,
                            const u32* B, const RowWork& w, u32* counts, int cu_count)
{
    auto k = sym_hash_kernel<G, CAP, THREADS>;
    const u32 lds = symbolic_lds_bytes(cls);
    set_dyn_lds(k, lds);
    SPECK_LAUNCH(k, dim3(grid_for(count, lds, THREADS, cu_count, THREADS / G::SIZE)),
                       dim3(THREADS), lds, s, A, B, w, counts, cls);
}

// `fused_vsize` (8 / 4, replayed sequence with direct placement only; 0 = off): the rows of SYM_G8 / SYM_G16 are
// finished by this launch (sym_light_fused_kernel); a_val / b_val are then the values of A and B.
void launch_symbolic_light(hipStream_t s, const u32* counts_hint, u32 mask, const u32* a_ro,
                           const uint2* b_sl, const u32* b_col, const RowWork& w,
                           u32* counts, int cu_count, bool exact, u32 fused_vsize, const void* a_val,
                           const void* b_val, hipEvent_t e0, hipEvent_t e1)
{
    constexpr int NS = 9;
    static const int slots[NS] = {SYM_BM1, SYM_B4K, SYM_W1K, SYM_W256, SYM_R64, SYM_R32, SYM_W128, SYM_G16, SYM_G8};
    static const u32 rows_per_block[NS] = {1, 1, 4, 8, 4, 8, 16, 16, 32};
    const bool fused = fused_vsize != 0 && (mask & kSymEscMask) != 0;
    auto class_lds = [&](int cls) -> u32 {
        if (fused && cls == SYM_G8) return 32 * (fused_vsize == 8 ? num_esc_group_lds<double, 8>() : num_esc_group_lds<float, 8>());
        if (fused && cls == SYM_G16) return 16 * (fused_vsize == 8 ? num_esc_group_lds<double, 16>() : num_esc_group_lds<float, 16>());
        if (fused && cls == SYM_R32) return 8 * (fused_vsize == 8 ? num_escw_group_lds<double, 32>() : num_escw_group_lds<float, 32>());
        if (fused && cls == SYM_R64) return 4 * (fused_vsize == 8 ? num_escw_group_lds<double, 64>() : num_escw_group_lds<float, 64>());
        return symbolic_lds_bytes(cls);
    };
    u32 lds = 0;
    for (int k = 0; k < NS; ++k)
        if (mask >> slots[k] & 1u) lds = lds > class_lds(slots[k]) ? lds : class_lds(slots[k]);
    ClassGrid cg{};
    for (int k = 0; k < NS; ++k) {
        const bool on = (mask >> slots[k] & 1u) && counts_hint[slots[k]];
        cg.first[k + 1] = cg.first[k] + (on ? grid_for(counts_hint[slots[k]], lds, 256, cu_count, rows_per_block[k]) : 0u);
        cg.cnt[k] = exact ? counts_hint[slots[k]] : kNoCount;
    }
    if (cg.first[NS] == 0) {
        if (e0) (void)hipEventRecord(e0, s), (void)hipEventRecord(e1, s);  // (nothing to time: an empty interval)
        return;
    }
    if (fused && fused_vsize == 8) {
        const ProductSrc<double> nsrc{b_sl, static_cast<const double*>(a_val), b_col, static_cast<const double*>(b_val), w.w_sl};
        SPECK_LAUNCH_TIMED((sym_light_fused_kernel<double>), dim3(cg.first[NS]), dim3(256), lds, s, e0, e1, nsrc, a_ro, w, counts, cg);
        return;
    }
    if (fused) {
        const ProductSrc<float> nsrc{b_sl, static_cast<const float*>(a_val), b_col, static_cast<const float*>(b_val), w.w_sl};
        SPECK_LAUNCH_TIMED((sym_light_fused_kernel<float>), dim3(cg.first[NS]), dim3(256), lds, s, e0, e1, nsrc, a_ro, w, counts, cg);
        return;
    }
    const ProductSrc<float> src{b_sl, nullptr, b_col, nullptr, w.w_sl};
    SPECK_LAUNCH_TIMED(sym_light_kernel, dim3(cg.first[NS]), dim3(256), lds, s, e0, e1, src, a_ro, w, counts, cg);
}

void launch_symbolic(hipStream_t s, int cls, u32 count, const u32* a_ro, const uint2* b_sl,
                     const u32* b_col, const RowWork& w, u32* counts, int cu_count)
{
    if (count == 0) return;
    // (A, B) below = (product source, A.row_offsets): the kernels rebase the per-entry arrays
    const ProductSrc<float> A{b_sl, nullptr, b_col, nullptr, w.w_sl};
    const u32* B = a_ro;
    const u32 lds = symbolic_lds_bytes(cls);
    switch (cls) {
        case SYM_G8:
            SPECK_LAUNCH(sym_esc_kernel<8>, dim3(grid_for(count, lds, 256, cu_count, 32)), dim3(256), lds, s, A, B, w,
                               counts, cls);
            break;
        case SYM_G16:
            SPECK_LAUNCH(sym_esc_kernel<16>, dim3(grid_for(count, lds, 256, cu_count, 16)), dim3(256), lds, s, A, B, w,
                               counts, cls);
            break;
        case SYM_R32:
            SPECK_LAUNCH(sym_escw_kernel<32>, dim3(grid_for(count, lds, 256, cu_count, 8)), dim3(256), lds, s, A, B, w,
                               counts, cls);
            break;
        case SYM_R64:
            SPECK_LAUNCH(sym_escw_kernel<64>, dim3(grid_for(count, lds, 256, cu_count, 4)), dim3(256), lds, s, A, B, w,
                               counts, cls);
            break;
        case SYM_W128: launch_sym_hash<SubWave<16>, kSymW128Cap, 256>(s, cls, count, A, B, w, counts, cu_count); break;
        case SYM_W256: launch_sym_hash<SubWave<32>, kSymW256Cap, 256>(s, cls, count, A, B, w, counts, cu_count); break;
        case SYM_W1K: launch_sym_hash<SubWave<64>, kSymW1KCap, 256>(s, cls, count, A, B, w, counts, cu_count); break;
        case SYM_B4K: launch_sym_hash<Block<256>, kSymB4KCap, 256>(s, cls, count, A, B, w, counts, cu_count); break;
        case SYM_B16K: launch_sym_hash<Block<512>, kSymB16KCap, 512>(s, cls, count, A, B, w, counts, cu_count); break;
        case SYM_B32K: launch_sym_hash<Block<1024>, kSymB32KCap, 1024>(s, cls, count, A, B, w, counts, cu_count); break;
        case SYM_BM1: {
            auto k = sym_bitmap_kernel<kSymBm1Words, 256>;
            SPECK_LAUNCH(k, dim3(grid_for(count, lds, 256, cu_count, 1)), dim3(256), lds, s, A, B,
                               w, counts, cls);
            break;
        }
        case SYM_BM2: {
            auto k = sym_bitmap_kernel<kSymBm2Words, 1024>;
            set_dyn_lds(k, lds);
            SPECK_LAUNCH(k, dim3(grid_for(count, lds, 1024, cu_count, 1)), dim3(1024), lds, s, A,
                               B, w, counts, cls);
            break;
        }
        case SYM_GH:
            SPECK_LAUNCH(sym_global_hash_kernel, dim3(grid_for(count, lds, kGhThreads, cu_count, 1)),
                               dim3(kGhThreads), lds, s, A, B, w, counts);
            break;
    }
}

}  // namespace speck
