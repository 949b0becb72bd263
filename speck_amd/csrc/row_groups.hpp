// row_groups.hpp -- the lane-group abstraction shared by the symbolic and numeric kernels.
//
// A row of C is produced by a GROUP: SubWave<16> (16 lanes of a wave), SubWave<64> (a wave) or
// Block<THREADS> (a workgroup).  All groups run the same three steps:
//   1. stage the A row's metadata in LDS: for each entry a_ik the inclusive prefix of
//      nnz(B_k), the B-row start rebased to that prefix, and a_ik itself;
//   2. walk the FLATTENED product space p in [0, ops): lane -> p; consecutive lanes read
//      consecutive B entries (coalesced) and every lane has the same amount of work whatever the
//      B-row length distribution is (the reference balances this with getThreadShiftNew,
//      include/common.cuh:509-555, which assumes near-uniform B rows).  The A entry that owns a
//      product is found per WINDOW of products from counts of where the entries' products end
//      (window_owners below) -- one LDS read and one DPP scan per group-width of products;
//   3. the kBatch gathers of a lane are all issued before the first of them is accumulated.
#pragma once
#include "device_common.hpp"
#include "launch.hpp"

namespace speck {

// Opt-in phase clocks (make PHASE_CLOCKS=1): cycles (s_memtime) a row spends in each phase of the
// hash kernels, summed per class by the first lane of every group.  Read and reset through
// speck_debug_phase_clocks(); scripts/phase_clocks.py prints the table.
#ifdef SPECK_PHASE_CLOCKS
constexpr int kPhaseSlots = 1024;  // spread over block ids: same-address atomics would serialise
static __device__ unsigned long long g_phase_clk[kPhaseSlots * kMaxClasses * 16];
#define PHASE_BEGIN(cls_) \
    long long pc_t_ = clock64(); \
    const int pc_cls_ = (cls_)
#define PHASE_MARK(i_) \
    do { \
        const long long pc_n_ = clock64(); \
        if (g.lane == 0) atomicAdd(&g_phase_clk[(blockIdx.x % kPhaseSlots) * kMaxClasses * 16 + pc_cls_ * 16 + (i_)], (unsigned long long)(pc_n_ - pc_t_)); \
        pc_t_ = pc_n_; \
    } while (0)
#define PHASE_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define PHASE_BEGIN(cls_)
#define PHASE_MARK(i_)
#define PHASE_WAIT_VMEM()
#endif


// ---- group policies -------------------------------------------------------------
template <int L>
struct SubWave {
    static_assert(L == 8 || L == 16 || L == 32 || L == 64, "sub-wave width");
    static constexpr int SIZE = L;
    static constexpr bool kIsBlock = false;
    u32 lane;       // index inside the group
    u32 base_lane;  // wave lane of group lane 0
    __device__ __forceinline__ SubWave()
    {
        const u32 wl = lane_id();
        lane = wl & (L - 1);
        base_lane = wl & ~u32(L - 1);
    }
    __device__ __forceinline__ void sync() const { wave_lds_fence(); }
    __device__ __forceinline__ u32 inclusive_scan(u32 v, u32* total, u32* /*scratch*/) const
    {
        if constexpr (L == 64) {
            v = wave_inclusive_scan(v);
            *total = (u32)__builtin_amdgcn_readlane((int)v, 63);
        } else if constexpr (L == 32) {
            v = row16_inclusive_scan(v);
            v += dpp_move<kDppRowBcast15, 0xA>(0, v);  // rows 1 and 3 take the total of rows 0 and 2
            // lane 31 of the own 32-lane half: ds_swizzle bit mode, lane' = (lane & 0) | 0x1F
            *total = (u32)__builtin_amdgcn_ds_swizzle((int)v, 0x1F << 5);
        } else if constexpr (L == 16) {
            v = row16_inclusive_scan(v);
            // lane 15 of the own 16-lane row: ds_swizzle bit mode, lane' = (lane & 0x10) | 0x0F
            *total = (u32)__builtin_amdgcn_ds_swizzle((int)v, 0x10 | (0x0F << 5));
        } else {
            // halves of a 16-lane DPP row: a shifted-in value from the other half is dropped
            u32 t = dpp_move<kDppRowShr + 1>(0, v);
            v += lane >= 1u ? t : 0u;
            t = dpp_move<kDppRowShr + 2>(0, v);
            v += lane >= 2u ? t : 0u;
            t = dpp_move<kDppRowShr + 4>(0, v);
            v += lane >= 4u ? t : 0u;
            // lane 7 of the own 8-lane group: lane' = (lane & 0x18) | 0x07
            *total = (u32)__builtin_amdgcn_ds_swizzle((int)v, 0x18 | (0x07 << 5));
        }
        return v;
    }
    __device__ __forceinline__ u32 reduce_add(u32 v, u32* scratch) const
    {
        u32 total;
        inclusive_scan(v, &total, scratch);
        return total;
    }
    // bit i of the result <=> group lane i voted true
    __device__ __forceinline__ u64 ballot(bool p) const
    {
        const u64 m = __ballot(p);
        if (L == 64) return m;
        return (m >> base_lane) & ((1ull << (L & 63)) - 1ull);
    }
    // contiguous slice of [0,total) walked by this lane: start, stride, end
    __device__ __forceinline__ void product_range(u32 total, u32& p0, u32& step, u32& end) const
    {
        p0 = lane;
        step = L;
        end = total;
    }
};

template <int THREADS>
struct Block {
    static constexpr int SIZE = THREADS;
    static constexpr bool kIsBlock = true;
    u32 lane;
    __device__ __forceinline__ Block() : lane(threadIdx.x) {}
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    __device__ __forceinline__ u32 inclusive_scan(u32 v, u32* total, u32* scratch) const
    {
        return block_exclusive_scan<THREADS>(v, scratch, total) + v;
    }
    __device__ __forceinline__ u32 reduce_add(u32 v, u32* scratch) const
    {
        v = wave_reduce_add(v);
        __syncthreads();
        if (lane_id() == 0) scratch[threadIdx.x >> 6] = v;
        __syncthreads();
        u32 s = 0;
#pragma unroll
        for (int w = 0; w < THREADS / 64; ++w) s += scratch[w];
        __syncthreads();
        return s;
    }
    // every wave owns a contiguous, 64-aligned slice of the product space and strides by 64
    // inside it: coalesced, and the owning A entry only moves forward (cheap hinted search)
    __device__ __forceinline__ void product_range(u32 total, u32& p0, u32& step, u32& end) const
    {
        constexpr u32 NW = THREADS / 64;
        const u32 per_wave = (((total + NW - 1) / NW) + 63u) & ~63u;
        const u32 w = threadIdx.x >> 6;
        const u32 lo = w * per_wave;
        p0 = lo + lane_id();
        step = 64;
        end = min(total, lo + per_wave);
    }
};

// ---- which rows of a class list a group walks ---------------------------------------------
// The lists are in ascending row order and neighbouring rows of A reference neighbouring rows of B
// (bands, communities, meshes).  Workgroup b of a launch runs on XCD b % 8 and every XCD has its own
// L2: with the plain mapping (row = block * groups + group, stride = all blocks) the eight L2s each
// fetch every B row.  XCD-aware mapping: the workgroups of one XCD walk a CONTIGUOUS eighth of the
// list, so a B row is fetched by (mostly) one L2.  Placement is only a speed assumption.
struct RowSlice {
    u32 idx, end, stride;
};
__device__ __forceinline__ RowSlice row_slice(u32 count, u32 bidx, u32 nblk, u32 groups, u32 gid, bool xcd_aware)
{
    if (!xcd_aware || nblk < 16u) return RowSlice{bidx * groups + gid, count, nblk * groups};
    const u32 first = blockIdx.x - bidx;            // first physical block of this class' range
    const u32 x = blockIdx.x & 7u;
    const u32 i0 = (x + 8u - (first & 7u)) & 7u;    // first block of the range with my residue
    const u32 nb_x = (nblk - i0 + 7u) >> 3;         // >= 1: nblk >= 16
    const u32 j = (bidx - i0) >> 3;
    const u32 lo = (u32)(u64(count) * x >> 3), hi = (u32)(u64(count) * (x + 1u) >> 3);
    return RowSlice{lo + j * groups + gid, hi, nb_x * groups};
}

// DeviceStats::capacity_miss ("this launch sequence is void") can be raised by ANOTHER workgroup or kernel while this workgroup
// starts -- a verifying body that meets a row it was not planned for, a scan that finds another nnz(C).  The lanes that
// synchronise with each other must take ONE decision on it: a wave that leaves while the others of its workgroup stay takes
// its share of every cooperative LDS write with it -- staging entries of the A row that hold whatever the previous kernel
// left in LDS, a gather at a wild index (found by the hostile-B suite under canary zones: nf_dense_kernel of a replayed
// sequence, a memory fault once in ~10 runs).  Workgroup groups: thread 0 decides for all (one barrier); sub-wave groups: per wave.
__device__ __forceinline__ bool block_void(u32 miss)
{
    __shared__ u32 s_void;  // thread 0's reading of the flag is the workgroup's: one LDS word, one barrier
    if (threadIdx.x == 0) s_void = miss;
    __syncthreads();
    return s_void != 0u;
}
__device__ __forceinline__ bool wave_void(u32 miss) { return __ballot(miss != 0u) != 0ull; }
template <class G>
__device__ __forceinline__ bool group_void(const G&, u32 miss)
{
    if constexpr (G::kIsBlock) return block_void(miss);
    else return wave_void(miss);
}

// First step of every class body: which rows of the class' list this group walks, and the first record.
// The list lives at a fixed place (launch.hpp, class_rec_at), so with a host-known count of the class (ClassGrid::cnt) the
// first record is requested AT ONCE, next to the device-side class table that confirms the count: a workgroup of the
// sub-wave classes lives for one or two rows, and the chain table -> record -> A entries -> B entries is most of that life.
struct RowCursor {
    const RowRec* recs;
    u32 m, cls;
    u32 idx, end, stride;
    RowRec next;  // record of list entry idx (if idx < end)
    u32 miss;     // the (replayed) launch sequence was declared void by an earlier kernel: walk nothing
    __device__ __forceinline__ bool more() const { return idx < end; }
    __device__ __forceinline__ RowRec at(u32 i) const { return *class_rec_at(recs, m, cls, i); }
    // the record of the current row; the cursor moves on (the next row's record is requested now)
    __device__ __forceinline__ RowRec take()
    {
        const RowRec r = next;
        if (idx + stride < end) next = at(idx + stride);
        idx += stride;
        return r;
    }
};
template <bool SYM>
__device__ __forceinline__ RowCursor open_list(const RowWork& w, int cls, u32 hint_cnt, u32 bidx, u32 nblk, u32 groups,
                                               u32 gid, bool xcd_aware)
{
    const BinTable& bt = SYM ? w.st->sym : w.st->num;
    RowCursor L;
    L.recs = w.recs;
    L.m = w.m;
    L.cls = (u32)cls;
    L.miss = w.st->capacity_miss;
    const u32 cnt = min(bt.count[cls], w.m);
    L.next = RowRec{};
    RowSlice rs{0u, 0u, 1u};
    const bool hinted = hint_cnt != kNoCount && hint_cnt <= w.m;
    if (hinted) {  // speculative: the record is in flight beside the table
        rs = row_slice(hint_cnt, bidx, nblk, groups, gid, xcd_aware);
        if (rs.idx < rs.end) L.next = L.at(rs.idx);
    }
    if (!hinted || cnt != hint_cnt) {
        rs = row_slice(cnt, bidx, nblk, groups, gid, xcd_aware);
        if (rs.idx < rs.end) L.next = L.at(rs.idx);
    }
    L.idx = rs.idx;
    L.end = rs.end;
    L.stride = rs.stride;
    return L;
}

// Per-group LDS staging area for one chunk of A entries (SIZE entries).
template <typename T>
struct RowMeta {
    u32* incl;  // inclusive prefix of B-row lengths
    u32* off;   // B-row start minus exclusive prefix: ib = off[s] + p  (u32 wrap-around)
    T* av;      // a_ik (unused by the symbolic kernels: pass nullptr)
    u32* win;   // win_words<G>() words of owner-window scratch (wave and workgroup groups)
};

constexpr int kBatch = 4;  // products per lane fetched before accumulating (memory-level parallelism)

// Where the products of a row come from.  b_sl holds, per entry of A, the first entry (x) and the length (y)
// of the B row it references: the analysis pass reads B.row_offsets for every A entry anyway and writes the
// pair out (8 B per A entry, ONE load for the class kernels), so the symbolic and numeric kernels load it
// coalesced next to a_ik instead of gathering B.row_offsets behind A.col_ids -- one dependent global round
// trip less per row.  Indexed by (absolute A entry - e_base).
template <typename T>
struct ProductSrc {
    const uint2* b_sl;  // (no __restrict__: the windowed walk of a multi-window row points this at w_sl, which
                        //  the same workgroup rewrites between two windows)
    const T* __restrict__ a_val;
    const u32* __restrict__ b_col;
    const T* __restrict__ b_val;
    uint2* w_sl;  // window cursors of multi-window rows (WindowCursors below), same indexing
    // rebase the per-entry arrays so that they can be indexed with absolute A entries
    __device__ __forceinline__ void rebase(const u32* a_row_offsets)
    {
        const u32 e_base = a_row_offsets[0];
        b_sl -= e_base;
        w_sl -= e_base;
    }
};

// ---- product -> owning A entry ---------------------------------------------------------
// Product p of a staged chunk belongs to the smallest entry s with incl[s] > p.
//
// Wave and workgroup groups: a wave walks a contiguous slice of the product space in windows of
// kBatch*64 products, lane l taking products base + 64u + l.  The owners of a whole window are
// found cooperatively instead of by 256 binary searches (which made these kernels VALU- and
// LDS-issue bound): the entries from s0 = owner(base) on drop a count at the window position
// where their products END (incl[e] - base; 16-bit counters, two per LDS word -- empty B rows
// share a position), and the owner of position i is s0 + the number of ends at positions <= i:
// one LDS read and one DPP scan per 64 products.  s0 is carried from window to window.
constexpr u32 kWinProducts = kBatch * 64;
constexpr u32 kWinWords = kWinProducts / 2;  // LDS words per wave

template <class G>
constexpr u32 win_words()
{
    return G::SIZE >= 64 ? (G::SIZE / 64) * kWinWords : kBatch * G::SIZE / 2;  // sub-wave groups: kBatch*SIZE positions
}

// owner(p), the same for all lanes of the wave (broadcast LDS reads, scalar control flow)
__device__ __forceinline__ u32 uniform_owner(const u32* incl, u32 cnt, u32 p)
{
    u32 lo = 0, hi = cnt - 1;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        const u32 v = (u32)__builtin_amdgcn_readfirstlane((int)incl[mid]);
        if (v > p) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// The 16-bit counters of a lane's kBatch positions (l, 64 + l, 128 + l, 192 + l) share one 8-byte LDS
// word pair, and the scans run on PACKED pairs of counters (two 16-bit fields per register: a field never
// exceeds the entries of a chunk) -- one LDS read and two DPP scans per window instead of four and four:
// these kernels are bound by VALU issue.
__device__ __forceinline__ void window_owners(const u32* incl, u32* win, u32 cnt, u32 base, u32& s0,
                                              u32 (&own)[kBatch])
{
    static_assert(kBatch == 4, "four counters per lane: one uint2");
    const u32 l = lane_id();
    reinterpret_cast<uint2*>(win)[l] = make_uint2(0u, 0u);
    wave_lds_fence();
    u32 sw = s0, at_end = 0;
    bool again;
    do {
        const u32 e = sw + l;
        const u32 b = e < cnt ? incl[e] - base : 0xFFFFFFFFu;  // >= 1: incl[s0] > base
        if (b < kWinProducts) {
            const u32 slot = ((b & 63u) << 2) | (b >> 6);  // counter of position b: lane (b & 63), batch (b >> 6)
            atomicAdd(&win[slot >> 1], 1u << ((slot & 1u) * 16u));
        }
        at_end += (u32)__popcll(__ballot(b == kWinProducts));
        again = (u32)__builtin_amdgcn_readlane((int)b, 63) <= kWinProducts;
        sw += 64;
    } while (again);
    wave_lds_fence();
    const uint2 x = reinterpret_cast<const uint2*>(win)[l];
    const u32 inc01 = wave_inclusive_scan(x.x), inc23 = wave_inclusive_scan(x.y);
    const u32 t01 = (u32)__builtin_amdgcn_readlane((int)inc01, 63), t23 = (u32)__builtin_amdgcn_readlane((int)inc23, 63);
    const u32 c1 = s0 + (t01 & 0xFFFFu), c2 = c1 + (t01 >> 16), c3 = c2 + (t23 & 0xFFFFu);
    own[0] = s0 + (inc01 & 0xFFFFu);
    own[1] = c1 + (inc01 >> 16);
    own[2] = c2 + (inc23 & 0xFFFFu);
    own[3] = c3 + (inc23 >> 16);
    s0 = c3 + (t23 >> 16) + at_end;
    wave_lds_fence();
}

// Walk all products of row [a0,a1) of A.  f(cols[kBatch], products[kBatch], nvalid).
template <bool WITH_VALUES, class G, typename T, typename F>
__device__ __forceinline__ void for_each_product(const G& g, const ProductSrc<T>& src, u32 a0, u32 a1,
                                                 const RowMeta<T>& m, u32* scratch, F&& f,
                                                 int dbg_cls = kMaxClasses - 1)
{
    PHASE_BEGIN(dbg_cls);
    u32 nlen = 0, nbs = 0;
    T nav = T(0);
    if (a0 + g.lane < a1) {
        const u32 e = a0 + g.lane;
        if (WITH_VALUES) nav = src.a_val[e];
        const uint2 sl = src.b_sl[e];
        nbs = sl.x;
        nlen = sl.y;
    }
    for (u32 chunk = a0; chunk < a1; chunk += G::SIZE) {
        const u32 cnt = min((u32)G::SIZE, a1 - chunk);
        const u32 len = nlen, bs = nbs;
        const T av = nav;
        u32 total;
        const u32 incl = g.inclusive_scan(len, &total, scratch);
        if (g.lane < cnt) {
            m.incl[g.lane] = incl;
            m.off[g.lane] = bs - (incl - len);
            if (WITH_VALUES) m.av[g.lane] = av;
        }
        g.sync();
        // the next chunk's entries are fetched while this chunk's products are walked (issued
        // here, with no older load pending, so that no wait is placed right behind them)
        nlen = 0;
        if (chunk + G::SIZE + g.lane < a1) {
            const u32 e = chunk + G::SIZE + g.lane;
            if (WITH_VALUES) nav = src.a_val[e];
            const uint2 sl = src.b_sl[e];
            nbs = sl.x;
            nlen = sl.y;
        }
        PHASE_MARK(10);
        u32 p, step, end;
        g.product_range(total, p, step, end);
        if constexpr (G::SIZE >= 64) {
            const u32 l = lane_id();
            u32 base = (u32)__builtin_amdgcn_readfirstlane((int)(p - l));
            const u32 wend = (u32)__builtin_amdgcn_readfirstlane((int)end);
            u32* win = m.win + (G::kIsBlock ? (threadIdx.x >> 6) * kWinWords : 0u);
            u32 s0 = 0;
            if constexpr (!G::kIsBlock) {
                // (a wave per row: lane e holds incl[e] of the chunk -- the owner of `base` is a ballot, not a binary
                //  search of five dependent LDS reads)
                const u64 behind = g.ballot(g.lane < cnt && incl > base);
                s0 = behind ? (u32)__builtin_ctzll(behind) : cnt - 1u;
            } else if (base < wend) {
                s0 = uniform_owner(m.incl, cnt, base);
            }
            PHASE_MARK(11);
            while (base < wend) {
                u32 own[kBatch];
                window_owners(m.incl, win, cnt, base, s0, own);
                PHASE_MARK(14);
                u32 c[kBatch];
                T bv[kBatch], a[kBatch], prod[kBatch];
                if (base + kWinProducts <= wend) {
                    // a FULL window (wave-uniform): no predicate on any lane -- the LDS reads of the four owners'
                    // offsets go out together, the gathers behind them, and `f` is compiled for a constant count
                    // (these loops are bound by VALU issue: the predicated form below spends an eighth of its
                    // instructions on the tail it only has in a row's last window)
                    u32 off[kBatch];
#pragma unroll
                    for (int u = 0; u < kBatch; ++u) off[u] = m.off[own[u]];
#pragma unroll
                    for (int u = 0; u < kBatch; ++u) {
                        const u32 ib = off[u] + base + u * 64 + l;
                        c[u] = src.b_col[ib];
                        bv[u] = T(0);
                        a[u] = T(0);
                        if (WITH_VALUES) {
                            bv[u] = src.b_val[ib];
                            a[u] = m.av[own[u]];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kBatch; ++u) prod[u] = a[u] * bv[u];
                    PHASE_WAIT_VMEM();
                    PHASE_MARK(15);
                    f(c, prod, (u32)kBatch);
                    PHASE_MARK(12);
                    base += kWinProducts;
                    continue;
                }
                u32 nvalid = 0;
                // all kBatch gathers are issued before the first product is formed
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const u32 pu = base + u * 64 + l;
                    c[u] = kEmptyKey;
                    bv[u] = T(0);
                    a[u] = T(0);
                    if (pu < wend) {
                        const u32 ib = m.off[own[u]] + pu;
                        c[u] = src.b_col[ib];
                        if (WITH_VALUES) {
                            bv[u] = src.b_val[ib];
                            a[u] = m.av[own[u]];
                        }
                        ++nvalid;  // the valid products of a lane are a prefix in u
                    }
                }
#pragma unroll
                for (int u = 0; u < kBatch; ++u)
                    prod[u] = a[u] * bv[u];  // rounded product, added later (no FMA across the add)
                PHASE_WAIT_VMEM();
                PHASE_MARK(15);
                f(c, prod, nvalid);
                PHASE_MARK(12);
                base += kWinProducts;
            }
        } else {
            // 16- and 32-lane groups: the same end-count scheme on a window of kBatch*SIZE products.
            // The chunk has at most SIZE entries and lane e holds incl[e] in a register, so the
            // ends need no LDS read and the entries before the window are a ballot.
            constexpr u32 L = G::SIZE;
            static_assert(L == 8 || L == 16 || L == 32, "sub-wave groups");
            PHASE_MARK(11);
            u32* win = m.win;  // kBatch*L 16-bit counters of this group
            const bool mine = g.lane < cnt;
            for (u32 base = 0; base < total; base += kBatch * L) {
                reinterpret_cast<uint2*>(win)[g.lane] = make_uint2(0u, 0u);  // my four counters
                wave_lds_fence();
                const u32 before = (u32)__popcll(g.ballot(mine && incl <= base));
                const u32 b = incl - base;  // position where my entry's products end
                if (mine && incl > base && b < kBatch * L) {
                    const u32 slot = ((b & (L - 1u)) << 2) | (b / L);  // lane (b mod L), batch (b / L)
                    atomicAdd(&win[slot >> 1], 1u << ((slot & 1u) * 16u));
                }
                wave_lds_fence();
                // packed pairs of 16-bit counters: two group scans instead of four
                const uint2 x = reinterpret_cast<const uint2*>(win)[g.lane];
                u32 t01, t23;
                const u32 inc01 = g.inclusive_scan(x.x, &t01, nullptr), inc23 = g.inclusive_scan(x.y, &t23, nullptr);
                u32 ownv[kBatch];
                ownv[0] = before + (inc01 & 0xFFFFu);
                ownv[1] = before + (t01 & 0xFFFFu) + (inc01 >> 16);
                ownv[2] = before + (t01 & 0xFFFFu) + (t01 >> 16) + (inc23 & 0xFFFFu);
                ownv[3] = before + (t01 & 0xFFFFu) + (t01 >> 16) + (t23 & 0xFFFFu) + (inc23 >> 16);
                PHASE_MARK(14);
                u32 c[kBatch];
                T bv[kBatch], a[kBatch], prod[kBatch];
                // (a full window for EVERY group of the wave that is still walking: the unpredicated form, as above)
                if (__ballot(base + kBatch * L <= total) == __ballot(true)) {
                    u32 off[kBatch];
#pragma unroll
                    for (int u = 0; u < kBatch; ++u) off[u] = m.off[ownv[u]];
#pragma unroll
                    for (int u = 0; u < kBatch; ++u) {
                        const u32 ib = off[u] + base + u * L + g.lane;
                        c[u] = src.b_col[ib];
                        bv[u] = T(0);
                        a[u] = T(0);
                        if (WITH_VALUES) {
                            bv[u] = src.b_val[ib];
                            a[u] = m.av[ownv[u]];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kBatch; ++u) prod[u] = a[u] * bv[u];
                    PHASE_WAIT_VMEM();
                    PHASE_MARK(15);
                    f(c, prod, (u32)kBatch);
                    wave_lds_fence();
                    PHASE_MARK(12);
                    continue;
                }
                u32 nvalid = 0;
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const u32 pu = base + u * L + g.lane;
                    const u32 own = ownv[u];
                    c[u] = kEmptyKey;
                    bv[u] = T(0);
                    a[u] = T(0);
                    if (pu < total) {
                        const u32 ib = m.off[own] + pu;
                        c[u] = src.b_col[ib];
                        if (WITH_VALUES) {
                            bv[u] = src.b_val[ib];
                            a[u] = m.av[own];
                        }
                        ++nvalid;
                    }
                }
#pragma unroll
                for (int u = 0; u < kBatch; ++u) prod[u] = a[u] * bv[u];
                PHASE_WAIT_VMEM();
                PHASE_MARK(15);
                f(c, prod, nvalid);
                wave_lds_fence();
                PHASE_MARK(12);
            }
        }
        PHASE_MARK(12);
        g.sync();
        PHASE_MARK(13);
    }
}

// ---- column windows over a heavy row: per-entry cursors ---------------------------------
// A row whose reachable column range exceeds one LDS window (dense accumulator, bitmap) is produced
// window by window.  B rows are sorted, so the B entries of A entry e that fall into the window are a
// contiguous run behind the entry's cursor: every B entry is READ ONCE per row however many windows
// there are (role of the reference's per-A-nnz resume cursors, include/GPU/spECK_HashSpGEMM.cuh:
// 1175-1298, 1475-1569; kept in LDS there, spilled to a global cursor pool for long A rows, :1600-1619).
// Here the cursors always live in global memory -- w_sl, one (start, length) pair per A entry, touched only
// by the workgroup that owns the row -- because the windowed run lengths then plug into the same
// flattened product walk (for_each_product reads them where it otherwise reads b_sl).
// A window starts at the SMALLEST column not yet consumed, so no window is empty.
template <int THREADS>
struct WindowCursors {
    const uint2* __restrict__ b_sl;  // rebased like ProductSrc's
    const u32* __restrict__ b_col;
    uint2* __restrict__ w_sl;
    u32 a0, a1;

    __device__ __forceinline__ void reset() const
    {
        for (u32 e = a0 + threadIdx.x; e < a1; e += THREADS) w_sl[e] = make_uint2(b_sl[e].x, 0u);
        __syncthreads();
    }
    // advances every cursor past the previous window and returns the first column of the next one
    // (0xFFFFFFFF: the row is finished); `s_min` is one LDS word + THREADS/64 words of scratch
    __device__ __forceinline__ u32 next_window(u32 wcols, u32* s_red) const
    {
        u32 mn = 0xFFFFFFFFu;
        for (u32 e = a0 + threadIdx.x; e < a1; e += THREADS) {
            const uint2 ws = w_sl[e], bs = b_sl[e];
            const u32 lo = ws.x + ws.y, hi = bs.x + bs.y;
            if (lo < hi) mn = min(mn, b_col[lo]);
        }
        mn = wave_reduce_min(mn);
        __syncthreads();
        if (lane_id() == 0) s_red[threadIdx.x >> 6] = mn;
        __syncthreads();
        mn = 0xFFFFFFFFu;
#pragma unroll
        for (int w = 0; w < THREADS / 64; ++w) mn = min(mn, s_red[w]);
        __syncthreads();
        if (mn == 0xFFFFFFFFu) return mn;
        const u64 wend = u64(mn) + wcols;  // exclusive
        for (u32 e = a0 + threadIdx.x; e < a1; e += THREADS) {
            const uint2 ws = w_sl[e], bs = b_sl[e];
            const u32 first = ws.x + ws.y, hi = bs.x + bs.y;
            u32 lo = first, up = hi;  // first entry with column >= wend: gallop from the cursor (a window
            u32 stepw = 1;            //   usually takes a few entries of a B row), then bisect
            while (lo < up) {
                const u32 probe = min(lo + stepw - 1u, up - 1u);
                if (u64(b_col[probe]) < wend) {
                    lo = probe + 1;
                    stepw <<= 1;
                } else {
                    up = probe;
                    break;
                }
            }
            while (lo < up) {
                const u32 mid = lo + ((up - lo) >> 1);
                if (u64(b_col[mid]) < wend) lo = mid + 1; else up = mid;
            }
            w_sl[e] = make_uint2(first, lo - first);
        }
        __syncthreads();  // the walk below reads other threads' w_sl (same CU, same L1)
        return mn;
    }
};

// ---- open-addressed structures (LDS or global memory) ---------------------------------
// Batched forms: the first compare-and-swap of the kBatch products are issued back to back
// (independent atomics in flight); only a collision enters the probing loop.
// Collisions are resolved by DOUBLE HASHING when SPECK_PROBE_DOUBLE is set (probe step = an odd number derived
// from the key: coprime with the power-of-two capacity): the lanes of a wave leave their probing loops together,
// so what a batch costs is the LONGEST chain among 64 x kBatch keys -- and linear probing at a load of 2/3 grows
// long primary clusters.
#ifndef SPECK_PROBE_DOUBLE
#define SPECK_PROBE_DOUBLE 1
#endif
// The FIRST retry still goes to the neighbouring slot: consecutive columns (stencils, bands) fall on a low-
// discrepancy sequence under the multiplicative hash and the neighbour is usually free -- the nlpkkt stand-in lost
// 9 % with pure double hashing (29.6 ms against 27.0 with linear probing) and keeps 27.9 this way, while the
// stand-ins with scattered columns keep what double hashing gave them (mac_econ -6 %, webbase -5 %, scircuit -2 %).
#ifndef SPECK_FIRST_PROBE_INC
#define SPECK_FIRST_PROBE_INC 1
#endif
constexpr u32 kFirstProbeInc = SPECK_FIRST_PROBE_INC;  // 0: the key's step from the first retry on
__device__ __forceinline__ u32 probe_step(u32 key, u32 shift)
{
#if SPECK_PROBE_DOUBLE
    return ((key * 0x85EBCA6Bu) >> shift) | 1u;
#else
    (void)key;
    (void)shift;
    return 1u;
#endif
}

template <u32 CAP>
__device__ __forceinline__ u32 set_insert_batch(u32* tab, const u32 (&key)[kBatch], u32 nvalid)
{
    u32 slot[kBatch], old[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
        slot[u] = hash_slot<CAP>(key[u]);
        old[u] = kEmptyKey;
        if ((u32)u < nvalid) old[u] = atomicCAS(&tab[slot[u]], kEmptyKey, key[u]);
    }
    u32 added = 0;
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
        if ((u32)u >= nvalid) continue;
        if (old[u] != kEmptyKey && old[u] != key[u]) {
            const u32 step = probe_step(key[u], 32u - (u32)__builtin_ctz(CAP));
            u32 inc = kFirstProbeInc ? kFirstProbeInc : step;
            do {
                slot[u] = (slot[u] + inc) & (CAP - 1);
                inc = step;
                old[u] = atomicCAS(&tab[slot[u]], kEmptyKey, key[u]);
            } while (old[u] != kEmptyKey && old[u] != key[u]);
        }
        added += old[u] == kEmptyKey ? 1u : 0u;
    }
    return added;
}

// The table of a row uses the first 2^bits slots of the class' arrays (bits chosen per row from
// its exact nnz: every later pass over the slots costs LDS and VALU issue per SLOT, not per key).
// BOUNDED (numeric launches of a sequence that verifies the row lengths itself, RowWork::verify_numeric): the table is
// sized from the nnz the PREVIOUS identical call found -- if the row has more distinct columns now, a key may find
// neither itself nor a free slot.  From its second probe on a key walks home, home + step, home + 2 step, ... (odd step,
// power-of-two table): every slot once, then `home` again -- a key that comes back there has seen the whole table and gives
// up (its product is dropped; the caller rejects the replay).  The retry loop is the hottest loop of these kernels (every
// batch of a wave enters it, for the longest chain among its lanes): the bound costs it one compare.  (Measured and
// dropped: a trip counter with a break, per lane or per wave on the ballot of pending lanes -- the loop doubled in
// instructions and the launch took 20-30 % longer.)
// Returns true if any key of this lane gave up.
template <typename T, bool BOUNDED = false>
__device__ __forceinline__ bool table_accumulate_batch(u32* keys, T* vals, u32 bits, const u32 (&key)[kBatch],
                                                       const T (&prod)[kBatch], u32 nvalid)
{
    const u32 mask = (1u << bits) - 1u;
    u32 slot[kBatch], old[kBatch];
    bool gave_up = false;
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
        slot[u] = (key[u] * 0x9E3779B1u) >> (32u - bits);
        old[u] = kEmptyKey;
        if ((u32)u < nvalid) old[u] = atomicCAS(&keys[slot[u]], kEmptyKey, key[u]);
    }
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
        if ((u32)u >= nvalid) continue;
        if (old[u] != kEmptyKey && old[u] != key[u]) {
            const u32 step = probe_step(key[u], 32u - bits);
            if constexpr (BOUNDED) {
                slot[u] = (slot[u] + (kFirstProbeInc ? kFirstProbeInc : step)) & mask;
                old[u] = atomicCAS(&keys[slot[u]], kEmptyKey, key[u]);
                if (old[u] != kEmptyKey && old[u] != key[u]) {
                    const u32 home = slot[u];
                    do {
                        slot[u] = (slot[u] + step) & mask;
                        old[u] = atomicCAS(&keys[slot[u]], kEmptyKey, key[u]);
                    } while (old[u] != kEmptyKey && old[u] != key[u] && slot[u] != home);
                }
            } else {
                u32 inc = kFirstProbeInc ? kFirstProbeInc : step;
                do {
                    slot[u] = (slot[u] + inc) & mask;
                    inc = step;
                    old[u] = atomicCAS(&keys[slot[u]], kEmptyKey, key[u]);
                } while (old[u] != kEmptyKey && old[u] != key[u]);
            }
        }
        if (!BOUNDED || old[u] == kEmptyKey || old[u] == key[u]) atomicAdd(&vals[slot[u]], prod[u]);
        else gave_up = true;
    }
    return gave_up;
}

// Bitmap marks of a batch of products: neighbouring lanes hold neighbouring columns of one B row, i.e. mostly
// the SAME bitmap word -- up to 32 lanes per address for every ds_or.  The bits of equal neighbouring words are
// OR-ed together inside the 16-lane DPP rows first (OR is idempotent, so merging runs that are not adjacent is
// harmless) and only the last lane of a run touches the LDS.  `word` = 0xFFFFFFFF for a lane with nothing to mark.
__device__ __forceinline__ void bitmap_or_runs(u32* bm, u32 word, u32 bits)
{
#define SPECK_OR_STEP(S_)                                                             \
    {                                                                                 \
        const bool same = dpp_move<kDppRowShr + S_>(0xFFFFFFFEu, word) == word;       \
        const u32 t = dpp_move<kDppRowShr + S_>(0u, bits);                            \
        bits |= same ? t : 0u;                                                        \
    }
    SPECK_OR_STEP(1)
    SPECK_OR_STEP(2)
    SPECK_OR_STEP(4)
    SPECK_OR_STEP(8)
#undef SPECK_OR_STEP
    const bool tail = dpp_move<kDppRowShl + 1>(0xFFFFFFFEu, word) != word;
    if (tail && word != 0xFFFFFFFFu) atomicOr(&bm[word], bits);
}

// Exclusive prefix of popcounts over bm[0..nwords) into pref[]; returns the total.
// Every lane owns a contiguous run of words.
// STRIDE 2: words and prefixes interleaved ({word, prefix} pairs: bm = base, pref = base + 1), so that a
// later rank lookup is ONE 8-byte LDS read.
template <class G, u32 STRIDE = 1>
__device__ __forceinline__ u32 bitmap_prefix(const G& g, const u32* bm, u32* pref, u32 nwords,
                                             u32* scratch)
{
    const u32 wpt = (nwords + G::SIZE - 1) / G::SIZE;
    const u32 w_begin = min(g.lane * wpt, nwords), w_end = min(w_begin + wpt, nwords);
    u32 local = 0;
    for (u32 i = w_begin; i < w_end; ++i) local += __popc(bm[i * STRIDE]);
    u32 total;
    u32 run = g.inclusive_scan(local, &total, scratch) - local;
    for (u32 i = w_begin; i < w_end; ++i) {
        pref[i * STRIDE] = run;
        run += __popc(bm[i * STRIDE]);
    }
    g.sync();
    return total;
}

}  // namespace speck
