// row_groups.hpp -- the lane-group abstraction shared by the symbolic and numeric kernels.
//
// A row of C is produced by a GROUP: SubWave<16> (16 lanes of a wave), SubWave<64> (a wave) or
// Block<THREADS> (a workgroup).  All groups run the same three steps:
//   1. stage the A row's metadata in LDS: for each entry a_ik the inclusive prefix of
//      nnz(B_k), the B-row start rebased to that prefix, and a_ik itself;
//   2. walk the FLATTENED product space p in [0, ops): lane -> p, the owning A entry is found
//      by a (hinted) binary search in the LDS prefix; consecutive lanes read consecutive
//      B entries (coalesced) and every lane has the same amount of work whatever the B-row
//      length distribution is (the reference balances this with getThreadShiftNew,
//      include/common.cuh:509-555, which assumes near-uniform B rows);
//   3. the loads of product p+stride are issued before product p is accumulated, so the
//      dependent chain LDS-search -> global load -> LDS atomic of successive products overlaps.
#pragma once
#include "device_common.hpp"

namespace speck {

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
#pragma unroll
        for (int off = 1; off < L; off <<= 1) {
            const u32 t = (u32)__shfl_up((int)v, off, L);
            if (lane >= (u32)off) v += t;
        }
        *total = (u32)__shfl((int)v, L - 1, L);
        return v;
    }
    __device__ __forceinline__ u32 reduce_add(u32 v, u32* /*scratch*/) const
    {
#pragma unroll
        for (int off = L >> 1; off > 0; off >>= 1) v += (u32)__shfl_xor((int)v, off, L);
        return v;
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

// Workgroup-per-row classes pull their next row from a device-side queue (rows of the heavy
// classes differ a lot in cost; a static stride leaves CUs idle).  One returning device-scope
// atomic per row (~0.3-1 us, MI355X_MICROARCH "dequeue") against >= 5 us of row time.
__device__ __forceinline__ u32 next_queued_row(u32* queue_head, u32* lds_slot)
{
    __syncthreads();
    if (threadIdx.x == 0) *lds_slot = atomicAdd(queue_head, 1u);
    __syncthreads();
    return *lds_slot;
}

// Per-group LDS staging area for one chunk of A entries (SIZE entries).
template <typename T>
struct RowMeta {
    u32* incl;  // inclusive prefix of B-row lengths
    u32* off;   // B-row start minus exclusive prefix: ib = off[s] + p  (u32 wrap-around)
    T* av;      // a_ik (unused by the symbolic kernels: pass nullptr)
};

template <int SIZE, typename T>
constexpr u32 row_meta_bytes(bool with_values)
{
    return SIZE * (8 + (with_values ? (u32)sizeof(T) : 0));
}

constexpr int kBatch = 4;  // products per lane fetched before accumulating (memory-level parallelism)

// Smallest s in [lo, cnt) with incl[s] > p.  `lo` is a lower bound carried between calls.
__device__ __forceinline__ u32 owner_search(const u32* incl, u32 lo, u32 cnt, u32 p)
{
    if (incl[lo] > p) return lo;
    u32 hi = cnt - 1;
    ++lo;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (incl[mid] > p) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// Where the products of a row come from.  b_start / b_len hold, per entry of A, the first entry
// and the length of the B row it references: the analysis pass reads B.row_offsets for every A
// entry anyway and writes them out (8 B per A entry), so the symbolic and numeric kernels load
// them coalesced next to a_ik instead of gathering B.row_offsets behind A.col_ids -- one
// dependent global round trip less per row.  Both are indexed by (absolute A entry - e_base).
template <typename T>
struct ProductSrc {
    const u32* __restrict__ b_start;
    const u32* __restrict__ b_len;
    const T* __restrict__ a_val;
    const u32* __restrict__ b_col;
    const T* __restrict__ b_val;
    // rebase the per-entry arrays so that they can be indexed with absolute A entries
    __device__ __forceinline__ void rebase(const u32* a_row_offsets)
    {
        const u32 e_base = a_row_offsets[0];
        b_start -= e_base;
        b_len -= e_base;
    }
};

// Walk all products of row [a0,a1) of A.  f(col, value) for WITH_VALUES, f(col) otherwise.
template <bool WITH_VALUES, class G, typename T, typename F>
__device__ __forceinline__ void for_each_product(const G& g, const ProductSrc<T>& src, u32 a0, u32 a1,
                                                 const RowMeta<T>& m, u32* scratch, F&& f)
{
    for (u32 chunk = a0; chunk < a1; chunk += G::SIZE) {
        const u32 cnt = min((u32)G::SIZE, a1 - chunk);
        u32 len = 0, bs = 0;
        T av = T(0);
        if (g.lane < cnt) {
            const u32 e = chunk + g.lane;
            if (WITH_VALUES) av = src.a_val[e];
            bs = src.b_start[e];
            len = src.b_len[e];
        }
        u32 total;
        const u32 incl = g.inclusive_scan(len, &total, scratch);
        if (g.lane < cnt) {
            m.incl[g.lane] = incl;
            m.off[g.lane] = bs - (incl - len);
            if (WITH_VALUES) m.av[g.lane] = av;
        }
        g.sync();
        u32 p, step, end;
        g.product_range(total, p, step, end);
        u32 s = 0;
        // kBatch products per lane are fetched back to back (independent gathers in flight)
        // before any of them touches the accumulator.
        while (p < end) {
            u32 c[kBatch];
            T bv[kBatch], av_[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const u32 pu = p + u * step;
                c[u] = kEmptyKey;
                bv[u] = T(0);
                av_[u] = T(0);
                if (pu < end) {
                    s = owner_search(m.incl, s, cnt, pu);
                    const u32 ib = m.off[s] + pu;
                    c[u] = src.b_col[ib];
                    if (WITH_VALUES) {
                        bv[u] = src.b_val[ib];
                        av_[u] = m.av[s];
                    }
                }
            }
            // the valid products of a batch are a prefix (p + u*step is increasing in u)
            u32 nvalid = 0;
            T prod[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                nvalid += (p + u * step < end) ? 1u : 0u;
                prod[u] = av_[u] * bv[u];  // rounded product, added later (no FMA across the add)
            }
            f(c, prod, nvalid);
            p += kBatch * step;
        }
        g.sync();
    }
}

// ---- open-addressed structures (LDS or global memory) ---------------------------------
// Batched forms: the first compare-and-swap of the kBatch products are issued back to back
// (independent atomics in flight); only a collision enters the probing loop.
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
        while (old[u] != kEmptyKey && old[u] != key[u]) {
            slot[u] = (slot[u] + 1) & (CAP - 1);
            old[u] = atomicCAS(&tab[slot[u]], kEmptyKey, key[u]);
        }
        added += old[u] == kEmptyKey ? 1u : 0u;
    }
    return added;
}

template <u32 CAP, typename T>
__device__ __forceinline__ void table_accumulate_batch(u32* keys, T* vals, const u32 (&key)[kBatch],
                                                       const T (&prod)[kBatch], u32 nvalid)
{
    u32 slot[kBatch], old[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
        slot[u] = hash_slot<CAP>(key[u]);
        old[u] = kEmptyKey;
        if ((u32)u < nvalid) old[u] = atomicCAS(&keys[slot[u]], kEmptyKey, key[u]);
    }
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
        if ((u32)u >= nvalid) continue;
        while (old[u] != kEmptyKey && old[u] != key[u]) {
            slot[u] = (slot[u] + 1) & (CAP - 1);
            old[u] = atomicCAS(&keys[slot[u]], kEmptyKey, key[u]);
        }
        atomicAdd(&vals[slot[u]], prod[u]);
    }
}

// Exclusive prefix of popcounts over bm[0..nwords) into pref[]; returns the total.
// Every lane owns a contiguous run of words.
template <class G>
__device__ __forceinline__ u32 bitmap_prefix(const G& g, const u32* bm, u32* pref, u32 nwords,
                                             u32* scratch)
{
    const u32 wpt = (nwords + G::SIZE - 1) / G::SIZE;
    const u32 w_begin = min(g.lane * wpt, nwords), w_end = min(w_begin + wpt, nwords);
    u32 local = 0;
    for (u32 i = w_begin; i < w_end; ++i) local += __popc(bm[i]);
    u32 total;
    u32 run = g.inclusive_scan(local, &total, scratch) - local;
    for (u32 i = w_begin; i < w_end; ++i) {
        pref[i] = run;
        run += __popc(bm[i]);
    }
    g.sync();
    return total;
}

}  // namespace speck
