// esc_wide.hpp -- the WIDE register classes: expand / sort / compress (esc.hpp) for rows of up to 128 products
// (32 lanes per row, two rows per wave: SYM_R32 / NUM_R32) and up to 256 products (a wave per row: SYM_R64 / NUM_R64).
//
// Why: after the 8- and 16-lane register classes the rows LEFT in the hash classes of the short-row inputs
// (scircuit / mac_econ stand-ins: 27-29 k rows, 90-98 % of them with <= 256 products) made a launch of their own that
// took as long as the launch of the 180 k small rows -- each such row lives ~29 k cycles in a hash class (2 k table
// clear + 15 k product walk in two windows + 12 k two-level bitmap sort), at 5 workgroups per CU.  In
// registers the row is ONE round of gathers (4 products per lane, all in flight), a sorting network, and a segmented
// sum: no table, no probing, no LDS atomics on the product path, 1.2 / 2.9 KB of LDS per row instead of 3.75 KB.
//
// What is new against esc.hpp:
//   * the sorting network crosses the 16-lane DPP rows: partner lane ^ 16 through v_permlane16_swap_b32, lane ^ 32
//     through v_permlane32_swap_b32 (both new on gfx950; one VALU instruction for the exchange of a whole register),
//     lane ^ 8 through row_ror:8.  The network is written as a loop over its stages (fetch_xor<mask>), so the 32-, 64-
//     128- and 256-element sorts are the same code;
//   * the owner of a product (which entry of A) comes from a bit mask of where the entries' products END that lives in
//     LDS (4 / 8 words, one ds_or per entry, {word, prefix} pairs as in the bitmap sorts) instead of in a register;
//   * the sort key is (column - first reachable column of the row) << 7 or 8 | product number: a condition on the row's
//     column RANGE (< 2^25 / 2^24, known from the analysis), not on cols(B);
//   * the segmented sums cross the DPP rows with row_bcast15 / row_bcast31 on (sum, stop) pairs; neighbours across a row
//     boundary come from wave_shr:1 / wave_shl:1.
// (Role of the reference's hash blocks for short rows, include/GPU/spECK_HashSpGEMM.cuh:591-738, 740-866, and of its
//  O(n^2) in-LDS rank sort, :813-865.)
#pragma once
#include <type_traits>

#include "device_common.hpp"
#include "launch.hpp"
#include "row_groups.hpp"
#include "esc.hpp"
#include "esc_rows.hpp"

namespace speck {

constexpr int kDppRowRor8 = 0x128, kDppWaveShl1 = 0x130, kDppWaveShr1 = 0x138;

// the value the lane (lane ^ M) holds (M inside a 16-lane DPP row)
template <u32 M>
__device__ __forceinline__ u32 fetch_xor(u32 v)
{
    if constexpr (M == 1) return dpp_fetch<kDppQuadXor1>(v);
    else if constexpr (M == 2) return dpp_fetch<kDppQuadXor2>(v);
    else if constexpr (M == 3) return dpp_fetch<kDppQuadMirror>(v);
    else if constexpr (M == 4) return esc_xor4(v);
    else if constexpr (M == 7) return dpp_fetch<kDppRowHalfMirror>(v);
    else if constexpr (M == 8) return dpp_fetch<kDppRowRor8>(v);
    else {
        static_assert(M == 15, "lane masks inside a DPP row");
        return dpp_fetch<kDppRowMirror>(v);
    }
}

// compare-exchange of x with the value `v` holds in lane ^ M: k = 0 keeps the minimum, k = ~0 the maximum.
// Across the DPP rows: v_permlane16_swap_b32 swaps the odd rows of its first operand with the even rows of its second
// (v_permlane32_swap_b32: the upper half with the lower half).  With both operands = v the two results hold, in EVERY
// lane, the value of the even-row lane and of the odd-row lane of the pair -- min / max do not care which is which, so the
// exchange is one swap + one median, no select.
template <u32 M>
__device__ __forceinline__ u32 esc_cx_xor(u32 x, u32 v, u32 k)
{
    if constexpr (M <= 15) return esc_med3(x, fetch_xor<M>(v), k);
    else if constexpr (M == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return esc_med3(r[0], r[1], k);  // (only when v IS x: the half-cleaners)
    } else {
        static_assert(M == 32, "half-cleaner masks");
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return esc_med3(r[0], r[1], k);
    }
}
// the value of lane ^ 16 / lane ^ 32 (the flips need the partner itself: it is another register's value)
__device__ __forceinline__ u32 fetch_xor16(u32 v)
{
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (lane_id() & 16u) ? r[0] : r[1];
}
__device__ __forceinline__ u32 fetch_xor32(u32 v)
{
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (lane_id() & 32u) ? r[0] : r[1];
}

// One stage of the bitonic network ("flip" form, element i = lane * 4 + register, every compare-exchange gives the
// lower index the minimum): merge sorted runs of 2^(K-1) elements into runs of 2^K.
template <u32 K>
__device__ __forceinline__ void esc_merge_stage(u32 (&x)[4], u32 gl)
{
    static_assert(K == 7 || K == 8, "the stages that leave the 16-lane row");
    // flip: partner index = i ^ (2^K - 1) -> lane ^ (2^(K-2) - 1), register 3 - r; lower <=> lane bit K-3 clear
    {
        const u32 k = esc_dir(gl, K - 3);
        u32 p[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            p[r] = fetch_xor16(fetch_xor<15>(x[3 - r]));    // lane ^ 31
            if constexpr (K == 8) p[r] = fetch_xor32(p[r]);  // lane ^ 63
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = esc_med3(x[r], p[r], k);
    }
    // half-cleaners of distance 2^j, j = K-2 .. 2: partner lane ^ 2^(j-2), same register; lower <=> lane bit j-2 clear
#define SPECK_HALF(J_)                                                               \
    if constexpr (K - 2 >= (J_)) {                                                   \
        constexpr u32 M = 1u << ((J_) - 2);                                          \
        const u32 k = esc_dir(gl, (J_) - 2);                                         \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) x[r] = esc_cx_xor<M>(x[r], x[r], k); \
    }
    SPECK_HALF(6)
    SPECK_HALF(5)
    SPECK_HALF(4)
    SPECK_HALF(3)
    SPECK_HALF(2)
#undef SPECK_HALF
    // distances 2, 1: inside the lane
    esc_cx(x[0], x[2]);
    esc_cx(x[1], x[3]);
    esc_cx(x[0], x[1]);
    esc_cx(x[2], x[3]);
}

// ascending sort of the 4 L elements of an L-lane group (L = 32, 64); `gl` = lane inside the group
template <u32 L>
__device__ __forceinline__ void esc_sort_wide(u32 (&x)[4], u32 gl)
{
    static_assert(L == 32 || L == 64, "32 or 64 lanes per row");
    esc_sort64(x, gl & 15u);  // every 16-lane DPP row ascending (stages 1 .. 6 never leave the row)
    esc_merge_stage<7>(x, gl);
    if constexpr (L == 64) esc_merge_stage<8>(x, gl);
}

// move a double one lane down / up across the whole wave (lanes without a source get `fill`)
template <int CTRL>
__device__ __forceinline__ double wave_move_f64(double fill, double v)
{
    return dpp_move_f64<CTRL>(fill, v);
}

// ---- the ends mask: bit k of an NP-bit LDS bitmap <=> a (non-empty) entry's products end at k.  {word, set bits before
// the word} pairs; the owner of product p is the number of set bits <= p.
template <u32 L>
struct EscEnds {
    static constexpr u32 NP = 4u * L, NW = NP / 32u;
    uint2* pref;  // [NW]
    template <class G>
    __device__ __forceinline__ void build(const G& g, bool nonempty, u32 incl) const
    {
        if (g.lane < NW) pref[g.lane].x = 0u;
        wave_lds_fence();
        if (nonempty && incl < NP) atomicOr(&pref[incl >> 5].x, 1u << (incl & 31u));
        wave_lds_fence();
        const u32 w = g.lane < NW ? pref[g.lane].x : 0u;
        const u32 pc = (u32)__popc(w);
        u32 total;
        const u32 inc = g.inclusive_scan(pc, &total, nullptr);
        if (g.lane < NW) pref[g.lane].y = inc - pc;
        wave_lds_fence();
    }
    __device__ __forceinline__ u32 owner(u32 p) const
    {
        const uint2 e = pref[p >> 5];
        return e.y + (u32)__popc(e.x & ((2u << (p & 31u)) - 1u));  // (2 << 31 wraps to 0: all 32 bits)
    }
};

// LDS per group: products by number | a_ik of the non-empty entries | their B-row offsets | ends mask
template <typename T, u32 L>
constexpr u32 num_escw_group_lds()
{
    return 4u * L * (u32)sizeof(Acc<T>) + L * ((u32)sizeof(Acc<T>) + 4u) + (4u * L / 32u) * 8u;
}
template <u32 L>
constexpr u32 sym_escw_group_lds()
{
    return L * 4u + (4u * L / 32u) * 8u;
}

// ------------------------------------------------------------------ NUM_R32 / NUM_R64 (and their FUSED form, esc_rows.hpp)
// One row by its lane group (see num_esc_row): `cmin` = the first column the row can reach (the sort key holds the
// column relative to it).  Returns the row's nnz; the row is stored if nnz <= room.
template <typename T, u32 L>
__device__ __forceinline__ u32 num_escw_row(const SubWave<L>& g, unsigned char* mine, const ProductSrc<T>& src, u32 a0, u32 a1,
                                            u32 cmin, u32* __restrict__ out_col, T* __restrict__ out_val, u32 room)
{
    static_assert(L == 32 || L == 64, "32 or 64 lanes per row");
    constexpr u32 PER = kEscPerLane, NP = PER * L, TAG = L == 32 ? 7u : 8u;
    static_assert((1u << TAG) == NP, "the product number fills the tag");
    static_assert(TAG + (L == 32 ? kNumEsc32RangeBits : kNumEsc64RangeBits) == 32, "key = column offset << TAG | product");
    Acc<T>* s_vals = reinterpret_cast<Acc<T>*>(mine);     // [4 L] products by number
    Acc<T>* s_av = s_vals + NP;                           // [L]   a_ik of the j-th non-empty entry
    u32* s_off = reinterpret_cast<u32*>(s_av + L);        // [L]   its B-row start minus its first product number
    const EscEnds<L> ends{reinterpret_cast<uint2*>(s_off + L)};
    const u32 gl = g.lane;
    // ---- expand: my entry of A, where its products end
    const bool have = a0 + gl < a1;
    uint2 sl = make_uint2(0u, 0u);
    Acc<T> av = 0;
    if (have) {
        sl = src.b_sl[a0 + gl];
        av = (Acc<T>)src.a_val[a0 + gl];
    }
    u32 total;
    const u32 incl = g.inclusive_scan(sl.y, &total, nullptr);
    const bool nonempty = sl.y != 0;
    const u32 before = (u32)__popcll(g.ballot(nonempty) & ((1ull << gl) - 1ull));  // non-empty entries before mine
    if (nonempty) {
        s_off[before] = sl.x - (incl - sl.y);
        s_av[before] = av;
    }
    ends.build(g, nonempty, incl);  // (fences inside: the staging above is visible as well)
    u32 key[PER];
#pragma unroll
    for (u32 u = 0; u < PER; ++u) {
        const u32 p = u * L + gl;
        key[u] = kEscInvalid;
        if (p < total) {
            const u32 j = ends.owner(p);
            const u32 ib = s_off[j] + p;
            const u32 c = src.b_col[ib];
            const T bv = src.b_val[ib];
            const T prod = (T)s_av[j] * bv;  // rounded product, added later (no FMA across the add)
            s_vals[p] = (Acc<T>)prod;
            key[u] = ((c - cmin) << TAG) | p;
        }
    }
    wave_lds_fence();
    // ---- sort by (column, product number)
    esc_sort_wide<L>(key, gl);
    // ---- compress: sums of the runs of equal columns, in sorted order (validity by position: esc_rows.hpp)
    u32 col[PER];
    Acc<T> sum[PER];
#pragma unroll
    for (u32 r = 0; r < PER; ++r) {
        const bool valid = gl * PER + r < total;
        col[r] = valid ? key[r] >> TAG : kEscInvalid;
        sum[r] = valid ? s_vals[key[r] & (NP - 1u)] : Acc<T>(0);
    }
    bool lead[PER];  // element r continues the run of element 0 of this lane
    lead[0] = true;
#pragma unroll
    for (u32 r = 1; r < PER; ++r) {
        const bool same = col[r] == col[r - 1] && col[r] != kEscInvalid;
        lead[r] = lead[r - 1] && same;
        sum[r] += same ? sum[r - 1] : Acc<T>(0);
    }
    // across the lanes: what the lanes before me contribute to the run my element 0 continues
    const u32 prev_col = dpp_move<kDppWaveShr1>(kEscInvalid, col[PER - 1]);
    const bool cont = gl != 0 && col[0] != kEscInvalid && prev_col == col[0];
    Acc<T> chain = sum[PER - 1];             // running sum of the run that ends this lane
    bool stop = !(lead[PER - 1] && cont);    // ... which does not reach back into the lane before
    // segmented scan of (chain, stop) inside every 16-lane row (Kogge-Stone; a lane whose source would lie in the
    // row before keeps what it has: it already covers its row from the start) ...
    const u32 rl = gl & 15u;
#define SPECK_SEG_ROW(D_)                                                        \
    {                                                                            \
        const Acc<T> t = dpp_move_f64<kDppRowShr + D_>(0.0, chain);              \
        const bool ts = dpp_move<kDppRowShr + D_>(1u, (u32)stop) != 0;           \
        const bool take = !stop && rl >= D_;                                     \
        chain += take ? t : Acc<T>(0);                                           \
        stop = take ? ts : stop;                                                 \
    }
    SPECK_SEG_ROW(1)
    SPECK_SEG_ROW(2)
    SPECK_SEG_ROW(4)
    SPECK_SEG_ROW(8)
#undef SPECK_SEG_ROW
    // ... then across the rows of the group: lane 15 of the row before (rows 1 and 3), lane 31 (rows 2 and 3)
    {
        const Acc<T> t = __longlong_as_double(
            (u64(dpp_move<kDppRowBcast15, 0xA>(0u, (u32)(__double_as_longlong(chain) >> 32))) << 32) |
            dpp_move<kDppRowBcast15, 0xA>(0u, (u32)__double_as_longlong(chain)));
        const bool ts = dpp_move<kDppRowBcast15, 0xA>(1u, (u32)stop) != 0;
        const bool take = !stop && (gl & 16u) != 0;
        chain += take ? t : Acc<T>(0);
        stop = take ? ts : stop;
    }
    if constexpr (L == 64) {
        const Acc<T> t = __longlong_as_double(
            (u64(dpp_move<kDppRowBcast31, 0xC>(0u, (u32)(__double_as_longlong(chain) >> 32))) << 32) |
            dpp_move<kDppRowBcast31, 0xC>(0u, (u32)__double_as_longlong(chain)));
        const bool ts = dpp_move<kDppRowBcast31, 0xC>(1u, (u32)stop) != 0;
        const bool take = !stop && (gl & 32u) != 0;
        chain += take ? t : Acc<T>(0);
        stop = take ? ts : stop;
    }
    const Acc<T> from_prev = wave_move_f64<kDppWaveShr1>(0.0, chain);
    const Acc<T> carry = cont ? from_prev : Acc<T>(0);
#pragma unroll
    for (u32 r = 0; r < PER; ++r) sum[r] += lead[r] ? carry : Acc<T>(0);
    // the last element of a run carries the entry; its rank = runs that end before it
    const u32 next_col = dpp_move<kDppWaveShl1>(kEscInvalid, col[0]);
    bool tail[PER];
    u32 ntail = 0;
#pragma unroll
    for (u32 r = 0; r < PER; ++r) {
        const u32 after = r + 1 < PER ? col[r + 1] : (gl == L - 1 ? kEscInvalid : next_col);
        tail[r] = col[r] != kEscInvalid && after != col[r];
        ntail += tail[r] ? 1u : 0u;
    }
    u32 all;
    u32 pos = g.inclusive_scan(ntail, &all, nullptr) - ntail;
    const bool write = all <= room;
#pragma unroll
    for (u32 r = 0; r < PER; ++r)
        if (tail[r] && write) {
            out_col[pos] = col[r] + cmin;
            out_val[pos] = (T)sum[r];
            ++pos;
        }
    wave_lds_fence();  // the next row overwrites the staging and the products
    return all;
}

template <typename T, u32 L, int THREADS, bool FUSED = false>
__device__ __forceinline__ void num_escw_body(unsigned char* smem, const ProductSrc<T>& src, const RowWork& w,
                                              u32* __restrict__ c_col, T* __restrict__ c_val, int cls, u32 bidx,
                                              u32 nblk, u32 hint = kNoCount, u32* __restrict__ counts = nullptr)
{
    using G = SubWave<L>;
    constexpr u32 NG = THREADS / L;
    const G g;
    const u32 gid = threadIdx.x / L;
    unsigned char* mine = smem + gid * num_escw_group_lds<T, L>();
    RowCursor cur = open_list<FUSED>(w, cls, hint, bidx, nblk, NG, gid, (w.xcd_aware & 8u) != 0);
    // (a replayed sequence that an earlier kernel has declared void walks nothing)
    if (wave_void(cur.miss)) return;
    while (cur.more()) {
        const RowRec rec = cur.take();  // (its successor's record is requested now: RowCursor)
        u32 place = rec.base, room = 0xFFFFFFFFu;
        if constexpr (FUSED) {
            place = w.nf_pred_off[rec.row];
            room = w.nf_pred_off[rec.row + 1] - place;
        }
        const u32 all = num_escw_row<T, L>(g, mine, src, rec.a0, rec.a1, rec.cmin, c_col + place, c_val + place, room);
        if constexpr (FUSED) {
            if (g.lane == 0) {
                counts[rec.row] = all;
                if (all != room) const_cast<DeviceStats*>(w.st)->capacity_miss = 1;
            }
        }
    }
}

// ------------------------------------------------------------------ SYM_R32 / SYM_R64: the columns sorted in registers,
// nnz = the number of places where the column changes
template <u32 L, int THREADS>
__device__ __forceinline__ void sym_escw_body(unsigned char* smem, const ProductSrc<float>& src, const RowWork& w,
                                              u32* __restrict__ counts, int cls, u32 bidx, u32 nblk,
                                              u32 hint = kNoCount)
{
    static_assert(L == 32 || L == 64, "32 or 64 lanes per row");
    using G = SubWave<L>;
    constexpr u32 NG = THREADS / L, PER = kEscPerLane;
    const G g;
    const u32 gid = threadIdx.x / L;
    u32* s_off = reinterpret_cast<u32*>(smem + gid * sym_escw_group_lds<L>());
    const EscEnds<L> ends{reinterpret_cast<uint2*>(s_off + L)};
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
        ends.build(g, nonempty, incl);
        u32 col[PER];
#pragma unroll
        for (u32 u = 0; u < PER; ++u) {
            const u32 p = u * L + gl;
            col[u] = kEscInvalid;
            if (p < total) col[u] = src.b_col[s_off[ends.owner(p)] + p];
        }
        esc_sort_wide<L>(col, gl);
        const u32 prev_col = dpp_move<kDppWaveShr1>(kEscInvalid, col[PER - 1]);
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

}  // namespace speck
