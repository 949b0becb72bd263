// esc_rows.hpp -- the row loop of the register-resident numeric classes (NUM_G8 / NUM_G16, esc.hpp), shared by
// the numeric launches (numeric.hip) and by the FUSED form inside the symbolic light launch of a replayed
// sequence (symbolic.hip): there the row is finished in the symbolic phase and written straight to the place
// the previous identical call gave it in C -- no second walk of the row in the numeric phase.
#pragma once
#include <type_traits>

#include "device_common.hpp"
#include "launch.hpp"
#include "row_groups.hpp"
#include "esc.hpp"

namespace speck {

// LDS accumulator cell.  fp32 rows accumulate in fp64 cells too: ds_add_f32 is ~10x slower than
// ds_add_f64 on gfx950 (192 vs 20.6 cycles per wave instruction, scripts/ubench/lds_atomics.hip).
// The product a*b is still rounded to T first (as the reference does); the sum is rounded to T
// once, when the row is written.
template <typename T>
using Acc = double;

// ------------------------------------------------------------------ NUM_G8 / NUM_G16: expand / sort / compress (esc.hpp)
// Rows with at most 4 L products and L entries of A (L = 8 or 16 lanes per row): 4 products per lane, nothing but
// registers, DPP and 4 L values in LDS.  LDS per group: the products by number | a_ik and B-row offsets of the
// (non-empty) entries.
template <typename T, u32 L>
constexpr u32 num_esc_group_lds()
{
    return 4u * L * (u32)sizeof(Acc<T>) + L * (4u + (u32)sizeof(Acc<T>));
}

// One row of the class by its lane group: expand / sort / compress, the finished row (ascending columns, one entry per
// distinct column) to out_col / out_val[0 .. nnz) -- if nnz <= room (else nothing is stored).  Returns the row's nnz.
// `mine`: the group's num_esc_group_lds<T, L>() bytes of LDS.  An idle group passes a0 == a1 (nothing is read or stored);
// every lane of the wave must call (wave-wide ballots and fences inside).
template <typename T, u32 L>
__device__ __forceinline__ u32 num_esc_row(const SubWave<L>& g, unsigned char* mine, const ProductSrc<T>& src, u32 a0, u32 a1,
                                           u32* __restrict__ out_col, T* __restrict__ out_val, u32 room)
{
    static_assert(L == 8 || L == 16, "8 or 16 lanes per row");
    using Mask = typename std::conditional<L <= 8, u32, u64>::type;
    constexpr u32 PER = kEscPerLane, NP = PER * L, TAG = L == 8 ? 5u : 6u;
    Acc<T>* s_vals = reinterpret_cast<Acc<T>*>(mine);     // [4 L] products by number
    Acc<T>* s_av = s_vals + NP;                           // [L]   a_ik of the j-th non-empty entry
    u32* s_off = reinterpret_cast<u32*>(s_av + L);        // [L]   its B-row start minus its first product number
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
    // bit k: a (non-empty) entry's products end at k -- the owner of product p is the number of set bits <= p
    Mask ends;
    if constexpr (L == 8) {
        ends = esc_group_or((nonempty && incl < 32u) ? (1u << incl) : 0u);
    } else {
        const u64 bit = (nonempty && incl < 64u) ? (1ull << incl) : 0ull;
        ends = (u64(esc_row_or((u32)(bit >> 32))) << 32) | esc_row_or((u32)bit);
    }
    wave_lds_fence();
    u32 key[PER];
#pragma unroll
    for (u32 u = 0; u < PER; ++u) {
        const u32 p = u * L + gl;
        key[u] = kEscInvalid;
        if (p < total) {
            u32 j;
            if constexpr (L <= 8) j = (u32)__popc(ends & ((2u << p) - 1u));
            else j = (u32)__popcll(ends & ((2ull << p) - 1ull));
            const u32 ib = s_off[j] + p;
            const u32 c = src.b_col[ib];
            const T bv = src.b_val[ib];
            const T prod = (T)s_av[j] * bv;  // rounded product, added later (no FMA across the add)
            s_vals[p] = (Acc<T>)prod;
            key[u] = (c << TAG) | p;
        }
    }
    wave_lds_fence();
    // ---- sort by (column, product number)
    esc_sort<L>(key, gl);
    // ---- compress: sums of the runs of equal columns, in sorted order
    // (validity by POSITION, not by the sentinel: the `total` valid keys sort to the front, and the packed key of
    //  the last product of a full row that ends in column 2^27 - 1 (2^26 - 1 for 16 lanes) IS 0xFFFFFFFF --
    //  check_inputs admits cols(B) == 2^27, Multiply.cu:57-66)
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
    const u32 prev_col = dpp_move<kDppRowShr + 1>(kEscInvalid, col[PER - 1]);
    const bool cont = gl != 0 && col[0] != kEscInvalid && prev_col == col[0];
    Acc<T> chain = sum[PER - 1];             // running sum of the run that ends this lane
    bool stop = !(lead[PER - 1] && cont);    // ... which does not reach back into the lane before
#pragma unroll
    for (u32 d = 1; d < L; d <<= 1) {
        // (the moves first, for ALL lanes: a DPP read from a lane a branch has switched off returns the fill value)
        Acc<T> t;
        bool ts;
        if (d == 1) {
            t = dpp_move_f64<kDppRowShr + 1>(0.0, chain);
            ts = dpp_move<kDppRowShr + 1>(1u, (u32)stop) != 0;
        } else if (d == 2) {
            t = dpp_move_f64<kDppRowShr + 2>(0.0, chain);
            ts = dpp_move<kDppRowShr + 2>(1u, (u32)stop) != 0;
        } else if (d == 4) {
            t = dpp_move_f64<kDppRowShr + 4>(0.0, chain);
            ts = dpp_move<kDppRowShr + 4>(1u, (u32)stop) != 0;
        } else {
            t = dpp_move_f64<kDppRowShr + 8>(0.0, chain);
            ts = dpp_move<kDppRowShr + 8>(1u, (u32)stop) != 0;
        }
        const bool in_group = gl >= d;  // (the DPP row is 16 lanes: two groups of 8)
        if (!stop && in_group) chain += t;
        stop = stop || !in_group || ts;
    }
    const Acc<T> from_prev = dpp_move_f64<kDppRowShr + 1>(0.0, chain);
    const Acc<T> carry = cont ? from_prev : Acc<T>(0);
#pragma unroll
    for (u32 r = 0; r < PER; ++r) sum[r] += lead[r] ? carry : Acc<T>(0);
    // the last element of a run carries the entry; its rank = runs that end before it
    const u32 next_col = dpp_move<kDppRowShl + 1>(kEscInvalid, col[0]);
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
            out_col[pos] = col[r];
            out_val[pos] = (T)sum[r];
            ++pos;
        }
    wave_lds_fence();  // the next row overwrites the staging and the products
    return all;
}

// FUSED (replayed sequence, symbolic phase): `cls` is the SYMBOLIC class whose list is walked; the row's nnz goes
// to `counts`, the row itself to C at w.nf_pred_off[row] -- the offset the previous identical call gave it -- if its
// fresh nnz fits the place that call made for it (and only a row of exactly that nnz keeps the sequence alive: else the
// flag -- the scan compares EVERY fresh offset with the prediction, and the eager path re-runs the multiply).
template <typename T, u32 L, int THREADS, bool FUSED = false>
__device__ __forceinline__ void num_esc_body(unsigned char* smem, const ProductSrc<T>& src, const RowWork& w,
                                             u32* __restrict__ c_col, T* __restrict__ c_val, int cls, u32 bidx,
                                             u32 nblk, u32 hint = kNoCount, u32* __restrict__ counts = nullptr)
{
    using G = SubWave<L>;
    constexpr u32 NG = THREADS / L;
    const G g;
    const u32 gid = threadIdx.x / L;
    unsigned char* mine = smem + gid * num_esc_group_lds<T, L>();
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
        const u32 all = num_esc_row<T, L>(g, mine, src, rec.a0, rec.a1, c_col + place, c_val + place, room);
        if constexpr (FUSED) {
            if (g.lane == 0) {
                counts[rec.row] = all;
                if (all != room) const_cast<DeviceStats*>(w.st)->capacity_miss = 1;
            }
        }
    }
}

}  // namespace speck
