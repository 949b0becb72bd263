// esc.hpp -- expand / sort / compress in registers for rows with at most 32 products (8 lanes x 4 products).
//
// The hash classes pay a fixed price per row -- clear a table, stage the A row, insert with compare-and-swap,
// compact, rank-sort, emit: ~410 of the ~600 VALU wave-instructions an 8-row wave-iteration of the small-row
// launch issues, and that launch is bound by VALU issue (70 % of all issue slots on the mac_econ stand-in,
// DESIGN.md 6).  A row whose products fit the registers of its lane group needs none of it:
//   expand    lane <-> product (4 per lane), owner of a product from a bit mask of where the entries' products end
//   sort      the products as packed keys (column << 5 | product number: columns are < 2^27, Multiply.cu:57-66)
//             with a bitonic network over 8 lanes x 4 registers -- min / max only, DPP moves inside the group
//   compress  equal columns are neighbours: segmented sum in sorted order (deterministic: by product number),
//             the last element of every run carries the entry, its rank is the number of runs before it
// No LDS atomics, no probing loops, no table.  The values ride in LDS by product number (32 x 8 B per group).
// (Role of the reference's multi-row hash blocks for tiny rows, include/GPU/spECK_HashSpGEMM.cuh:591-738.)
#pragma once
#include "device_common.hpp"

namespace speck {

// Two widths: 8 lanes (32 products, key = column << 5 | product) and 16 lanes (64 products, key = column << 6 |
// product -- columns must then be < 2^26: ClassifyParams::esc16).
constexpr u32 kEscLanes = 8, kEscPerLane = 4, kEscProducts = kEscLanes * kEscPerLane;  // 32
constexpr u32 kEscInvalid = 0xFFFFFFFFu;
constexpr int kDppRowMirror = 0x140;
constexpr int kDppRowShl4 = 0x104, kDppRowShr4 = 0x114;

constexpr int kDppQuadXor1 = 0xB1;   // quad_perm [1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;   // quad_perm [2,3,0,1]
constexpr int kDppQuadMirror = 0x1B; // quad_perm [3,2,1,0]
constexpr int kDppRowHalfMirror = 0x141;

// a DPP move whose every lane has a source (quad permutations, mirrors, rotations): no `old` operand -- update_dpp with
// one costs a v_mov per move to set it up (a third of the VALU instructions of the sorting networks)
template <int CTRL>
__device__ __forceinline__ u32 dpp_fetch(u32 v)
{
    return (u32)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}

// OR over the 8 lanes of a group, result in every lane
__device__ __forceinline__ u32 esc_group_or(u32 v)
{
    v |= dpp_fetch<kDppQuadXor1>(v);
    v |= dpp_fetch<kDppQuadXor2>(v);
    v |= dpp_fetch<kDppRowHalfMirror>(v);
    return v;
}

// OR over the 16 lanes of a DPP row
__device__ __forceinline__ u32 esc_row_or(u32 v)
{
    v = esc_group_or(v);
    v |= dpp_fetch<kDppRowMirror>(v);
    return v;
}
// the value of lane ^ 4 (two bank-masked row shifts: quads 0 / 2 read the quad above, quads 1 / 3 the quad below)
__device__ __forceinline__ u32 esc_xor4(u32 v)
{
    u32 t = (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, kDppRowShl4, 0xF, 0x5, false);
    t = (u32)__builtin_amdgcn_update_dpp((int)t, (int)v, kDppRowShr4, 0xF, 0xA, false);
    return t;
}

// median of three unsigned values in ONE instruction.  With k = 0 it is min(x, p), with k = 0xFFFFFFFF max(x, p): a
// compare-exchange whose direction differs from lane to lane is mov_dpp + v_med3_u32 instead of min + max + select
// (the launches of the register classes are bound by VALU issue since they took the rows of the hash classes, round 4).
__device__ __forceinline__ u32 esc_med3(u32 x, u32 p, u32 k)
{
    u32 r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(p), "v"(k));
    return r;
}
// k of a lane for the compare-exchanges that pair lanes across bit `b` of the lane number: the lower lane of a pair
// (bit clear) keeps the minimum
__device__ __forceinline__ u32 esc_dir(u32 gl, u32 b) { return 0u - ((gl >> b) & 1u); }

// compare-exchange with the same register of a partner lane: k = 0 keeps the minimum, k = ~0 the maximum
template <int CTRL>
__device__ __forceinline__ void esc_cx_lane(u32& x, u32 src, u32 k)
{
    x = esc_med3(x, dpp_fetch<CTRL>(src), k);
}
__device__ __forceinline__ void esc_cx(u32& lo, u32& hi)
{
    const u32 a = min(lo, hi), b = max(lo, hi);
    lo = a;
    hi = b;
}

// Ascending bitonic sort of the 32 elements of an 8-lane group, element i = lane * 4 + register ("flip" form:
// every compare-exchange gives the lower index the minimum).  `gl` = lane inside the group.
__device__ __forceinline__ void esc_sort32(u32 (&x)[4], u32 gl)
{
    const u32 l0 = esc_dir(gl, 0), l1 = esc_dir(gl, 1), l2 = esc_dir(gl, 2);
    // k = 1, 2: inside the lane
    esc_cx(x[0], x[1]);
    esc_cx(x[2], x[3]);
    esc_cx(x[0], x[3]);
    esc_cx(x[1], x[2]);
    esc_cx(x[0], x[1]);
    esc_cx(x[2], x[3]);
    // k = 3: flip over 8 elements = partner lane ^ 1, register 3 - r; then distances 2, 1 inside the lane
    {
        const u32 y0 = x[0], y1 = x[1], y2 = x[2], y3 = x[3];
        esc_cx_lane<kDppQuadXor1>(x[0], y3, l0);
        esc_cx_lane<kDppQuadXor1>(x[1], y2, l0);
        esc_cx_lane<kDppQuadXor1>(x[2], y1, l0);
        esc_cx_lane<kDppQuadXor1>(x[3], y0, l0);
    }
    esc_cx(x[0], x[2]);
    esc_cx(x[1], x[3]);
    esc_cx(x[0], x[1]);
    esc_cx(x[2], x[3]);
    // k = 4: flip over 16 = lane ^ 3, register 3 - r; distance 4 = lane ^ 1; then 2, 1
    {
        const u32 y0 = x[0], y1 = x[1], y2 = x[2], y3 = x[3];
        esc_cx_lane<kDppQuadMirror>(x[0], y3, l1);
        esc_cx_lane<kDppQuadMirror>(x[1], y2, l1);
        esc_cx_lane<kDppQuadMirror>(x[2], y1, l1);
        esc_cx_lane<kDppQuadMirror>(x[3], y0, l1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) esc_cx_lane<kDppQuadXor1>(x[r], x[r], l0);
    esc_cx(x[0], x[2]);
    esc_cx(x[1], x[3]);
    esc_cx(x[0], x[1]);
    esc_cx(x[2], x[3]);
    // k = 5: flip over 32 = lane ^ 7, register 3 - r; distance 8 = lane ^ 2; 4 = lane ^ 1; then 2, 1
    {
        const u32 y0 = x[0], y1 = x[1], y2 = x[2], y3 = x[3];
        esc_cx_lane<kDppRowHalfMirror>(x[0], y3, l2);
        esc_cx_lane<kDppRowHalfMirror>(x[1], y2, l2);
        esc_cx_lane<kDppRowHalfMirror>(x[2], y1, l2);
        esc_cx_lane<kDppRowHalfMirror>(x[3], y0, l2);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) esc_cx_lane<kDppQuadXor2>(x[r], x[r], l1);
#pragma unroll
    for (int r = 0; r < 4; ++r) esc_cx_lane<kDppQuadXor1>(x[r], x[r], l0);
    esc_cx(x[0], x[2]);
    esc_cx(x[1], x[3]);
    esc_cx(x[0], x[1]);
    esc_cx(x[2], x[3]);
}

// ... and of the 64 elements of a 16-lane group (one DPP row)
__device__ __forceinline__ void esc_sort64(u32 (&x)[4], u32 gl)
{
    const u32 l0 = esc_dir(gl, 0), l1 = esc_dir(gl, 1), l2 = esc_dir(gl, 2), l3 = esc_dir(gl, 3);
    esc_sort32(x, gl);  // every half of 8 lanes ascending (the k = 1 .. 5 stages never leave the half)
    // k = 6: flip over 64 = lane ^ 15, register 3 - r; distances 32 .. 4 = lane ^ 4, ^ 2, ^ 1; then 2, 1 inside the lane
    {
        const u32 y0 = x[0], y1 = x[1], y2 = x[2], y3 = x[3];
        esc_cx_lane<kDppRowMirror>(x[0], y3, l3);
        esc_cx_lane<kDppRowMirror>(x[1], y2, l3);
        esc_cx_lane<kDppRowMirror>(x[2], y1, l3);
        esc_cx_lane<kDppRowMirror>(x[3], y0, l3);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = esc_med3(x[r], esc_xor4(x[r]), l2);
#pragma unroll
    for (int r = 0; r < 4; ++r) esc_cx_lane<kDppQuadXor2>(x[r], x[r], l1);
#pragma unroll
    for (int r = 0; r < 4; ++r) esc_cx_lane<kDppQuadXor1>(x[r], x[r], l0);
    esc_cx(x[0], x[2]);
    esc_cx(x[1], x[3]);
    esc_cx(x[0], x[1]);
    esc_cx(x[2], x[3]);
}
template <u32 L>
__device__ __forceinline__ void esc_sort(u32 (&x)[4], u32 gl)
{
    if constexpr (L == 8) esc_sort32(x, gl);
    else esc_sort64(x, gl);
}

// move a double one lane up / down inside the 16-lane DPP row (lanes without a source get `fill`)
template <int CTRL>
__device__ __forceinline__ double dpp_move_f64(double fill, double v)
{
    const u64 f = __double_as_longlong(fill), x = __double_as_longlong(v);
    const u32 lo = dpp_move<CTRL>((u32)f, (u32)x), hi = dpp_move<CTRL>((u32)(f >> 32), (u32)(x >> 32));
    return __longlong_as_double((u64(hi) << 32) | lo);
}

}  // namespace speck
