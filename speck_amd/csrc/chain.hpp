// chain.hpp -- exclusive prefix over the workgroups of ONE kernel launch, without a second kernel and without a fence.
//
// The integer stages need, per workgroup, what the workgroups before it found (rows per kernel class, nnz, scratch
// entries) before they can place their own rows in the class lists / in C.row_offsets.  Rounds 1-4 did that with a
// kernel boundary: every block left partials, every block of a SECOND kernel folded all of them (sym_scatter_kernel,
// num_count_kernel + num_apply_kernel: 10 + 18 us of a 145-us multiply on the scircuit stand-in).  Here a workgroup
// PUBLISHES its aggregate and LOOKS BACK at its predecessors inside the same launch (role of the single-pass scan the
// reference takes from cub::DeviceScan, source/GPU/Multiply.cu:570, and of its load balancer's tile tree,
// include/GPU/scan_largearray_kernel.cuh:182-281):
//   * a descriptor is a handful of 64-bit words, each {tag : 16 | payload : 48}, written and read with RELAXED AGENT-SCOPE
//     ATOMICS (they bypass the XCD's non-coherent L2) -- a word is valid iff its tag is the launch's, so no flag,
//     no release / acquire fence (a device-scope fence writes the whole L2 back on this multi-XCD part: the grid-barrier
//     forms of rounds 2 and 4 lost 45-80 us to it) and no clearing between launches;
//   * two levels: the last workgroup of every SUPER-block of kChainSuper workgroups sums its super-block's aggregates
//     and publishes them; a workgroup then needs the aggregates of its own super-block before it (<= 63) and the sums
//     of the super-blocks before (<= 64): one or two dependent round trips, whatever the grid;
//   * a workgroup only ever waits for workgroups with a LOWER index (in-order dispatch makes that deadlock-free; a
//     wait that does not end in ~1 s raises Chain::error -- agent-scope, BEFORE the workgroup publishes anything made
//     from the truncated prefix -- the workgroup places nothing, the last workgroup reports it together with
//     capacity_miss (so that the kernels queued behind walk nothing), the host returns SPECK_ERR_HIP and clears the flag);
//   * the tag comes from the host with every launch (1 .. 65535 in turn; the host clears the buffers, one memset on the
//     stream, before a tag comes round again: no stale word can carry the tag of a later launch).
#pragma once
#include "device_common.hpp"

namespace speck {

constexpr u32 kChainWords = 24;     // words of a workgroup's aggregate as the callers see them (u32 each)
constexpr u32 kChainPfxWords = 16;  // words [0, 16): every workgroup needs their PREFIX; [16, 24): only the totals matter
constexpr u32 kChainSumWords = 20;  // words [0, 20) are summed, [20, 24) are maxed
constexpr u32 kChainSuper = 64;     // workgroups per super-block
constexpr u32 kChainMaxBlocks = kChainSuper * kChainSuper;  // 4096: a lane per descriptor in either level

// Word layout shared by the two producers (analysis: symbolic phase; scan: numeric phase)
constexpr u32 kCwClass = 0;    // [0, 14): rows per class (< 2^24 per workgroup and per super-block: 131072 rows x 64)
constexpr u32 kCwPfxLo = 14;   // u64 whose prefix places things: scratch entries of the SYM_NF / SYM_GH rows | nnz
constexpr u32 kCwPfxHi = 15;
constexpr u32 kCwTotLo = 16;   // u64 of which only the total matters: products | products of the NUM_G rows
constexpr u32 kCwTotHi = 17;
constexpr u32 kCwFlags = 18;   // number of workgroups that met bad input
constexpr u32 kCwMax = 20;     // longest row (products | nnz)
constexpr u32 kCwAuxMax = 21;  // widest SYM_NF row
static_assert(SYM_CLASSES <= kCwPfxLo && NUM_CLASSES <= kCwPfxLo, "the class counts sit below the 64-bit words");

// In memory a descriptor is 16 64-bit words = ONE 128-byte line, each word {tag : 16 | payload : 48}:
//   D0 .. D6 : two class counts each (24 bits)        D7 : prefix quantity, low 48 bits
//   D8       : prefix quantity, high 16 bits | flags << 24           D9, D10 : total-only quantity (low 48, high 16)
//   D11, D12 : the two maxima                          D13 .. D15: unused (stored with the tag like the others)
// A workgroup stores its descriptor with ONE instruction (16 lanes x 8 bytes: one line, nobody else's), and a reader
// takes a descriptor with 8 lanes x 16 bytes -- a wave reads 8 descriptors per instruction, 64 in eight, every line
// exactly once.  What the first two layouts of this round cost (scripts/ubench/chain_probe.hip, 668 workgroups that all
// wait for their <= 63 predecessors): a lane per descriptor and a load per word = 63 x 24 uncoalesced transactions per
// workgroup and round of polling (+5 us for 24 words); words TRANSPOSED across workgroups = up to 64 writers of 8 bytes
// per line, which the memory side serialises (+8 us).  The hop itself takes 0.55 us (scripts/ubench/pingpong.hip).
// Every payload is read as two 24-bit fields and the fields are summed apart (a 48-bit value is lo24 + hi24 * 2^24, and
// 64 x 2^24 fits a u32): ONE uniform accumulate per lane, and a three-step reduction across the eight lane groups.
constexpr u32 kDescWords = 16;
constexpr u64 kPayMask = (1ull << 48) - 1ull;

struct Chain {
    u64* agg;    // [kChainMaxBlocks][kDescWords]
    u64* sup;    // [kChainSuper][kDescWords]
    u32* error;  // != 0: a wait timed out (never cleared by the device)
    u32 tag;     // 1 .. 65535, from the HOST: one per launch; the host clears the buffers before a tag comes round again
    u32 fault;   // test hook (option chain_fault): this workgroup never publishes -- the timeout path; ~0: none
};
constexpr size_t kChainAggWords = size_t(kChainMaxBlocks) * kDescWords, kChainSupWords = size_t(kChainSuper) * kDescWords;
constexpr size_t kChainBytes = (kChainAggWords + kChainSupWords) * 8 + 256;  // one buffer; the config holds two (launches alternate)
constexpr u32 kChainTags = 65535;

#ifdef __HIPCC__
// word d of the descriptor made from a workgroup's aggregate: `m(k)` = value k of its kChainWords values (as u64, so that
// sums can pass through; the 64-bit quantities complete in Lo + (Hi << 32)).  The values are READ WHERE THEY LIVE (LDS) by
// the lane that packs word d -- a lane-indexed local array of them went through scratch memory (round 5: a 208-byte private
// segment in both integer kernels, on the critical hop of the chain).
template <class M>
__device__ __forceinline__ u64 chain_pack(M&& m, u32 d, u32 tag)
{
    u64 v = 0;
    if (d < 7) v = (m(kCwClass + 2 * d) & 0xFFFFFFu) | ((m(kCwClass + 2 * d + 1) & 0xFFFFFFu) << 24);
    else if (d <= 10) {
        const bool is_pfx = d <= 8;
        const u64 q = is_pfx ? m(kCwPfxLo) + (m(kCwPfxHi) << 32) : m(kCwTotLo) + (m(kCwTotHi) << 32);
        if (d == 7 || d == 9) v = q & kPayMask;
        else v = (q >> 48) | (d == 8 ? (m(kCwFlags) & 0xFFFFFFu) << 24 : 0ull);
    } else if (d == 11) v = m(kCwMax);
    else if (d == 12) v = m(kCwAuxMax);
    return (u64(tag) << 48) | (v & kPayMask);
}
template <class M>
__device__ __forceinline__ void chain_publish(u64* desc, M&& m, u32 tag)  // lanes 0 .. 15 of a wave
{
    const u32 t = lane_id();
    if (t < kDescWords) __hip_atomic_store(desc + t, chain_pack(m, t, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One wave combines the first `n` (<= 64) descriptors of a set (see above).  out[k] (LDS, u64; caller's layout): sums /
// maxima over the descriptors -- the 64-bit quantities complete in their Lo word (Hi = 0).
__device__ __forceinline__ void chain_combine(const u64* set, u32 n, u32 tag, u64* out, bool* timed_out)
{
    const u32 lane = lane_id(), role = lane & 7u, grp = lane >> 3;
    u64 wa[8], wb[8];
#pragma unroll
    for (u32 i = 0; i < 8; ++i) wa[i] = wb[i] = (8u * i + grp < n) ? 0ull : (u64(tag) << 48);
    // Rounds of polling: EVERY word that does not carry the tag yet is requested again, all of a lane's requests in flight
    // together (agent scope: past the XCD's L2) -- a round costs one trip to memory however many words are late.  (Waiting
    // for the late words one after the other cost a trip EACH: workgroups publish and look back at the same moment, so the
    // first look finds almost nothing -- 4 .. 9 us per level instead of ~1.5, scripts/ubench/chain_bench.hip.)
    u32 rounds = 0;
    while (true) {
        bool late = false;
#pragma unroll
        for (u32 i = 0; i < 8; ++i) {
            const u64* p = set + size_t(8u * i + grp) * kDescWords + 2u * role;
            if ((u32)(wa[i] >> 48) != tag) wa[i] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((u32)(wb[i] >> 48) != tag) wb[i] = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (u32 i = 0; i < 8; ++i) late |= (u32)(wa[i] >> 48) != tag || (u32)(wb[i] >> 48) != tag;
        if (__ballot(late) == 0) break;
        if (++rounds > (1u << 20)) {  // (~1 s: the workgroups before this one are not coming)
            *timed_out = true;
            break;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    u32 s0 = 0, s1 = 0, s2 = 0, s3 = 0, m0 = 0, m1 = 0;
#pragma unroll
    for (u32 i = 0; i < 8; ++i) {
        const bool ok = (u32)(wa[i] >> 48) == tag && (u32)(wb[i] >> 48) == tag;  // (false only after a timeout)
        const u64 a = ok ? wa[i] & kPayMask : 0ull, b = ok ? wb[i] & kPayMask : 0ull;
        s0 += (u32)a & 0xFFFFFFu;
        s1 += (u32)(a >> 24);
        s2 += (u32)b & 0xFFFFFFu;
        s3 += (u32)(b >> 24);
        m0 = max(m0, (u32)a);
        m1 = max(m1, (u32)b);
    }
    // across the eight lane groups: lanes l, l + 8, .. hold the same words of different descriptors
#pragma unroll
    for (u32 d = 8; d < 64; d <<= 1) {
        s0 += __shfl_xor(s0, d, 64);
        s1 += __shfl_xor(s1, d, 64);
        s2 += __shfl_xor(s2, d, 64);
        s3 += __shfl_xor(s3, d, 64);
        m0 = max(m0, (u32)__shfl_xor(m0, d, 64));
        m1 = max(m1, (u32)__shfl_xor(m1, d, 64));
    }
    if (lane < 8) {
        const u64 v1 = u64(s2) + (u64(s3) << 24);  // the summed 48-bit payload of the lane's second word
        if (lane < 3) {
            out[kCwClass + 4 * lane] = s0;
            out[kCwClass + 4 * lane + 1] = s1;
            out[kCwClass + 4 * lane + 2] = s2;
            out[kCwClass + 4 * lane + 3] = s3;
        } else if (lane == 3) {
            out[kCwClass + 12] = s0;
            out[kCwClass + 13] = s1;
            out[kCwPfxLo] = v1;               // (+ the high parts << 48: lane 4, below)
        } else if (lane == 4) {
            out[kCwPfxHi] = u64(s0) << 16;    // sum of the high 16 bits, as a multiple of 2^32 (chain_u64: Lo + (Hi << 32))
            out[kCwFlags] = s1;
            out[kCwTotLo] = v1;
        } else if (lane == 5) {
            out[kCwTotHi] = u64(s0) << 16;
            out[kCwMax] = m1;
        } else if (lane == 6) {
            out[kCwAuxMax] = m0;
            out[19] = out[22] = out[23] = 0;  // (unused words of the callers' layout: summed / maxed like the others)
        }
    }
}

// A workgroup's aggregate goes out (wave 0; s_mine written, a barrier behind it) -- as early as possible: whatever the
// workgroup still has to do that needs no prefix hides the trip.
__device__ __forceinline__ void chain_publish_own(const Chain& ch, u32 b, const u32* s_mine)
{
    if (b == ch.fault) return;
    if (threadIdx.x < 64) chain_publish(ch.agg + size_t(b) * kDescWords, [&](u32 k) { return u64(s_mine[k]); }, ch.tag);
}

// a wait of the chain did not end: the flag goes out with a RETURNING agent-scope atomic -- it is performed at the
// coherence point when the value comes back, i.e. before anything this workgroup publishes afterwards
__device__ __forceinline__ void chain_raise_error(const Chain& ch)
{
    const u32 was = __hip_atomic_fetch_or(ch.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(was));
}
__device__ __forceinline__ u32 chain_error(const Chain& ch)
{
    return __hip_atomic_load(ch.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Exclusive combination over the workgroups before `b` (chain_publish_own called), for a workgroup of >= 128 threads
// (all of them call).
//   s_mine [kChainWords] (LDS, u32): this workgroup's aggregate -- written by the caller, a barrier behind it
//   s_pref [kChainWords] (LDS, u64): on return the sum / max over the workgroups [0, b); a 64-bit quantity is
//                                    chain_u64(s_pref, Lo, Hi) = Lo + (Hi << 32)
//   s_tmp  [2 * kChainWords + 2] (LDS, u64): scratch
// Returns false (to every thread) if a wait timed out: s_pref is then a truncated prefix and the caller must place NOTHING
// by it.  Chain::error is raised before this workgroup publishes a super-block sum made from such a prefix, so whoever
// builds on that sum -- in particular the LAST workgroup, which reports to the host -- finds the flag set.
__device__ __forceinline__ bool chain_exclusive(const Chain& ch, u32 b, u32 nb, const u32* s_mine, u64* s_pref, u64* s_tmp)
{
    const u32 t = threadIdx.x, wid = t >> 6;
    const u32 tag = ch.tag;
    const u32 sb = b / kChainSuper, first = sb * kChainSuper;
    bool timed_out = false;
    if (t == 0) s_tmp[2 * kChainWords] = s_tmp[2 * kChainWords + 1] = 0;  // timeout flags of wave 0 / wave 1
    __syncthreads();
    // the last workgroup of a FULL super-block publishes the super-block's sum (nobody needs the last, partial one)
    const bool closes = (b % kChainSuper) == kChainSuper - 1u && b + 1u < nb;
    // the aggregates of my super-block before me (wave 0) + the sums of the super-blocks before (wave 1)
    if (wid == 0) {
        chain_combine(ch.agg + size_t(first) * kDescWords, b - first, tag, s_tmp, &timed_out);
        if (__ballot(timed_out) != 0) {
            if (lane_id() == 0) {
                chain_raise_error(ch);
                s_tmp[2 * kChainWords] = 1;
            }
        }
        if (closes) {
            // ... published by the SAME wave, at once: the sum of a super-block must not wait for the sums of the
            // super-blocks before it (a workgroup barrier here would chain the closers one behind the other)
            wave_lds_fence();
            // (every 64-bit quantity is Lo + (Hi << 32), in s_tmp as in s_mine: chain_pack puts them together)
            chain_publish(ch.sup + size_t(sb) * kDescWords,
                          [&](u32 k) {
                              const u64 own = s_mine[k], pre = s_tmp[k];
                              return k < kChainSumWords ? pre + own : (pre > own ? pre : own);
                          },
                          tag);
        }
    } else if (wid == 1) {
        chain_combine(ch.sup, sb, tag, s_tmp + kChainWords, &timed_out);
        if (__ballot(timed_out) != 0 && lane_id() == 0) {
            chain_raise_error(ch);
            s_tmp[2 * kChainWords + 1] = 1;  // (its own word: wave 0 may be writing the other one)
        }
    }
    __syncthreads();
    if (t < kChainWords)
        s_pref[t] = t < kChainSumWords ? s_tmp[t] + s_tmp[kChainWords + t]
                                       : (s_tmp[t] > s_tmp[kChainWords + t] ? s_tmp[t] : s_tmp[kChainWords + t]);
    const bool ok = s_tmp[2 * kChainWords] == 0 && s_tmp[2 * kChainWords + 1] == 0;
    __syncthreads();
    return ok;
}
__device__ __forceinline__ u64 chain_u64(const u64* s, u32 lo, u32 hi) { return s[lo] + (s[hi] << 32); }
#endif

}  // namespace speck
