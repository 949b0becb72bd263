// chain3.hpp -- exclusive prefix of ONE 48-bit quantity over up to 2^24 workgroups of one launch: the look-back chain of
// chain.hpp with 8-byte descriptors and THREE levels of sums above the workgroups.
//
// chain.hpp carries a 128-byte descriptor (class counts, two 64-bit quantities, maxima) over at most 4096 workgroups --
// what the integer stages need.  The one-walk kernel of the hash classes (numeric.hip, walk_hash_kernel) has a workgroup
// per EIGHT rows and needs one number per workgroup, the entries of C before it: 10^6 workgroups on the nlpkkt stand-in.
// Same construction: a word is {tag : 16 | payload : 48}, written and read with relaxed AGENT-scope atomics (valid iff
// its tag is the launch's -- no flag, no fence, no clearing between launches), a workgroup waits only for workgroups with a
// lower index (dispatched before it), and the sum of a FULL block of 64^k workgroups is published by its last workgroup:
//   l0[g]                 nnz of workgroup g
//   lk[g / 64^k]          sum of a full block of 64^k workgroups, k = 1, 2, 3
//   prefix(g) = sum over k of  lk[ 64 * (g / 64^(k+1)) .. g / 64^k )        -- at most 63 words per level
// Four waves read the four sets side by side -- a lane per word, REQUESTED early (chain3_begin) and waited for late
// (chain3_finish), every late word re-requested in parallel -- so a prefix costs at most one trip to memory, and none when
// the caller has something to do in between.
// (Role: the reference places rows with cub::DeviceScan::ExclusiveSum in a kernel of its own, source/GPU/Multiply.cu:570.)
#pragma once
#include "device_common.hpp"

namespace speck {

constexpr u32 kChain3Fan = 64, kChain3Levels = 4;
constexpr u32 kChain3MaxGroups = 1u << 24;  // 64^4

struct Chain3 {
    u64* l[kChain3Levels];  // l[0]: per workgroup; l[k]: per block of 64^k
    u32* error;             // != 0: a wait timed out
    u32 tag;                // 1 .. 65535 from the host, one per launch
    u32 fault;              // test hook: this workgroup never publishes; ~0: none
};
inline size_t chain3_level_words(u64 groups, u32 k)
{
    u64 n = groups;
    for (u32 i = 0; i < k; ++i) n = (n + kChain3Fan - 1) / kChain3Fan;
    return size_t(n);
}
inline size_t chain3_words(u64 groups)
{
    size_t w = 64;
    for (u32 k = 0; k < kChain3Levels; ++k) w += chain3_level_words(groups, k);
    return w;
}

#ifdef __HIPCC__
constexpr u64 kChain3Mask = (1ull << 48) - 1ull;

__device__ __forceinline__ void chain3_store(u64* p, u64 v, u32 tag)
{
    __hip_atomic_store(p, (u64(tag) << 48) | (v & kChain3Mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// words set[0 .. n) by ONE wave (n <= 64 * PER), a lane per word: chain3_issue requests them ONCE and returns at once
// (whatever the caller does next hides the trip); chain3_sum polls the ones that were late and returns the sum of the
// payloads in every lane
template <u32 PER = 1>
struct Chain3Words {
    u64 w[PER];
};
template <u32 PER = 1>
__device__ __forceinline__ Chain3Words<PER> chain3_issue(const u64* set, u32 n, u32 tag)
{
    const u32 lane = lane_id();
    Chain3Words<PER> r;
#pragma unroll
    for (u32 i = 0; i < PER; ++i)
        r.w[i] = (i * 64u + lane < n) ? __hip_atomic_load(set + i * 64u + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                      : (u64(tag) << 48);
    return r;
}
template <u32 PER = 1>
__device__ __forceinline__ u64 chain3_sum(const u64* set, Chain3Words<PER> r, u32 tag, bool* timed_out)
{
    const u32 lane = lane_id();
    u32 rounds = 0;
    while (true) {
        bool late = false;
#pragma unroll
        for (u32 i = 0; i < PER; ++i) late |= (u32)(r.w[i] >> 48) != tag;
        if (__ballot(late) == 0) break;
        if (++rounds > (1u << 20)) {  // (~1 s: the workgroups before this one are not coming)
            *timed_out = true;
            break;
        }
        __builtin_amdgcn_s_sleep(2);
#pragma unroll
        for (u32 i = 0; i < PER; ++i)
            if ((u32)(r.w[i] >> 48) != tag) r.w[i] = __hip_atomic_load(set + i * 64u + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    u64 s = 0;
#pragma unroll
    for (u32 i = 0; i < PER; ++i) s += ((u32)(r.w[i] >> 48) == tag) ? (r.w[i] & kChain3Mask) : 0ull;
    return wave_reduce_add(s);
}

// The look-back of workgroup g in two halves.  chain3_begin: publish `mine` and REQUEST the words of the four levels (wave k
// level k); chain3_finish: wait for what was late, publish the sums this workgroup closes, return the sum over the
// workgroups [0, g).  All threads of a workgroup of 256 threads call both; s_tmp: 12 u64 of LDS, s_tmp[8] zeroed by the
// caller with a barrier behind it, not reused while a thread may still be inside chain3_finish.
// *ok = false (to every thread) if a wait timed out: the prefix is truncated, the caller places nothing (Chain3::error is
// raised first, with a returning atomic: whoever builds on a sum published afterwards -- in particular the last workgroup,
// which reports -- finds it set).
__device__ __forceinline__ void chain3_span(u32 g, u32 k, u32& first, u32& idx)
{
    idx = g;
    for (u32 i = 0; i < k; ++i) idx /= kChain3Fan;
    first = k + 1u < kChain3Levels ? (idx / kChain3Fan) * kChain3Fan : 0u;  // (the top level holds at most 64 words)
}
__device__ __forceinline__ Chain3Words<1> chain3_begin(const Chain3& ch, u32 g, u64 mine)
{
    const u32 t = threadIdx.x, wid = t >> 6;
    if (t == 0 && g != ch.fault) chain3_store(ch.l[0] + g, mine, ch.tag);
    u32 first, idx;
    chain3_span(g, wid, first, idx);
    return chain3_issue<1>(ch.l[wid] + first, idx - first, ch.tag);
}
__device__ __forceinline__ u64 chain3_finish(const Chain3& ch, u32 g, u32 ngroups, u64 mine, const Chain3Words<1>& pend, u64* s_tmp,
                                             bool* ok)
{
    const u32 t = threadIdx.x, wid = t >> 6, lane = lane_id();
    const u32 tag = ch.tag;
    // (s_tmp[8] = 0 by the CALLER, a barrier behind it: one barrier here instead of three -- a workgroup of the one-walk
    //  kernel makes this call once, and every barrier is a wait for its slowest wave)
    bool timed_out = false;
    u32 first, idx;
    chain3_span(g, wid, first, idx);
    const u64 sk = chain3_sum<1>(ch.l[wid] + first, pend, tag, &timed_out);
    if (__ballot(timed_out) != 0 && lane == 0) {
        const u32 was = __hip_atomic_fetch_or(ch.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::"v"(was));
        s_tmp[8] = 1;
    }
    // the last workgroup of a FULL block of 64 publishes the block's sum at once, from the wave that has it (one
    // workgroup in 64 closes such a block; the higher levels -- one in 4096 ... -- go out behind the barrier)
    if (wid == 0 && lane == 0 && (g % kChain3Fan) == kChain3Fan - 1u && g + 1u < ngroups) chain3_store(ch.l[1] + g / kChain3Fan, sk + mine, tag);
    if (lane == 0) s_tmp[wid] = sk;
    __syncthreads();
    if (t == 0 && g + 1u < ngroups) {
        u64 run = s_tmp[0] + mine;
        u32 span = kChain3Fan;
        for (u32 k = 2; k < kChain3Levels; ++k) {
            run += s_tmp[k - 1];
            span *= kChain3Fan;
            if ((g % span) == span - 1u) chain3_store(ch.l[k] + g / span, run, tag);
        }
    }
    *ok = s_tmp[8] == 0;
    return s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];  // (s_tmp is not written again by this call)
}
#endif

}  // namespace speck
