// numeric.hip -- numeric phase for gfx950: accumulate the products of every C row, then
// write the row's column ids ascending with their values.
// Role of the reference's spGEMMNumericLauncher / denseSpGEMMNumeric / hashSpGEMMSortingKernel
// (include/GPU/spECK_HashSpGEMM.cuh:1714-1794, 1439-1472, 1856-1925) and HashMap
// (include/HashMap.cuh:23-110).  Designed for wave64 + 160 KiB LDS:
//   NUM_DIRECT : A row with one entry -> scaled copy of a (sorted) B row, 16 lanes per row
//   NUM_G8     : 8 lanes per row (8 rows per wave), 32-entry table, rank sort (bitmap rank for narrow rows)
//   NUM_G16    : 16 lanes per row (4 rows per wave), 64-entry table, rank sort
//   NUM_W128   : 32 lanes per row (2 rows per wave), 128-entry table, ballot compaction + rank sort
//   NUM_W256   : 32 lanes per row, 256-entry table, two-level bitmap sort
//   NUM_W512   : one wave per row, 512-entry table, two-level bitmap sort
//   NUM_B2K/B8K: one workgroup per row, 2048/8192-entry table, two-level bitmap sort
//   NUM_D1/D2  : dense column-window accumulator (value per column + presence bitmap):
//                one ds_add_f64 per product + one ds_or per run of neighbouring lanes, no probing, output
//                sorted for free; also the numeric-first rows' kernel (nf_dense_kernel, window sized at run time)
//   NUM_G      : bucketed global-memory spill for heavy, wide rows (seven kernels)
// Two-level bitmap sort: the keys of a row are DISTINCT, so their sorted position is a prefix
// popcount.  Level 1 marks the occupied 32-column buckets (range/32 bits), its prefix ranks
// the occupied buckets; level 2 holds one 32-bit mask per OCCUPIED bucket (<= nnz words).
// O(nnz + range/1024) LDS operations instead of a comparison sort (the reference uses an
// O(nnz^2) rank sort below 500 entries and cub::BlockRadixSort above, :813-865, 1856-1925).
// The sort scratch aliases the hash table: by then every lane holds its slots in registers.
// The product a*b is rounded first and then added with an LDS atomic (ds_add_f64), as the
// reference does (spECK_HashSpGEMM.cuh:157-165) -- no FMA across the add.
// Algorithmic bytes per row: 8 + 20*lenA + 12*ops + 4 + 12*nnz for fp64 (device_common.hpp).
#include <algorithm>
#include <type_traits>

#include "chain3.hpp"
#include "device_common.hpp"
#include "launch.hpp"
#include "row_groups.hpp"
#include "esc_rows.hpp"
#include "esc_wide.hpp"

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
                                                u32 nblk, u32 hint = kNoCount)
{
    using G = Block<THREADS>;
    const G g;
    T* m_av = reinterpret_cast<T*>(smem);
    u32* m_incl = reinterpret_cast<u32*>(m_av + THREADS);
    u32* m_src = m_incl + THREADS;
    u32* m_dst = m_src + THREADS;
    u32* scratch = m_dst + THREADS;
    u32* win_all = scratch + THREADS / 64 + 2;
    const u32 l = lane_id();
    u32* win = win_all + (threadIdx.x >> 6) * kWinWords;
    // (a thread per row of the chunk: with a host-known count the records of the first chunk are requested at once, next
    //  to the device-side class table that confirms the count)
    const u32 miss = w.st->capacity_miss;
    const u32 count = min(w.st->num.count[NUM_DIRECT], w.m);
    const bool hinted = hint != kNoCount && hint <= w.m;
    RowSlice rs{0u, 0u, 1u};
    RowRec first_rec{};
    if (hinted) {
        rs = row_slice(hint, bidx, nblk, THREADS, 0u, (w.xcd_aware & 1u) != 0);
        if (rs.idx + threadIdx.x < rs.end) first_rec = *class_rec_at(w.recs, w.m, NUM_DIRECT, rs.idx + threadIdx.x);
    }
    if (block_void(miss)) return;
    const bool spec_ok = hinted && count == hint;
    if (!spec_ok) rs = row_slice(count, bidx, nblk, THREADS, 0u, (w.xcd_aware & 1u) != 0);
    for (u32 first = rs.idx; first < rs.end; first += rs.stride) {
        const u32 cnt = min((u32)THREADS, rs.end - first);
        u32 len = 0, bs = 0, base = 0;
        T av = T(0);
        if (threadIdx.x < cnt) {
            const RowRec rec = (spec_ok && first == rs.idx) ? first_rec : *class_rec_at(w.recs, w.m, NUM_DIRECT, first + threadIdx.x);
            len = rec.nnz;
            base = rec.base;
            av = src.a_val[rec.a0];
            bs = src.b_sl[rec.a0].x;
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

// (Acc<T>, the LDS accumulator cell: esc_rows.hpp)

// ------------------------------------------------------------------ sorting back-ends
// Rank sort for tiny tables.  The occupied slots are first COMPACTED (ballots) into `ckeys` with their
// slot numbers in `cslot`, then every lane ranks ceil(nnz / SIZE) <= EMAX compacted keys against all of
// them -- not its CAP / SIZE table slots, a third to a half of which are empty at a load <= 2/3: these
// kernels are bound by VALU issue (80 % busy, profiles/r02_pmc_mac_econ_tiny.csv), and the compare
// loop is their largest single part.  `ckeys` may alias the keys of the table (all slots are in
// registers before the first write); the values stay where they are and are fetched by slot number.
// `cap_row` (a power of two, SIZE <= cap_row <= CAP) slots of the table are in use.
// VERIFY (all three emitters; RowWork::verify_numeric): the row's room in C is what the previous identical call found
// (`room`), not what a symbolic pass of this call counted -- nothing is stored at or beyond it, and the number of entries
// the table holds NOW is returned for the caller to compare.
template <class G, typename T, u32 CAP, u32 NMAX, bool VERIFY = false>
__device__ __forceinline__ u32 emit_rank_sorted(const G& g, const u32* keys, const Acc<T>* vals,
                                                u32* ckeys, u8* cslot, u32 cap_row, u32 base,
                                                u32* __restrict__ c_col, T* __restrict__ c_val, u32 room = 0xFFFFFFFFu)
{
    constexpr u32 OWN = CAP / G::SIZE;
    constexpr u32 EMAX = (NMAX + G::SIZE - 1) / G::SIZE;
    static_assert(NMAX + 4 <= CAP, "the compacted keys (padded to a uint4) alias the table's keys");
    u32 k[OWN];
#pragma unroll
    for (u32 j = 0; j < OWN; ++j) {
        k[j] = kEmptyKey;
        if (j * G::SIZE < cap_row) k[j] = keys[j * G::SIZE + g.lane];
    }
    g.sync();
    u32 run = 0;
    const u64 lt = (1ull << g.lane) - 1ull;
#pragma unroll
    for (u32 j = 0; j < OWN; ++j) {
        const u64 mask = g.ballot(k[j] != kEmptyKey);
        if (k[j] != kEmptyKey) {
            const u32 pos = run + __popcll(mask & lt);
            if (!VERIFY || pos < NMAX) {  // (more entries than the class holds: the replay is rejected anyway)
                ckeys[pos] = k[j];
                cslot[pos] = (u8)(j * G::SIZE + g.lane);
            }
        }
        run += __popcll(mask);
    }
    const u32 found = run;
    if constexpr (VERIFY) run = min(run, NMAX);
    if (g.lane < 4) ckeys[run + g.lane] = kEmptyKey;  // pad the last uint4
    g.sync();
    u32 mk[EMAX], ms[EMAX], r[EMAX];
#pragma unroll
    for (u32 e = 0; e < EMAX; ++e) {
        const u32 idx = e * G::SIZE + g.lane;
        mk[e] = kEmptyKey;
        ms[e] = 0;
        r[e] = 0;
        if (idx < run) {
            mk[e] = ckeys[idx];
            ms[e] = cslot[idx];
        }
    }
    const uint4* ck4 = reinterpret_cast<const uint4*>(ckeys);
    // how many keys per lane the groups of this wave hold at most (uniform for the wave)
    const u32 depth = __ballot(run > 2u * G::SIZE) ? 3u : (__ballot(run > (u32)G::SIZE) ? 2u : 1u);
    if (depth == 1) {
        for (u32 q = 0; q < (run + 3) / 4; ++q) {
            const uint4 x = ck4[q];  // same address for the whole group: LDS broadcast
            r[0] += (x.x < mk[0]) + (x.y < mk[0]) + (x.z < mk[0]) + (x.w < mk[0]);
        }
    } else if (depth == 2 || EMAX < 3) {
        for (u32 q = 0; q < (run + 3) / 4; ++q) {
            const uint4 x = ck4[q];
#pragma unroll
            for (u32 e = 0; e < (EMAX < 2 ? EMAX : 2u); ++e)
                r[e] += (x.x < mk[e]) + (x.y < mk[e]) + (x.z < mk[e]) + (x.w < mk[e]);
        }
    } else {
        for (u32 q = 0; q < (run + 3) / 4; ++q) {
            const uint4 x = ck4[q];
#pragma unroll
            for (u32 e = 0; e < EMAX; ++e) r[e] += (x.x < mk[e]) + (x.y < mk[e]) + (x.z < mk[e]) + (x.w < mk[e]);
        }
    }
#pragma unroll
    for (u32 e = 0; e < EMAX; ++e)
        if (mk[e] != kEmptyKey && (!VERIFY || r[e] < room)) {
            c_col[base + r[e]] = mk[e];
            c_val[base + r[e]] = (T)vals[ms[e]];
        }
    return found;
}

// Narrow rows of the rank-sort classes: the reachable column range fits ONE bitmap word per lane of the
// group (16 lanes: 512 columns, a wave: 2048).  Rank of a key = set bits below it: one ds_or per key, one
// group scan of the word popcounts, one 8-byte LDS read + popcount per key -- ~60 wave instructions where
// the compare loop above takes 150-400.  `bm` (SIZE words) may alias the table's keys, `bp` (SIZE uint2)
// the A-row staging area.
template <class G, typename T, u32 CAP, bool VERIFY = false>
__device__ __forceinline__ u32 emit_narrow_sorted(const G& g, const u32* keys, const Acc<T>* vals, u32* bm,
                                                  uint2* bp, u32 cap_row, u32 cmin, u32 base,
                                                  u32* __restrict__ c_col, T* __restrict__ c_val, u32 room = 0xFFFFFFFFu)
{
    constexpr u32 OWN = CAP / G::SIZE;
    u32 k[OWN];
#pragma unroll
    for (u32 j = 0; j < OWN; ++j) {
        k[j] = kEmptyKey;
        if (j * G::SIZE < cap_row) k[j] = keys[j * G::SIZE + g.lane];
    }
    g.sync();
    bm[g.lane] = 0;
    g.sync();
    u32 outside = 0;  // (VERIFY: keys beyond the column range the records hold -- the structure has changed)
#pragma unroll
    for (u32 j = 0; j < OWN; ++j)
        if (k[j] != kEmptyKey) {
            const u32 d = k[j] - cmin;
            if constexpr (VERIFY) {
                if (d >= G::SIZE * 32u) {
                    k[j] = kEmptyKey;
                    outside = 1;
                    continue;
                }
            }
            atomicOr(&bm[d >> 5], 1u << (d & 31));
        }
    g.sync();
    const u32 word = bm[g.lane];
    u32 total;
    const u32 incl = g.inclusive_scan((u32)__popc(word), &total, nullptr);
    bp[g.lane] = make_uint2(word, incl - (u32)__popc(word));
    g.sync();
#pragma unroll
    for (u32 j = 0; j < OWN; ++j)
        if (k[j] != kEmptyKey) {
            const u32 d = k[j] - cmin;
            const uint2 e = bp[d >> 5];
            const u32 r = e.y + (u32)__popc(e.x & ((1u << (d & 31)) - 1u));
            if (VERIFY && r >= room) continue;
            c_col[base + r] = k[j];
            c_val[base + r] = (T)vals[j * G::SIZE + g.lane];
        }
    if constexpr (VERIFY) return g.ballot(outside != 0) ? 0xFFFFFFFFu : total;
    return total;
}

// Two-level bitmap sort (see the header comment).  S: LDS scratch of max(2*W1, 2*NMAX) words;
// it may alias the table (slots are loaded into registers first).
template <class G, typename T, u32 CAP, u32 W1, u32 NMAX, bool STAGE = G::kIsBlock, bool VERIFY = false>
__device__ __forceinline__ u32 emit_bitmap_sorted(const G& g, u32* keys, Acc<T>* vals, u32* S,
                                                   u32* scan_scratch, u32 cap_row, u32 cmin, u32 cmax,
                                                   u32 base, u32* __restrict__ c_col,
                                                   T* __restrict__ c_val, int cls = 0, u32 room = 0xFFFFFFFFu)
{
    u32* const st_keys = keys;       // staging of the sorted window (STAGE): the table's own arrays
    Acc<T>* const st_vals = vals;
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
    // both levels keep {bits, prefix} pairs: a rank lookup is one 8-byte LDS read
    uint2* l1x = reinterpret_cast<uint2*>(S);
    uint2* mx = reinterpret_cast<uint2*>(S);
    u32 emitted = 0;
    for (u64 w0 = cmin; w0 <= cmax; w0 += kWindowCols) {
        const u64 left = u64(cmax) - w0 + 1;
        const u32 ncols = left < kWindowCols ? (u32)left : (u32)kWindowCols;
        const u32 nw1 = (((ncols + 31) >> 5) + 31) >> 5;
        const u32 wbase = (u32)w0;
        for (u32 i = g.lane; i < nw1; i += G::SIZE) l1x[i].x = 0;
        g.sync();
#pragma unroll
        for (u32 j = 0; j < OWN; ++j) {
            if (j * G::SIZE >= cap_row) continue;
            const u32 d = k[j] - wbase;
            if (k[j] != kEmptyKey && d < ncols) atomicOr(&l1x[d >> 10].x, 1u << ((d >> 5) & 31));
        }
        g.sync();
        PHASE_MARK(4);
        const u32 nocc = bitmap_prefix<G, 2>(g, &l1x[0].x, &l1x[0].y, nw1, scan_scratch);
        PHASE_MARK(5);
#pragma unroll
        for (u32 j = 0; j < OWN; ++j) {
            if (j * G::SIZE >= cap_row) continue;
            const u32 d = k[j] - wbase;
            if (k[j] != kEmptyKey && d < ncols) {
                const uint2 e = l1x[d >> 10];
                brank[j] = e.y + __popc(e.x & ((1u << ((d >> 5) & 31)) - 1u));
            }
        }
        g.sync();  // level-1 arrays are dead from here: the masks alias them
        PHASE_MARK(6);
        for (u32 i = g.lane; i < nocc; i += G::SIZE) mx[i].x = 0;
        g.sync();
#pragma unroll
        for (u32 j = 0; j < OWN; ++j) {
            if (j * G::SIZE >= cap_row) continue;
            const u32 d = k[j] - wbase;
            if (k[j] != kEmptyKey && d < ncols) atomicOr(&mx[brank[j]].x, 1u << (d & 31));
        }
        g.sync();
        PHASE_MARK(7);
        const u32 total = bitmap_prefix<G, 2>(g, &mx[0].x, &mx[0].y, nocc, scan_scratch);
        PHASE_MARK(8);
        if constexpr (!STAGE) {
#pragma unroll
            for (u32 j = 0; j < OWN; ++j) {
                if (j * G::SIZE >= cap_row) continue;
                const u32 d = k[j] - wbase;
                if (k[j] != kEmptyKey && d < ncols) {
                    const uint2 e = mx[brank[j]];
                    const u32 r = emitted + e.y + __popc(e.x & ((1u << (d & 31)) - 1u));
                    if (VERIFY && r >= room) continue;
                    c_col[base + r] = k[j];
                    c_val[base + r] = (T)v[j];
                }
            }
        } else {
            // Every lane holds slots of the HASH order: their ranks are scattered over the row, and 2 x OWN store
            // instructions that each touch 64 different lines were the longest step of the sort for the workgroup
            // classes (16 slots per lane: 8.6 k of a row's 42 k cycles).  The sorted window is put together in LDS
            // first -- the table is dead (its slots are in registers, the masks are read before the barrier) -- and
            // leaves with coalesced stores.
            u32 r[OWN];
#pragma unroll
            for (u32 j = 0; j < OWN; ++j) {
                r[j] = 0xFFFFFFFFu;
                if (j * G::SIZE >= cap_row) continue;
                const u32 d = k[j] - wbase;
                if (k[j] != kEmptyKey && d < ncols) {
                    const uint2 e = mx[brank[j]];
                    r[j] = e.y + __popc(e.x & ((1u << (d & 31)) - 1u));
                }
            }
            g.sync();  // the masks are dead
#pragma unroll
            for (u32 j = 0; j < OWN; ++j)
                if (r[j] != 0xFFFFFFFFu) {
                    st_keys[r[j]] = k[j];
                    st_vals[r[j]] = v[j];
                }
            g.sync();
            const u32 fits = VERIFY ? min(total, room - min(room, emitted)) : total;
            for (u32 i = g.lane; i < fits; i += G::SIZE) {
                c_col[base + emitted + i] = st_keys[i];
                c_val[base + emitted + i] = (T)st_vals[i];
            }
        }
        emitted += total;
        g.sync();
        PHASE_MARK(9);
    }
    return emitted;
}

// The same sort in two halves, for a caller that learns WHERE the row goes only after its ranks are known (walk_hash_kernel:
// the look-back for the row's offset runs beside the sort): bitmap_rank_slots leaves every slot's key / value / rank in
// registers (rank 0xFFFFFFFF: empty slot, or a column outside [cmin, cmax]), store_ranked_slots writes them.  One sort
// window only (range <= W1 * 1024 columns); returns the number of ranked entries.
template <class G, typename T, u32 CAP, u32 W1>
__device__ __forceinline__ u32 bitmap_rank_slots(const G& g, const u32* keys, const Acc<T>* vals, u32* S, u32* scan_scratch,
                                                 u32 cap_row, u32 cmin, u32 cmax, u32 (&k)[CAP / G::SIZE],
                                                 Acc<T> (&v)[CAP / G::SIZE], u32 (&r)[CAP / G::SIZE])
{
    constexpr u32 OWN = CAP / G::SIZE;
    u32 brank[OWN];
#pragma unroll
    for (u32 j = 0; j < OWN; ++j) {
        k[j] = kEmptyKey;
        v[j] = 0;
        brank[j] = 0;
        r[j] = 0xFFFFFFFFu;
        if (j * G::SIZE < cap_row) {
            k[j] = keys[j * G::SIZE + g.lane];
            v[j] = vals[j * G::SIZE + g.lane];
        }
    }
    g.sync();
    uint2* l1x = reinterpret_cast<uint2*>(S);
    uint2* mx = reinterpret_cast<uint2*>(S);
    const u32 ncols = cmax - cmin + 1u;
    const u32 nw1 = (((ncols + 31) >> 5) + 31) >> 5;
    for (u32 i = g.lane; i < nw1; i += G::SIZE) l1x[i].x = 0;
    g.sync();
#pragma unroll
    for (u32 j = 0; j < OWN; ++j) {
        if (j * G::SIZE >= cap_row) continue;
        const u32 d = k[j] - cmin;
        if (k[j] != kEmptyKey && d < ncols) atomicOr(&l1x[d >> 10].x, 1u << ((d >> 5) & 31));
    }
    g.sync();
    const u32 nocc = bitmap_prefix<G, 2>(g, &l1x[0].x, &l1x[0].y, nw1, scan_scratch);
#pragma unroll
    for (u32 j = 0; j < OWN; ++j) {
        if (j * G::SIZE >= cap_row) continue;
        const u32 d = k[j] - cmin;
        if (k[j] != kEmptyKey && d < ncols) {
            const uint2 e = l1x[d >> 10];
            brank[j] = e.y + __popc(e.x & ((1u << ((d >> 5) & 31)) - 1u));
        }
    }
    g.sync();  // level-1 arrays are dead from here: the masks alias them
    for (u32 i = g.lane; i < nocc; i += G::SIZE) mx[i].x = 0;
    g.sync();
#pragma unroll
    for (u32 j = 0; j < OWN; ++j) {
        if (j * G::SIZE >= cap_row) continue;
        const u32 d = k[j] - cmin;
        if (k[j] != kEmptyKey && d < ncols) atomicOr(&mx[brank[j]].x, 1u << (d & 31));
    }
    g.sync();
    const u32 total = bitmap_prefix<G, 2>(g, &mx[0].x, &mx[0].y, nocc, scan_scratch);
#pragma unroll
    for (u32 j = 0; j < OWN; ++j) {
        if (j * G::SIZE >= cap_row) continue;
        const u32 d = k[j] - cmin;
        if (k[j] != kEmptyKey && d < ncols) {
            const uint2 e = mx[brank[j]];
            r[j] = e.y + __popc(e.x & ((1u << (d & 31)) - 1u));
        }
    }
    return total;
}
template <typename T, u32 OWN>
__device__ __forceinline__ void store_ranked_slots(const u32 (&k)[OWN], const Acc<T> (&v)[OWN], const u32 (&r)[OWN], u32 base,
                                                   u32 room, u32* __restrict__ c_col, T* __restrict__ c_val)
{
#pragma unroll
    for (u32 j = 0; j < OWN; ++j)
        if (r[j] < room) {
            c_col[base + r[j]] = k[j];
            c_val[base + r[j]] = (T)v[j];
        }
}

// A sequence without a symbolic pass (VERIFY) follows a completed replay of ITSELF: the sorted column ids of every row are
// still where that call left them -- in C, at the row's offset.  If the row has the same columns now, nothing has to be
// sorted and no column id has to be written: the group counts the entries of its table, and if that is the row's nnz it
// looks every column of the previous result up in the table (a read-only probe sequence: it ends at the key or at an empty
// slot -- the table holds fewer keys than slots) and stores the value at the column's place.  Same count, the previous
// columns strictly ascending and all of them found  <=>  the same set of columns: C.col_ids is right as it stands.  Anything
// else (a caller that scribbled over C, a structure that changed) is a mismatch like any other: capacity_miss, eager re-run.
// Returns the number of entries in the table, or 0xFFFFFFFF if a previous column is missing or out of order.
#ifndef SPECK_PF_MAX
#define SPECK_PF_MAX 14  // (the 8 Ki-table rows: 98 VGPRs with 14 + 14 prefetched ids, no sort registers; webbase -2.5 %)
#endif
#ifndef SPECK_EMIT_BY_PREVIOUS
#define SPECK_EMIT_BY_PREVIOUS 1
#endif
// PF > 0: the previous column ids (and their left neighbours) were requested when the row was opened -- PF per lane, in
// registers -- so that their round trip to memory runs beside the product walk instead of behind it.
template <class G, typename T, u32 CAP, u32 PF = 0>
__device__ __forceinline__ u32 emit_by_previous(const G& g, const u32* keys, const Acc<T>* vals, u32 cap_row, u32 bits,
                                                u32 base, u32 nnz, const u32* c_col_prev, T* __restrict__ c_val,
                                                u32* scratch, const u32* pf_col = nullptr, const u32* pf_left = nullptr)
{
    constexpr u32 OWN = CAP / G::SIZE;
    u32 mine = 0;
#pragma unroll
    for (u32 j = 0; j < OWN; ++j)
        if (j * G::SIZE < cap_row) mine += keys[j * G::SIZE + g.lane] != kEmptyKey ? 1u : 0u;
    const u32 cnt = g.reduce_add(mine, scratch);
    const bool go = cnt == nnz;  // (uniform for the group; both reductions stay outside the branch: DPP)
    const u32 mask = cap_row - 1u;
    u32 bad = 0;
    u32 e = 0;
    for (u32 i = g.lane; go && i < nnz; i += G::SIZE, ++e) {
        u32 col, left;
        if constexpr (PF > 0) {
            // (e is a compile-time index after unrolling: PF <= 8)
            col = 0, left = 0;
#pragma unroll
            for (u32 q = 0; q < PF; ++q)
                if (q == e) col = pf_col[q], left = pf_left[q];
        } else {
            col = c_col_prev[size_t(base) + i];
            left = i ? c_col_prev[size_t(base) + i - 1] : 0u;
        }
        if (i && left >= col) bad = 1;
        u32 slot = (col * 0x9E3779B1u) >> (32u - bits);
        u32 k = keys[slot];
        if (k != col && k != kEmptyKey) {
            const u32 step = probe_step(col, 32u - bits);
            u32 inc = kFirstProbeInc ? kFirstProbeInc : step;
            do {
                slot = (slot + inc) & mask;
                inc = step;
                k = keys[slot];
            } while (k != col && k != kEmptyKey);
        }
        if (k == col) c_val[size_t(base) + i] = (T)vals[slot];
        else bad = 1;
    }
    const u32 any_bad = g.reduce_add(bad, scratch);
    return go && any_bad ? 0xFFFFFFFFu : cnt;
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
    u32 bytes = CAP * ((u32)sizeof(Acc<T>) + 4u) + G::SIZE * (u32)sizeof(Acc<T>) + (words + 3u) / 4u * 16u;
    return bytes;
}

// Rows of the class with nnz <= NLO or nnz > NMAX are skipped: a class may be served by two
// launches with differently sized tables (NUM_B8K below).
// VERIFY: the launch of a replayed sequence that has NO symbolic pass for these rows (RowWork::verify_numeric): the record's
// nnz -- table size, room in C -- is what the previous identical call found.  The body stays inside the table (bounded
// probing) and inside the row's room whatever B holds now, counts what the table ends up with and raises capacity_miss if
// that is not the nnz it was given: the host then takes the eager path, as after any other rejected replay.
// KEEP: the row's previous column ids are kept if they are the table's keys (emit_by_previous) instead of sorted again.
// (Only in the verifying launches: measured for EVERY replayed sequence -- a symbolic pass gives the exact nnz, C holds the
//  previous result there as well -- the latency-bound light launches of the short-row inputs did not gain: scircuit
//  22.3 -> 23.1 us with the columns read at emit time, 22.2 -> 22.8 us with them requested when the row is opened.)
template <class G, typename T, u32 CAP, u32 W1, u32 NMAX, int MODE, int THREADS, u32 NLO = 0, bool VERIFY = false,
          bool KEEP = VERIFY>
__device__ __forceinline__ void num_hash_body(unsigned char* smem, const ProductSrc<T>& src, const RowWork& w,
                                              u32* __restrict__ c_col, T* __restrict__ c_val, int cls,
                                              u32 bidx, u32 nblk, u32 hint = kNoCount)
{
    constexpr u32 NG = THREADS / G::SIZE;
    constexpr u32 kGroupBytes = num_group_lds<G, T, CAP, THREADS>();
    // the sort scratch (rank: NMAX+8 words, bitmap: max(2*W1, 2*NMAX) words) fits in the table
    static_assert((MODE == SORT_RANK ? NMAX + 4 : (2 * W1 > 2 * NMAX ? 2 * W1 : 2 * NMAX)) * 4 <=
                      CAP * (sizeof(Acc<T>) + 4),
                  "sort scratch must fit in the table it aliases");
    static_assert(MODE != SORT_RANK || (NMAX <= G::SIZE * (sizeof(T) + 8) && CAP <= 256),
                  "rank sort: the slot numbers (one byte each) of the compacted keys fit in the A-row staging area");
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
    RowCursor cur = open_list<false>(w, cls, hint, bidx, nblk, NG, gid, (w.xcd_aware & (G::kIsBlock ? 4u : 1u)) != 0);
    // (a replayed sequence that an earlier kernel has declared void walks nothing)
    if (group_void(g, cur.miss)) return;
    while (cur.more()) {
        PHASE_BEGIN(cls);
        const RowRec rec = cur.take();  // (its successor's record is requested now: RowCursor)
        if (rec.nnz <= NLO || rec.nnz > NMAX) {  // the other launch's row (uniform for the group)
            continue;
        }
        // table of this row: the smallest power of two >= 1.5 nnz (load <= 2/3), at least one slot
        // per lane; the class limit guarantees it fits (nnz <= 2/3 CAP)
        u32 bits = table_bits(rec.nnz, G::kIsBlock ? SPECK_LOAD_PCT : SPECK_LOAD_TINY_PCT);
        bits = min(max(bits, (u32)__builtin_ctz(G::SIZE)), (u32)__builtin_ctz(CAP));
        if constexpr (G::SIZE >= 64) bits = (u32)__builtin_amdgcn_readfirstlane((int)bits);
        const u32 cap_row = 1u << bits;
        // (16 bytes per lane and step; cap_row >= the group's lanes, a power of two)
        static_assert(sizeof(Acc<T>) == 8, "two accumulator cells per 16 bytes");
        for (u32 q = g.lane; q < cap_row / 4; q += G::SIZE)
            reinterpret_cast<uint4*>(keys)[q] = make_uint4(kEmptyKey, kEmptyKey, kEmptyKey, kEmptyKey);
        for (u32 q = g.lane; q < cap_row / 2; q += G::SIZE) reinterpret_cast<uint4*>(vals)[q] = make_uint4(0u, 0u, 0u, 0u);
        g.sync();
        PHASE_MARK(0);
        // (KEEP: the row's previous column ids, requested now -- emit_by_previous)
        constexpr u32 PF = (KEEP && SPECK_EMIT_BY_PREVIOUS && (NMAX + G::SIZE - 1) / G::SIZE <= SPECK_PF_MAX) ? (NMAX + G::SIZE - 1) / G::SIZE : 0;
        u32 pf_col[PF ? PF : 1], pf_left[PF ? PF : 1];
        if constexpr (PF > 0) {
#pragma unroll
            for (u32 q = 0; q < PF; ++q) {
                const u32 i = q * G::SIZE + g.lane;
                pf_col[q] = i < rec.nnz ? c_col[size_t(rec.base) + i] : 0u;
                pf_left[q] = (i && i < rec.nnz) ? c_col[size_t(rec.base) + i - 1] : 0u;
            }
        }
        bool gave_up = false;
        for_each_product<true>(g, src, rec.a0, rec.a1, meta, scan_scratch,
                               [&](const u32(&c)[kBatch], const T(&p)[kBatch], u32 n) {
                                   Acc<T> pa[kBatch];
#pragma unroll
                                   for (int u = 0; u < kBatch; ++u) pa[u] = p[u];
                                   gave_up |= table_accumulate_batch<Acc<T>, VERIFY>(keys, vals, bits, c, pa, n);
                               }, cls);
        PHASE_MARK(1);
        u32 found;
        if constexpr (KEEP && SPECK_EMIT_BY_PREVIOUS) {
            g.sync();
            found = emit_by_previous<G, T, CAP, PF>(g, keys, vals, cap_row, bits, rec.base, rec.nnz, c_col, c_val, scan_scratch,
                                                    pf_col, pf_left);
        } else if constexpr (MODE == SORT_RANK) {
            // scratch: the compacted keys over the table's keys, their slot numbers over the A-row staging
            // the groups of a wave take the same sort (no wave ever runs both)
            const bool narrow = __ballot(u64(rec.cmax) - rec.cmin >= u64(G::SIZE) * 32) == 0;
            if (narrow)
                found = emit_narrow_sorted<G, T, CAP, VERIFY>(g, keys, vals, keys, reinterpret_cast<uint2*>(m_av), cap_row,
                                                              rec.cmin, rec.base, c_col, c_val, rec.nnz);
            else
                found = emit_rank_sorted<G, T, CAP, NMAX, VERIFY>(g, keys, vals, keys, reinterpret_cast<u8*>(m_av), cap_row,
                                                                  rec.base, c_col, c_val, rec.nnz);
        } else {
            found = emit_bitmap_sorted<G, T, CAP, W1, NMAX, G::kIsBlock, VERIFY>(g, keys, vals, S, scan_scratch, cap_row, rec.cmin,
                                                                                 rec.cmax, rec.base, c_col, c_val, cls, rec.nnz);
        }
        if constexpr (VERIFY || (KEEP && SPECK_EMIT_BY_PREVIOUS)) {
            // (any lane: a key that found no slot, a table that holds another number of entries than the row was given,
            //  previous column ids that are not the table's keys)
            if (gave_up || found != rec.nnz) const_cast<DeviceStats*>(w.st)->capacity_miss = 1;
        }
        g.sync();
        PHASE_MARK(2);
    }
}

// (NUM_G8 / NUM_G16, expand / sort / compress in registers: num_esc_body in esc_rows.hpp)

constexpr u32 kW256W1 = 256;   // 256 Ki columns per sort window
constexpr u32 kW512W1 = 768;   // 768 Ki columns per sort window (its level-1 pairs fill the 6 KiB table exactly)
constexpr u32 kB2KW1 = 1024;   // 1 Mi columns per sort window
constexpr u32 kB8KW1 = 2048;  // 2 Mi columns per sort window

// ------------------------------------------------------------------ NUM_B8K in column slices (round 6)
// A row of 1 741 .. 6 963 entries used to own a 4 Ki / 8 Ki table -- 61 / 106 KiB of LDS and eight waves for ~40 us, one
// or two such rows per CU: on the webbase stand-in the 6 k rows of this class held more LDS-time than the 36 k NUM_B2K
// rows with twice their products, and the numeric phase is bound by exactly that.  Here the row keeps the
// 30 KiB / four waves of a NUM_B2K row and is produced in COLUMN SLICES, one after the other:
//   1. the row's products are counted per bin of a column histogram (kSliceBins bins of 2^shift columns over
//      [cmin, cmax]; the bins live where the table's accumulators will);
//   2. a bin holds at most min(products, width) distinct columns; with P the running sum of these bounds, bin i belongs to
//      slice (P_i - 1) / C, C = capacity + 1 - (largest bound) -- a slice then sums to <= capacity (its first bin starts
//      above k C - bound, its last ends at or below (k + 1) C), every bin decides alone, no sequential cutting;
//   3. per slice: clear the table, walk the row's products again and accumulate those whose column falls into the slice
//      (the walk re-reads B from L1 / L2: the row was read a moment ago), sort (emit_bitmap_sorted over the slice's columns
//      only) and store behind the previous slice -- slices are column-ordered, the row comes out sorted.
// The table cannot overflow whatever B holds (a product outside [cmin, cmax] lands in no slice), so plain probing is safe;
// VERIFY (replayed sequence without a symbolic pass): nothing is stored beyond the row's room, the total is compared.
// Host: ClassifyParams::slice_ops / RowWork::sliced, for cols(B) <= kSliceMaxCols; rows with more products than
// kSliceMaxOps (more slices than the cut arrays hold) take the dense-window / spill classes.
// Role: the reference gives such rows its largest shared-memory maps or the global ones
// (include/GPU/spECK_HashSpGEMM.cuh:1300-1436, source/GPU/Multiply.cu:700-760).
// MEASURED AND LOST (option slice_rows, off; profiles/r06_sliced_webbase.txt): the webbase stand-in 1.164 -> 1.234 ms
// complete, 0.79 -> 0.94 replayed.  The 6 018 rows of the class take 782 us in this kernel (beside the light launch, which
// they slow from 536 to 700 us) against 592 + 166 us in the two workgroup(512) launches: a row is walked 1 + S times
// (S = 2 .. 5), every walk is a few barrier-separated windows of four waves, and what these rows cost is not the LDS they
// hold but the LATENCY of each such step -- a row takes ~100 us this way against ~40.
constexpr u32 num_sliced_lds(u32 group_bytes) { return group_bytes + (2u * (kSliceMax + 1u) + 3u) / 4u * 16u; }

template <typename T, bool VERIFY = false>
__device__ __forceinline__ void num_sliced_body(unsigned char* smem, const ProductSrc<T>& src, const RowWork& w,
                                                u32* __restrict__ c_col, T* __restrict__ c_val, int cls, u32 bidx, u32 nblk,
                                                u32 hint = kNoCount)
{
    using G = Block<256>;
    constexpr u32 CAP = kNumB2KCap, CAPMAX = kNumB2KMaxNnz;
    static_assert(kSliceBins * 4u == CAP * sizeof(Acc<T>), "the histogram takes the place of the accumulators");
    static_assert(kSliceMaxWidth <= (CAPMAX + 1u) / 2u, "a bin's bound never exceeds a slice's quota (slice ids rise by one)");
    static_assert(kSliceMaxOps / (CAPMAX + 1u - kSliceMaxWidth) + 2u <= kSliceMax, "a row's slices fit the cut arrays");
    const G g;
    Acc<T>* vals = reinterpret_cast<Acc<T>*>(smem);
    u32* keys = reinterpret_cast<u32*>(vals + CAP);
    T* m_av = reinterpret_cast<T*>(keys + CAP);
    u32* m_incl = reinterpret_cast<u32*>(m_av + G::SIZE);
    u32* scan_scratch = m_incl + 2 * G::SIZE;
    RowMeta<T> meta{m_incl, m_incl + G::SIZE, m_av, scan_scratch + scan_scratch_words<G, 256>()};
    u32* s_cut = reinterpret_cast<u32*>(smem + num_group_lds<G, T, CAP, 256>());  // first bin of slice k
    u32* s_pref = s_cut + kSliceMax + 1;                                           // bound of the slices before k
    u32* hist = reinterpret_cast<u32*>(smem);
    u32* S = reinterpret_cast<u32*>(smem);
    const u32 t = threadIdx.x;
    RowCursor cur = open_list<false>(w, cls, hint, bidx, nblk, 1u, 0u, (w.xcd_aware & 4u) != 0);
    if (block_void(cur.miss)) return;
    while (cur.more()) {
        const RowRec rec = cur.take();
        const u32 span_row = rec.cmax - rec.cmin;  // (columns - 1: no overflow for a row that reaches every column)
        u32 shift = 0;
        while ((span_row >> shift) >= kSliceBins) ++shift;
        if ((1u << shift) > kSliceMaxWidth) {  // (column ids beyond cols(B): the input check rejects the call)
            if (t == 0) const_cast<DeviceStats*>(w.st)->capacity_miss = 1;
            continue;
        }
        // (a column outside [cmin, cmax] -- a B the input check is about to reject -- counts in the LAST bin of the row:
        //  every slice starts at or below cmax, and no slice ever takes such a product)
        const u32 last_bin = span_row >> shift;
        for (u32 q = t; q < kSliceBins / 4; q += G::SIZE) reinterpret_cast<uint4*>(hist)[q] = make_uint4(0u, 0u, 0u, 0u);
        g.sync();
        for_each_product<false>(g, src, rec.a0, rec.a1, meta, scan_scratch,
                                [&](const u32(&c)[kBatch], const T(&)[kBatch], u32 n) {
#pragma unroll
                                    for (int u = 0; u < kBatch; ++u)
                                        if ((u32)u < n) atomicAdd(&hist[min((c[u] - rec.cmin) >> shift, last_bin)], 1u);
                                }, cls);
        // (for_each_product ends behind a barrier)
        constexpr u32 PER = kSliceBins / G::SIZE;
        static_assert(PER % 4 == 0, "bins are read 16 bytes at a time");
        const u32 width = 1u << shift;
        u32 b[PER], sum = 0, mx = 0;
#pragma unroll
        for (u32 q = 0; q < PER / 4; ++q) {
            const uint4 h = reinterpret_cast<const uint4*>(hist)[t * (PER / 4) + q];
            b[4 * q] = min(h.x, width), b[4 * q + 1] = min(h.y, width), b[4 * q + 2] = min(h.z, width), b[4 * q + 3] = min(h.w, width);
        }
#pragma unroll
        for (u32 i = 0; i < PER; ++i) sum += b[i], mx = max(mx, b[i]);
        mx = wave_reduce_max(mx);
        if (lane_id() == 0) s_cut[t >> 6] = mx;
        u32 total;
        u32 run = block_exclusive_scan<256>(sum, scan_scratch, &total);  // (its barriers publish s_cut[0..3] as well)
        mx = max(max(s_cut[0], s_cut[1]), max(s_cut[2], s_cut[3]));
        g.sync();
        const u32 quota = CAPMAX + 1u - max(mx, 1u);
        const u32 nslices = total ? (total - 1u) / quota + 1u : 0u;
        if (nslices > kSliceMax) {  // (cannot happen to a row the classifier let in; a replayed sequence on other inputs)
            if (t == 0) const_cast<DeviceStats*>(w.st)->capacity_miss = 1;
            continue;
        }
        u32 sid_prev = run ? (run - 1u) / quota : 0u;
        if (t == 0) s_cut[0] = 0u, s_pref[0] = 0u;
        if (t == G::SIZE - 1u) s_pref[nslices] = total;
#pragma unroll
        for (u32 i = 0; i < PER; ++i) {
            run += b[i];
            const u32 sid = run ? (run - 1u) / quota : 0u;
            if (sid != sid_prev) s_cut[sid] = t * PER + i, s_pref[sid] = run - b[i];
            sid_prev = sid;
        }
        g.sync();
        u32 emitted = 0;
        for (u32 k = 0; k < nslices; ++k) {
            const u32 lo = rec.cmin + (s_cut[k] << shift);
            const u32 last = k + 1u < nslices ? rec.cmin + (s_cut[k + 1u] << shift) - 1u : rec.cmax;  // (inclusive)
            const u32 cols_m1 = last - lo;
            u32 bits = table_bits(s_pref[k + 1u] - s_pref[k], SPECK_LOAD_PCT);
            bits = min(max(bits, (u32)__builtin_ctz(G::SIZE)), (u32)__builtin_ctz(CAP));
            bits = (u32)__builtin_amdgcn_readfirstlane((int)bits);
            const u32 cap_row = 1u << bits, mask = cap_row - 1u;
            for (u32 q = t; q < cap_row / 4; q += G::SIZE)
                reinterpret_cast<uint4*>(keys)[q] = make_uint4(kEmptyKey, kEmptyKey, kEmptyKey, kEmptyKey);
            for (u32 q = t; q < cap_row / 2; q += G::SIZE) reinterpret_cast<uint4*>(vals)[q] = make_uint4(0u, 0u, 0u, 0u);
            g.sync();
            for_each_product<true>(g, src, rec.a0, rec.a1, meta, scan_scratch,
                                   [&](const u32(&c)[kBatch], const T(&p)[kBatch], u32 n) {
                                       u32 slot[kBatch], old[kBatch];
                                       bool in[kBatch];
#pragma unroll
                                       for (int u = 0; u < kBatch; ++u) {
                                           in[u] = (u32)u < n && c[u] - lo <= cols_m1;
                                           slot[u] = (c[u] * 0x9E3779B1u) >> (32u - bits);
                                           old[u] = kEmptyKey;
                                           if (in[u]) old[u] = atomicCAS(&keys[slot[u]], kEmptyKey, c[u]);
                                       }
#pragma unroll
                                       for (int u = 0; u < kBatch; ++u) {
                                           if (!in[u]) continue;
                                           if (old[u] != kEmptyKey && old[u] != c[u]) {
                                               const u32 step = probe_step(c[u], 32u - bits);
                                               u32 inc = kFirstProbeInc ? kFirstProbeInc : step;
                                               do {
                                                   slot[u] = (slot[u] + inc) & mask;
                                                   inc = step;
                                                   old[u] = atomicCAS(&keys[slot[u]], kEmptyKey, c[u]);
                                               } while (old[u] != kEmptyKey && old[u] != c[u]);
                                           }
                                           atomicAdd(&vals[slot[u]], (Acc<T>)p[u]);
                                       }
                                   }, cls);
            const u32 room = rec.nnz - min(rec.nnz, emitted);
            emitted += emit_bitmap_sorted<G, T, CAP, kB2KW1, CAPMAX, true, VERIFY>(g, keys, vals, S, scan_scratch, cap_row, lo, last,
                                                                                  rec.base + emitted, c_col, c_val, cls, room);
            g.sync();
        }
        if constexpr (VERIFY) {
            if (emitted != rec.nnz) const_cast<DeviceStats*>(w.st)->capacity_miss = 1;
        }
    }
}

// ------------------------------------------------------------------ NUM_D1/D2
template <typename T, u32 WCOLS, int THREADS>
constexpr u32 num_dense_lds()
{
    return (WCOLS + THREADS) * (u32)sizeof(Acc<T>) +
           (2 * (WCOLS / 32) + 2 * THREADS + THREADS / 64 + 2 + win_words<Block<THREADS>>() + 3) / 4 * 16;
}

// (VERIFY: as num_hash_body -- nothing is stored beyond the row's room, the distinct columns found are compared with it)
template <typename T, u32 WCOLS, int THREADS, bool VERIFY = false>
__device__ __forceinline__ void num_dense_body(unsigned char* smem, const ProductSrc<T>& src, const RowWork& w,
                                               u32* __restrict__ c_col, T* __restrict__ c_val, int cls,
                                               u32 bidx, u32 nblk, u32 hint = kNoCount)
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
    RowCursor cur = open_list<false>(w, cls, hint, bidx, nblk, 1u, 0u, (w.xcd_aware & 2u) != 0);
    if (block_void(cur.miss)) return;
    for (u32 i = threadIdx.x; i < WCOLS; i += THREADS) vals[i] = 0;
    for (u32 i = threadIdx.x; i < WORDS; i += THREADS) bm[i] = 0;
    __syncthreads();
    while (cur.more()) {
        const RowRec rec = cur.take();  // (its successor's record is requested now: RowCursor)
        u32 emitted = 0;
        // a row wider than one window: per-entry cursors, every B entry is read once (WindowCursors)
        const bool multi = u64(rec.cmax) - rec.cmin + 1 > WCOLS;
        const WindowCursors<THREADS> cur{src.b_sl, src.b_col, src.w_sl, rec.a0, rec.a1};
        ProductSrc<T> wsrc = src;
        if (multi) {
            wsrc.b_sl = src.w_sl;
            cur.reset();
        }
        u32 wbase = rec.cmin;
        while (true) {
            if (multi) {
                wbase = cur.next_window(WCOLS, scratch);
                if (wbase == 0xFFFFFFFFu) break;
            }
            const u64 left = u64(rec.cmax) - wbase + 1;
            const u32 ncols = left < WCOLS ? (u32)left : WCOLS;
            const u32 nwords = (ncols + 31) >> 5;
            // (the accumulators and the bitmap are clean here: cleared once per workgroup below, and every
            //  emit zeroes exactly the cells it read -- a window costs its entries, not its width)
            for_each_product<true>(g, wsrc, rec.a0, rec.a1, meta, scratch,
                                   [&](const u32(&c)[kBatch], const T(&p)[kBatch], u32 n) {
#pragma unroll
                                       for (int u = 0; u < kBatch; ++u) {
                                           const u32 d = c[u] - wbase;
                                           const bool in = (u32)u < n && d < ncols;
                                           if (in) atomicAdd(&vals[d], (Acc<T>)p[u]);
                                           bitmap_or_runs(bm, in ? d >> 5 : 0xFFFFFFFFu, 1u << (d & 31));
                                       }
                                   });
            const u32 total = bitmap_prefix(g, bm, pref, nwords, scratch);
            if (total * 4u < ncols) {
                // sparse window: a thread per bitmap word walks its set bits
                for (u32 i = threadIdx.x; i < nwords; i += THREADS) {
                    u32 word = bm[i], r = emitted + pref[i];
                    if (word) bm[i] = 0;
                    while (word) {
                        const u32 d = i * 32u + (u32)__builtin_ctz(word);
                        word &= word - 1u;
                        if (!VERIFY || r < rec.nnz) {
                            c_col[rec.base + r] = wbase + d;
                            c_val[rec.base + r] = (T)vals[d];
                        }
                        vals[d] = 0;
                        ++r;
                    }
                }
            } else {
                for (u32 d = threadIdx.x; d < ncols; d += THREADS) {
                    const u32 word = bm[d >> 5];
                    if (word & (1u << (d & 31))) {
                        const u32 r = emitted + pref[d >> 5] + __popc(word & ((1u << (d & 31)) - 1u));
                        if (!VERIFY || r < rec.nnz) {
                            c_col[rec.base + r] = wbase + d;
                            c_val[rec.base + r] = (T)vals[d];
                        }
                        vals[d] = 0;
                    }
                }
                __syncthreads();
                for (u32 i = threadIdx.x; i < nwords; i += THREADS) bm[i] = 0;
            }
            emitted += total;
            __syncthreads();
            if (!multi) break;
        }
        if constexpr (VERIFY) {
            if (emitted != rec.nnz) const_cast<DeviceStats*>(w.st)->capacity_miss = 1;
        }
    }
}

// ------------------------------------------------------------------ numeric-first rows (SYM_NF / NUM_NFCOPY)
// A row whose reachable column range fits one dense window needs no nnz to size anything, so its numeric
// kernel does not have to wait for the symbolic phase -- it REPLACES it: the row is accumulated in the dense
// window (as NUM_D1), written sorted to a scratch slot (slot size = min(column range, products) >= nnz, offsets from the
// ordered scatter of the analysis) and counted.  After the scan NUM_NFCOPY moves it to its place in C.
// For banded / FEM inputs (cant: every row) the whole symbolic walk -- as long as the numeric one --
// turns into one copy of C.  (The reference always runs both phases; new functionality.)
// (DIRECT: the launch of a replayed sequence that places the rows straight into C -- a template parameter only so that
//  the kernel carries another NAME than the eager launch in a trace; the code is the same)
template <typename T, int THREADS, bool DIRECT = false>
__global__ __launch_bounds__(THREADS) void nf_dense_kernel(ProductSrc<T> src, const u32* a_ro, RowWork w,
                                                           u32* __restrict__ counts, u32 wcols)
{
    SPECK_POISON();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // The window is as wide as the widest numeric-first row of the previous identical call / of the analysis
    // read-back (a multiple of 256 columns), not kNumD1Cols: the launch holds 160 KiB / LDS workgroups per CU,
    // and this kernel's time is ~ floor + latency / (waves per SIMD) -- cant stand-in, range 2187: 6 instead of
    // 4 workgroups per CU.
    const u32 WCOLS = wcols, WORDS = WCOLS / 32;
    if (block_void(w.st->capacity_miss)) return;  // the scratch pool of this (replayed) sequence is too small
    if (w.st->sym.count[SYM_NF] == 0) return;  // eager path: launched for every class, rows or not
    src.rebase(a_ro);
    using G = Block<THREADS>;
    const G g;
    // LDS: values | a_ik staging | one FLAG BYTE per column | A-row staging (incl, off) | scan scratch | owner windows.
    // A product marks its column with a plain byte store -- no atomic, no run detection in registers (this kernel
    // sits at its VALU ceiling on the cant stand-in); the bitmap the emit step ranks with is built from the flags
    // once per row, over the A-row staging area, which is dead between the walk and the next row.
    Acc<T>* vals = reinterpret_cast<Acc<T>*>(smem);
    T* m_av = reinterpret_cast<T*>(vals + WCOLS);
    unsigned char* flags = reinterpret_cast<unsigned char*>(m_av + THREADS);
    u32* flag_words = reinterpret_cast<u32*>(flags);
    u32* stage = reinterpret_cast<u32*>(flags + WCOLS);
    u32* bm = stage;
    u32* pref = stage + WORDS;
    u32* scratch = stage + 2 * THREADS;
    RowMeta<T> meta{stage, stage + THREADS, m_av, scratch + THREADS / 64 + 2};
    static_assert(THREADS >= kNumD1Cols / 32, "bitmap + prefix fit the staging area");
    const bool direct = DIRECT;  // replayed sequence: rows go straight to C (RowWork::nf_pred_off)
    u32* __restrict__ o_col = direct ? w.nf_direct_col : w.nf_col;
    T* __restrict__ o_val = static_cast<T*>(direct ? w.nf_direct_val : w.nf_val);
    RowCursor cur = open_list<true>(w, SYM_NF, kNoCount, blockIdx.x, gridDim.x, 1u, 0u, (w.xcd_aware & 2u) != 0);
    for (u32 i = threadIdx.x; i < WCOLS; i += THREADS) vals[i] = 0;
    for (u32 i = threadIdx.x; i < WCOLS / 4; i += THREADS) flag_words[i] = 0;
    __syncthreads();
    while (cur.more()) {
        const RowRec rec = cur.take();  // (its successor's record is requested now: RowCursor)
        u64 slot = 0;
        u32 slot_len = 0;
        if (direct) {
            slot = w.nf_pred_off[rec.row];
            slot_len = w.nf_pred_off[rec.row + 1] - (u32)slot;
        } else
            slot = w.nf_off[rec.row];
        const u32 wbase = rec.cmin, ncols = rec.cmax - rec.cmin + 1u, nwords = (ncols + 31) >> 5;
        // a replayed sequence met a wider row than its window was sized for, or a slot past the pool it was
        // planned with: the complete call re-runs
        if (ncols > WCOLS || (!direct && slot + nf_slot_entries(rec.cmin, rec.cmax, rec.ops) > w.nf_cap)) {
            if (threadIdx.x == 0) const_cast<DeviceStats*>(w.st)->capacity_miss = 1;
            continue;
        }
        for_each_product<true>(g, src, rec.a0, rec.a1, meta, scratch,
                               [&](const u32(&c)[kBatch], const T(&p)[kBatch], u32 n) {
#pragma unroll
                                   for (int u = 0; u < kBatch; ++u) {
                                       const u32 d = c[u] - wbase;
                                       if ((u32)u < n && d < ncols) {
                                           atomicAdd(&vals[d], (Acc<T>)p[u]);
                                           flags[d] = 1;
                                       }
                                   }
                               });
        // flags -> bitmap (8 flag words per bitmap word; the flags are cleared on the way)
        for (u32 i = threadIdx.x; i < nwords; i += THREADS) {
            u32 mask = 0;
#pragma unroll
            for (u32 j = 0; j < 8; ++j) {
                const u32 x = flag_words[i * 8 + j];
                if (x) flag_words[i * 8 + j] = 0;
                mask |= ((x * 0x01020408u) >> 24 & 0xFu) << (4 * j);
            }
            bm[i] = mask;
        }
        __syncthreads();
        const u32 total = bitmap_prefix(g, bm, pref, nwords, scratch);
        // direct placement: only a row that still has the nnz its place in C was made for is written (the window is
        // cleaned either way); any other row voids the replay
        const bool place = !direct || total == slot_len;
        for (u32 d = threadIdx.x; d < ncols; d += THREADS) {
            const u32 word = bm[d >> 5];
            if (word & (1u << (d & 31))) {
                const u32 r = pref[d >> 5] + __popc(word & ((1u << (d & 31)) - 1u));
                if (place) {
                    o_col[slot + r] = wbase + d;
                    o_val[slot + r] = (T)vals[d];
                }
                vals[d] = 0;
            }
        }
        if (threadIdx.x == 0) {
            counts[rec.row] = total;
            if (!place) const_cast<DeviceStats*>(w.st)->capacity_miss = 1;
        }
        __syncthreads();  // bitmap and prefix are dead: the next row stages its A entries over them
    }
}

// a wave per row: scratch slot -> C
template <typename T>
__global__ __launch_bounds__(256) void nf_copy_kernel(RowWork w, u32* __restrict__ c_col, T* __restrict__ c_val)
{
    SPECK_POISON();
    if (block_void(w.st->capacity_miss)) return;
    const u32 count = min(w.st->num.count[NUM_NFCOPY], w.m);
    const u32* __restrict__ s_col = w.nf_col;
    const T* __restrict__ s_val = static_cast<const T*>(w.nf_val);
    const u32 lane = lane_id();
    const u32 wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = (gridDim.x * 256u) >> 6;
    for (u32 idx = wave; idx < count; idx += nwaves) {
        const RowRec rec = *class_rec_at(w.recs, w.m, NUM_NFCOPY, idx);
        const u64 slot = w.nf_off[rec.row];
        for (u32 j = lane; j < rec.nnz; j += 64) {
            c_col[size_t(rec.base) + j] = s_col[slot + j];
            c_val[size_t(rec.base) + j] = s_val[slot + j];
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
    SPECK_POISON();
    src.rebase(a_ro);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    num_direct_body<T, THREADS>(smem, src, w, c_col, c_val, blockIdx.x, gridDim.x);
}

template <class G, typename T, u32 CAP, u32 W1, u32 NMAX, int MODE, int THREADS, u32 NLO = 0, bool VERIFY = false>
__global__ __launch_bounds__(THREADS) void num_hash_kernel(ProductSrc<T> src, const u32* a_ro, RowWork w,
                                                           u32* __restrict__ c_col,
                                                           T* __restrict__ c_val, int cls)
{
    SPECK_POISON();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    src.rebase(a_ro);
    num_hash_body<G, T, CAP, W1, NMAX, MODE, THREADS, NLO, VERIFY>(smem, src, w, c_col, c_val, cls, blockIdx.x,
                                                                   gridDim.x);
}

template <typename T, bool VERIFY = false>
__global__ __launch_bounds__(256) void num_sliced_kernel(ProductSrc<T> src, const u32* a_ro, RowWork w, u32* __restrict__ c_col,
                                                         T* __restrict__ c_val, int cls)
{
    SPECK_POISON();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    src.rebase(a_ro);
    num_sliced_body<T, VERIFY>(smem, src, w, c_col, c_val, cls, blockIdx.x, gridDim.x);
}

template <typename T, u32 WCOLS, int THREADS, bool VERIFY = false>
__global__ __launch_bounds__(THREADS) void num_dense_kernel(ProductSrc<T> src, const u32* a_ro, RowWork w,
                                                            u32* __restrict__ c_col,
                                                            T* __restrict__ c_val, int cls)
{
    SPECK_POISON();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    src.rebase(a_ro);
    num_dense_body<T, WCOLS, THREADS, VERIFY>(smem, src, w, c_col, c_val, cls, blockIdx.x, gridDim.x);
}

template <typename T, u32 L>
__global__ __launch_bounds__(256) void num_esc_kernel(ProductSrc<T> src, const u32* a_ro, RowWork w,
                                                      u32* __restrict__ c_col, T* __restrict__ c_val, int cls)
{
    SPECK_POISON();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    src.rebase(a_ro);
    num_esc_body<T, L, 256>(smem, src, w, c_col, c_val, cls, blockIdx.x, gridDim.x);
}

template <typename T, u32 L>
__global__ __launch_bounds__(256) void num_escw_kernel(ProductSrc<T> src, const u32* a_ro, RowWork w,
                                                       u32* __restrict__ c_col, T* __restrict__ c_val, int cls)
{
    SPECK_POISON();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    src.rebase(a_ro);
    num_escw_body<T, L, 256>(smem, src, w, c_col, c_val, cls, blockIdx.x, gridDim.x);
}


// WITH_ESC = false: the launch of a sequence whose register-class rows are finished in its symbolic phase (fused
// replay) -- those bodies are not even compiled in, and the kernel carries another NAME than the launch of an eager
// call, so that a kernel trace tells the two apart (profiles/: per-kernel averages of the replayed sequence alone).
// VERIFY = true: the launch of a sequence without a symbolic pass for its rows (RowWork::verify_numeric): every body checks
// the row's nnz itself (num_hash_body / num_dense_body; the scaled copies of NUM_DIRECT hold what the analysis verifies).
template <typename T, bool WITH_ESC = true, bool VERIFY = false>
__global__ __launch_bounds__(256) void num_light_kernel(ProductSrc<T> src, const u32* a_ro, RowWork w,
                                                        u32* __restrict__ c_col, T* __restrict__ c_val,
                                                        ClassGrid cg)
{
    SPECK_POISON();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    src.rebase(a_ro);
    const u32 b = blockIdx.x;
    // launch order (ClassGrid slots): D1, B2K, W512, W256, R64, R32, W128, G16, G8, DIRECT
    // (every body starts with open_list: its first row ids are requested by the host-known count while the
    //  device-side table and the capacity_miss flag are still on their way)
    if (b < cg.first[1])
        num_dense_body<T, kNumD1Win, 256, VERIFY>(smem, src, w, c_col, c_val, NUM_D1, b - cg.first[0], cg.first[1] - cg.first[0], cg.cnt[0]);
    else if (b < cg.first[2])
        num_hash_body<Block<256>, T, kNumB2KCap, kB2KW1, kNumB2KMaxNnz, SORT_BITMAP, 256, 0, VERIFY>(
            smem, src, w, c_col, c_val, NUM_B2K, b - cg.first[1], cg.first[2] - cg.first[1], cg.cnt[1]);
    else if (b < cg.first[3])
        num_hash_body<SubWave<64>, T, kNumW512Cap, kW512W1, kNumW512MaxNnz, SORT_BITMAP, 256, 0, VERIFY>(
            smem, src, w, c_col, c_val, NUM_W512, b - cg.first[2], cg.first[3] - cg.first[2], cg.cnt[2]);
    else if (b < cg.first[4])
        num_hash_body<SubWave<32>, T, kNumW256Cap, kW256W1, kNumW256MaxNnz, SORT_BITMAP, 256, 0, VERIFY>(
            smem, src, w, c_col, c_val, NUM_W256, b - cg.first[3], cg.first[4] - cg.first[3], cg.cnt[3]);
    else if (b < cg.first[5]) {
        if constexpr (WITH_ESC)
            num_escw_body<T, 64, 256>(smem, src, w, c_col, c_val, NUM_R64, b - cg.first[4], cg.first[5] - cg.first[4], cg.cnt[4]);
    } else if (b < cg.first[6]) {
        if constexpr (WITH_ESC)
            num_escw_body<T, 32, 256>(smem, src, w, c_col, c_val, NUM_R32, b - cg.first[5], cg.first[6] - cg.first[5], cg.cnt[5]);
    } else if (b < cg.first[7])
        num_hash_body<SubWave<32>, T, kNumW128Cap, 0, kNumW128MaxNnz, SORT_RANK, 256, 0, VERIFY>(
            smem, src, w, c_col, c_val, NUM_W128, b - cg.first[6], cg.first[7] - cg.first[6], cg.cnt[6]);
    else if (b < cg.first[8]) {
        if constexpr (WITH_ESC)
            num_esc_body<T, 16, 256>(smem, src, w, c_col, c_val, NUM_G16, b - cg.first[7], cg.first[8] - cg.first[7], cg.cnt[7]);
    } else if (b < cg.first[9]) {
        if constexpr (WITH_ESC)
            num_esc_body<T, 8, 256>(smem, src, w, c_col, c_val, NUM_G8, b - cg.first[8], cg.first[9] - cg.first[8], cg.cnt[8]);
    } else if (b < cg.first[10])
        num_direct_body<T, 256>(smem, src, w, c_col, c_val, b - cg.first[9], cg.first[10] - cg.first[9], cg.cnt[9]);
    else if (!w.st->capacity_miss)  // the staged row offsets -> C.row_offsets (RowWork::off_src; not for a sequence declared void)
        for (u32 i = (b - cg.first[10]) * 256u + threadIdx.x; i < w.off_n; i += (cg.first[11] - cg.first[10]) * 256u)
            w.off_dst[i] = w.off_src[i];
}

// ------------------------------------------------------------------ one walk for the hash rows (round 6)
// Inputs whose rows ALL fit the 256-entry sub-wave table (stencils, meshes: the nlpkkt stand-in -- 8.4 M rows of 729
// products and 125 entries) are walked ONCE: no symbolic pass, no scan kernel, no read-back.  A workgroup owns EIGHT
// CONTIGUOUS rows (two per wave, as NUM_W256): every 32-lane group accumulates its row into a table sized from the row's
// PRODUCT bound (at most 256 slots; bounded probing, so a row with more distinct columns than slots is noticed, not
// overrun), counts the slots it filled, the workgroup publishes the eight counts' sum, takes the entries of C before it
// from the three-level chain (chain3.hpp) -- while its waves would otherwise idle through one trip to memory -- and then
// sorts and stores each row at its place (emit_bitmap_sorted; the row offsets go to scratch like the scan's).  What the
// table cannot hold (nnz > its slots), a row beyond `max_ops` products, buffers of C that do not hold the product:
// capacity_miss, the two-phase call re-runs.  Nothing of a previous call is read; the host only CHOOSES this path from the
// previous call's figures (longest row of C <= 170, as the two-phase NUM_W256 class guarantees; pipeline.hip).
// Role: the reference's symbolic + scan + numeric passes for these rows (source/GPU/Multiply.cu:488-602, 848-1014).
// MEASURED AND LOST (option one_walk_hash, off): a fifth of the nlpkkt stand-in 4.29 ms against 4.71 for the two-phase call
// (the kernel 3.78 ms against 2.78 for the numeric light launch of the same rows), the FULL size 33.1 ms against 24.2 (the
// kernel 30.7 against 14.9): a group cannot store before every group before it has published its count, so the resident
// workgroups -- which hold their LDS while they wait -- finish in lock step with their slowest predecessor, and the longer
// and more variable the gathers (B no longer cache-resident at full size) the more of the chip waits.  DESIGN.md 4.8.
template <typename T>
__global__ __launch_bounds__(256) void walk_hash_kernel(ProductSrc<T> src, WalkHashArgs a, Chain3 chain)
{
    SPECK_POISON();
    using G = SubWave<32>;
    constexpr u32 CAP = kNumW256Cap, NG = 256 / G::SIZE;
    static_assert(NG == kWalkHashRows, "eight rows per workgroup");
    constexpr u32 kGroupBytes = num_group_lds<G, T, CAP, 256>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ u32 s_cnt[NG];
    __shared__ u32 s_bad;
    __shared__ u64 s_tmp[12];
    src.rebase(a.a_ro);
    const G g;
    const u32 t = threadIdx.x, gid = t / G::SIZE;
    unsigned char* mine = smem + gid * kGroupBytes;
    Acc<T>* vals = reinterpret_cast<Acc<T>*>(mine);
    u32* keys = reinterpret_cast<u32*>(vals + CAP);
    T* m_av = reinterpret_cast<T*>(keys + CAP);
    u32* m_incl = reinterpret_cast<u32*>(m_av + G::SIZE);
    u32* scan_scratch = m_incl + 2 * G::SIZE;
    RowMeta<T> meta{m_incl, m_incl + G::SIZE, m_av, scan_scratch + scan_scratch_words<G, 256>()};
    u32* S = reinterpret_cast<u32*>(mine);
    const u32 ngroups = (a.m + NG - 1u) / NG;
    // (a call an earlier kernel has declared void computes nothing; the chain still runs so that the last group reports)
    const bool void_call = block_void(a.st->capacity_miss);  // (one decision per workgroup: row_groups.hpp)
    // ONE 8-row group per workgroup: a group waits only for groups with a lower number, i.e. workgroups DISPATCHED BEFORE
    // this one (chain.hpp's argument).  Measured and dropped: a persistent grid of exactly the chip's capacity with the
    // groups taken in turn (6.5 ms instead of 3.7 at a fifth of the nlpkkt stand-in: the input check beside it held wave
    // slots, part of the grid could not start, and the resident part polled the fabric until it did); several CONSECUTIVE
    // groups per workgroup (seconds: a workgroup's second group needs the LAST group of the workgroup before it, which is
    // computed three turns later -- the workgroups run one behind the other).
    {
        const u32 grp = blockIdx.x;
        const u32 row = grp * NG + gid;
        const bool valid = row < a.m;
        u32 a0 = 0, a1 = 0, ops = 0, cmin = 0, cmax = 0;
        if (valid) {
            a0 = a.a_ro[row];
            a1 = a.a_ro[row + 1];
            ops = a.row_ops[row];
            cmin = a.row_col_min[row];
            cmax = a.row_col_max[row];
        }
        if (t == 0) {
            s_bad = 0;
            s_tmp[8] = 0;  // (chain3_finish: the timeout flag; the barrier behind the counts orders it)
        }
        bool bad = valid && ops > a.max_ops;
        const bool walk = valid && !bad && !void_call && ops != 0;
        // the row's table: the smallest power of two that keeps min(products, class limit) entries at the class load, at
        // least one slot per lane
        u32 bits = table_bits(min(ops, kNumW256MaxNnz), SPECK_LOAD_TINY_PCT);
        bits = min(max(bits, (u32)__builtin_ctz(G::SIZE)), (u32)__builtin_ctz(CAP));
        const u32 cap_row = 1u << bits;
        for (u32 q = g.lane; q < cap_row / 4; q += G::SIZE)
            reinterpret_cast<uint4*>(keys)[q] = make_uint4(kEmptyKey, kEmptyKey, kEmptyKey, kEmptyKey);
        for (u32 q = g.lane; q < cap_row / 2; q += G::SIZE) reinterpret_cast<uint4*>(vals)[q] = make_uint4(0u, 0u, 0u, 0u);
        g.sync();
        bool gave_up = false;
        // (an idle group walks an empty entry range: every lane of the wave takes part in the wave-wide steps)
        for_each_product<true>(g, src, walk ? a0 : 0u, walk ? a1 : 0u, meta, scan_scratch,
                               [&](const u32(&c)[kBatch], const T(&p)[kBatch], u32 n) {
                                   Acc<T> pa[kBatch];
#pragma unroll
                                   for (int u = 0; u < kBatch; ++u) pa[u] = p[u];
                                   gave_up |= table_accumulate_batch<Acc<T>, true>(keys, vals, bits, c, pa, n);
                               }, NUM_W256);
        g.sync();
        u32 cnt = 0;
        for (u32 q = g.lane; q < cap_row; q += G::SIZE) cnt += keys[q] != kEmptyKey ? 1u : 0u;
        cnt = g.reduce_add(cnt, nullptr);
        bad |= g.ballot(gave_up) != 0;
        if (!walk) cnt = 0;
        if (g.lane == 0) {
            s_cnt[gid] = cnt;
            if (bad) s_bad = 1;
        }
        if (a.want_bytes && a.bytes_acc && g.lane == 0 && cnt)
            atomicAdd((unsigned long long*)&a.bytes_acc[kMaxClasses + NUM_W256], (unsigned long long)numeric_row_bytes(a1 - a0, ops, cnt, a.vsize));
        __syncthreads();
        u32 tile = 0, before_me = 0, tile_max = 0;
#pragma unroll
        for (u32 k = 0; k < NG; ++k) {
            before_me += k < gid ? s_cnt[k] : 0u;
            tile += s_cnt[k];
            tile_max = max(tile_max, s_cnt[k]);
        }
        const bool tile_bad = s_bad != 0;
        // my group's count goes out NOW; the rows are ranked (the two-level bitmap sort, a third of a row's time) before
        // anybody asks for the prefix: by then the groups before this one have published theirs
        // ... and the words of the groups before mine are REQUESTED now: their trip runs beside the sort as well
        const Chain3Words<1> pend = chain3_begin(chain, grp, tile);
        constexpr u32 OWN = CAP / G::SIZE;
        u32 sk[OWN], sr[OWN];
        Acc<T> sv[OWN];
        // (one sort window covers the row: wave-uniform, so that no wave runs both forms)
        const bool one_window = __ballot(valid && u64(cmax) - cmin + 1u > u64(kW256W1) * 1024u) == 0;
        if (one_window && !void_call && !tile_bad)
            bitmap_rank_slots<G, T, CAP, kW256W1>(g, keys, vals, S, scan_scratch, cap_row, cmin, cmax < cmin ? cmin : cmax, sk, sv, sr);
        bool chain_ok = true;
        u64 pre = 0;
        if (!(a.debug & 1u)) pre = chain3_finish(chain, grp, ngroups, tile, pend, s_tmp, &chain_ok);
        else pre = u64(grp) * 1000u;  // (development: wrong places, no look-back)
        const bool fits = pre + tile <= a.c_cap && pre + tile <= 0xFFFFFFFFull;
        if (t == 0) {
            if (tile_max) atomicMax(&a.st->max_row_nnz_c, tile_max);   // (device-scope atomics: any group may hold the maximum)
            if (tile_bad || !fits || !chain_ok) atomicOr(&a.st->capacity_miss, 1u);
            if (grp == ngroups - 1u) {
                const u64 nnz_c = pre + tile;
                a.st->nnz_c = nnz_c;
                if (nnz_c > 0xFFFFFFFFull) a.st->nnz_overflow = 1;
                if (nnz_c > a.c_cap) atomicOr(&a.st->capacity_miss, 1u);
                a.st->walk_rows = a.m;
                if (!chain_ok || __hip_atomic_load(chain.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    a.st->chain_error = 1;
                    atomicOr(&a.st->capacity_miss, 1u);
                }
                a.offsets_out[a.m] = (u32)nnz_c;
                if (a.pred_off_out) a.pred_off_out[a.m] = (u32)nnz_c;
            }
        }
        // (a truncated prefix: this workgroup places nothing more -- the groups behind it time out likewise and report)
        if (!chain_ok) return;
        if (fits && !tile_bad && !void_call) {
            const u32 base = (u32)pre + before_me;
            if (valid && g.lane == 0) {
                a.offsets_out[row] = base;
                if (a.pred_off_out) a.pred_off_out[row] = base;
            }
            // store: at most `cnt` entries leave, whatever B holds (a column outside the row's range got no rank)
            if (one_window)
                store_ranked_slots<T, OWN>(sk, sv, sr, base, cnt, a.c_col, static_cast<T*>(a.c_val));
            else
                emit_bitmap_sorted<G, T, CAP, kW256W1, CAP, false, true>(g, keys, vals, S, scan_scratch, cap_row, cmin, cmax, base,
                                                                         a.c_col, static_cast<T*>(a.c_val), NUM_W256, cnt);
        }
    }
}

template <typename T>
void launch_walk_hash(hipStream_t s, const WalkHashArgs& args, const ProductSrc<T>& src, const Chain3& chain, hipEvent_t e0,
                      hipEvent_t e1)
{
    const u32 groups = (args.m + kWalkHashRows - 1) / kWalkHashRows;
    const u32 lds = kWalkHashRows * num_group_lds<SubWave<32>, T, kNumW256Cap, 256>();
    const WalkHashArgs& a = args;
    const u32 grid = groups;
    SPECK_LAUNCH_TIMED((walk_hash_kernel<T>), dim3(grid), dim3(256), lds, s, e0, e1, src, a, chain);
}
template void launch_walk_hash<double>(hipStream_t, const WalkHashArgs&, const ProductSrc<double>&, const Chain3&, hipEvent_t, hipEvent_t);
template void launch_walk_hash<float>(hipStream_t, const WalkHashArgs&, const ProductSrc<float>&, const Chain3&, hipEvent_t, hipEvent_t);

// The small classes alone: the merged kernel above takes the register count of its
// hungriest body (5 waves per SIMD); these bodies need 70 VGPRs, and a launch of their own
// reaches 7 waves per SIMD -- what rows that are one short chain of dependent loads need.
template <typename T>
__global__ __launch_bounds__(256) void num_tiny_kernel(ProductSrc<T> src, const u32* a_ro, RowWork w,
                                                       u32* __restrict__ c_col, T* __restrict__ c_val,
                                                       ClassGrid cg)
{
    SPECK_POISON();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    src.rebase(a_ro);
    const u32 b = blockIdx.x;
    if (b < cg.first[5])
        num_escw_body<T, 64, 256>(smem, src, w, c_col, c_val, NUM_R64, b - cg.first[4], cg.first[5] - cg.first[4], cg.cnt[4]);
    else if (b < cg.first[6])
        num_escw_body<T, 32, 256>(smem, src, w, c_col, c_val, NUM_R32, b - cg.first[5], cg.first[6] - cg.first[5], cg.cnt[5]);
    else if (b < cg.first[7])
        num_hash_body<SubWave<32>, T, kNumW128Cap, 0, kNumW128MaxNnz, SORT_RANK, 256>(
            smem, src, w, c_col, c_val, NUM_W128, b - cg.first[6], cg.first[7] - cg.first[6], cg.cnt[6]);
    else if (b < cg.first[8])
        num_esc_body<T, 16, 256>(smem, src, w, c_col, c_val, NUM_G16, b - cg.first[7], cg.first[8] - cg.first[7], cg.cnt[7]);
    else if (b < cg.first[9])
        num_esc_body<T, 8, 256>(smem, src, w, c_col, c_val, NUM_G8, b - cg.first[8], cg.first[9] - cg.first[8], cg.cnt[8]);
    else
        num_direct_body<T, 256>(smem, src, w, c_col, c_val, b - cg.first[9], cg.first[10] - cg.first[9], cg.cnt[9]);
}

// ------------------------------------------------------------------ NUM_G
// Spill path for heavy rows whose column range would need many dense windows (role of the
// reference's global hash maps, include/HashMap.cuh:112-134 and spECK_HashSpGEMM.cuh:25-36; the
// reference hard-disables the numeric one, Multiply.cu:699-700, and falls back to multi-window
// dense rows instead).
// A hash table in global memory costs two L2 atomics per product, and random global atomics top
// out at ~20 G/s on this chip whatever the table size or scope (scripts/ubench/global_atomics.hip):
// 2 ms for the 39 M atomics of the webbase-like input.  So the products of such a row are not
// accumulated in global memory at all.  They are PARTITIONED by column into buckets of
// ~2-4 k products with plain stores, and every bucket is then reduced in LDS like a NUM_B8K row:
//   plan    : per row a FINE column grid (width 2^shift, up to kGCellsPerBucket cells per wanted bucket), pool
//             offsets (one workgroup)
//   count   : products per fine cell         -- column walk, LDS histogram, one global add per
//                                               (workgroup, cell)
//   offsets : cells are merged into buckets of ~kGBucketTarget products (bucket = products before
//             the cell / unit: balanced whatever the column skew), bucket starts and column spans
//   scatter : (col, a*b) to the buckets      -- per staged chunk: LDS histogram, ONE global
//                                               reservation per (chunk, bucket), LDS ranks
//   reduce  : per bucket LDS hash accumulate + two-level bitmap sort (dense column windows for a
//             bucket that outgrew the table), result to the second pool, distinct count
//   copy    : the reduced buckets of a row, in bucket order, are its sorted C row
// kGParts workgroups share the walk of one row; buckets are independent workgroups.
constexpr u32 kGParts = 16;         // workgroups sharing the product walk of one row
constexpr u32 kGMaxBuckets = 2048;  // per row: the LDS histograms of scatter
constexpr u32 kGMaxCells = 4096;    // per row: the LDS histogram of count, the cell -> bucket map
constexpr int kGWalkThreads = 256;

__global__ __launch_bounds__(1024) void num_spill_plan_kernel(RowWork w, int cls)
{
    SPECK_POISON();
    __shared__ u32 s_scan[1024 / 64 + 2];
    __shared__ u64 s_run_p;
    __shared__ u32 s_run_b, s_run_f;
    if (block_void(w.st->capacity_miss)) return;
    const u32 count = min(w.st->num.count[cls], w.m);
    if (threadIdx.x == 0) {
        s_run_p = 0;
        s_run_b = 0;
        s_run_f = 0;
    }
    __syncthreads();
    for (u32 i0 = 0; i0 < count; i0 += 1024) {
        const u32 i = i0 + threadIdx.x;
        u32 ops = 0, nb = 0, nf = 0, shift = 0, unit = 1;
        if (i < count) {
            const RowRec rec = *class_rec_at(w.recs, w.m, cls, i);
            ops = rec.ops;
            const u32 range_m1 = rec.cmax - rec.cmin;
            nb = (ops + kGBucketTarget - 1) / kGBucketTarget;
            nb = nb < 1u ? 1u : (nb > kGMaxBuckets ? kGMaxBuckets : nb);
            unit = (ops + nb - 1) / nb;  // products per bucket (kGBucketTarget unless nb was clamped)
            const u32 cells = min(kGMaxCells, kGCellsPerBucket * nb);
            while ((range_m1 >> shift) + 1u > cells) ++shift;
            nf = (range_m1 >> shift) + 1u;
        }
        u32 t_lo, t_hi, t_nb, t_nf;
        const u32 e_lo = block_exclusive_scan<1024>(ops & 0xFFFFu, s_scan, &t_lo);
        const u32 e_hi = block_exclusive_scan<1024>(ops >> 16, s_scan, &t_hi);
        const u32 e_nb = block_exclusive_scan<1024>(nb, s_scan, &t_nb);
        const u32 e_nf = block_exclusive_scan<1024>(nf, s_scan, &t_nf);
        if (i < count) {
            GRowPlan p;
            p.pbase = s_run_p + e_lo + (u64(e_hi) << 16);
            p.bbase = s_run_b + e_nb;
            p.nb = nb;
            p.fbase = s_run_f + e_nf;
            p.nf = nf;
            p.shift = shift;
            p.unit = unit;
            w.spill.plan[i] = p;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            s_run_p += t_lo + (u64(t_hi) << 16);
            s_run_b += t_nb;
            s_run_f += t_nf;
        }
        __syncthreads();
    }
}

// LDS of the two walking kernels: A-row staging (+ values) | scan scratch | owner windows | histograms
template <typename T, bool WITH_VALUES>
constexpr u32 num_spill_walk_lds()
{
    using G = Block<kGWalkThreads>;
    return kGWalkThreads * (WITH_VALUES ? (u32)sizeof(T) : 0u) +
           (2 * kGWalkThreads + kGWalkThreads / 64 + 2 + win_words<G>() +
            (WITH_VALUES ? 2u * kGMaxBuckets + kGMaxCells / 2 : kGMaxCells) + 3) / 4 * 16;
}

// my slice of the A row: whole staging chunks, so that the slices of a short row collapse into
// its first parts
__device__ __forceinline__ bool spill_slice(const RowRec& rec, u32& lo, u32& hi)
{
    const u32 len = rec.a1 - rec.a0;
    const u32 per = ((len + gridDim.y - 1) / gridDim.y + kGWalkThreads - 1) / kGWalkThreads * kGWalkThreads;
    const u64 lo64 = u64(rec.a0) + u64(blockIdx.y) * per;
    if (lo64 >= rec.a1) return false;
    lo = (u32)lo64;
    hi = (u32)min(u64(rec.a1), lo64 + per);
    return true;
}

template <typename T>
__global__ __launch_bounds__(kGWalkThreads) void num_spill_count_kernel(ProductSrc<T> src, const u32* a_ro,
                                                                        RowWork w, int cls)
{
    SPECK_POISON();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using G = Block<kGWalkThreads>;
    const G g;
    u32* m_incl = reinterpret_cast<u32*>(smem);
    u32* scratch = m_incl + 2 * kGWalkThreads;
    u32* win = scratch + kGWalkThreads / 64 + 2;
    u32* hist = win + win_words<G>();
    RowMeta<T> meta{m_incl, m_incl + kGWalkThreads, nullptr, win};
    if (block_void(w.st->capacity_miss)) return;
    src.rebase(a_ro);
    const u32 count = min(w.st->num.count[cls], w.m);
    for (u32 idx = blockIdx.x; idx < count; idx += gridDim.x) {
        const RowRec rec = *class_rec_at(w.recs, w.m, cls, idx);
        u32 lo, hi;
        if (!spill_slice(rec, lo, hi)) continue;
        const GRowPlan pl = w.spill.plan[idx];
        for (u32 f = threadIdx.x; f < pl.nf; f += kGWalkThreads) hist[f] = 0;
        __syncthreads();
        for_each_product<false>(g, src, lo, hi, meta, scratch,
                                [&](const u32(&c)[kBatch], const T(&)[kBatch], u32 n) {
#pragma unroll
                                    for (int u = 0; u < kBatch; ++u)
                                        if ((u32)u < n) atomicAdd(&hist[min((c[u] - rec.cmin) >> pl.shift, pl.nf - 1u)], 1u);
                                }, cls);
        for (u32 f = threadIdx.x; f < pl.nf; f += kGWalkThreads)
            if (hist[f]) atomicAdd(&w.spill.fcount[pl.fbase + f], hist[f]);
        __syncthreads();
    }
}

// table entries a bucket can need: its distinct columns are bounded by its products and by its column span
__device__ __forceinline__ u32 spill_bucket_need(u32 products, u32 c_lo, u32 c_hi)
{
    return products ? min(products, c_hi - c_lo + 1u) : 0u;
}

__global__ __launch_bounds__(256) void num_spill_offsets_kernel(RowWork w, int cls)
{
    SPECK_POISON();
    __shared__ u32 s_scan[256 / 64 + 2];
    __shared__ u32 s_map[256];
    __shared__ u32 s_bc[kGMaxBuckets];  // products per bucket of the row (this workgroup owns them all)
    __shared__ u32 s_lo[kGMaxBuckets], s_hi[kGMaxBuckets];  // column span of every bucket
    if (block_void(w.st->capacity_miss)) return;
    const u32 count = min(w.st->num.count[cls], w.m);
    for (u32 idx = blockIdx.x; idx < count; idx += gridDim.x) {
        const GRowPlan pl = w.spill.plan[idx];
        const RowRec rec = *class_rec_at(w.recs, w.m, cls, idx);
        for (u32 b = threadIdx.x; b < pl.nb; b += 256) s_bc[b] = 0;
        __syncthreads();
        u32 run = 0;                 // products before the current chunk of cells
        u32 prev_last = 0xFFFFFFFFu;  // bucket of the last cell of the previous chunk
        for (u32 f0 = 0; f0 < pl.nf; f0 += 256) {
            const u32 f = f0 + threadIdx.x;
            const u32 v = f < pl.nf ? w.spill.fcount[pl.fbase + f] : 0u;
            u32 total;
            const u32 ex = run + block_exclusive_scan<256>(v, s_scan, &total);
            // bucket of a cell = products before it / unit: monotone in f, every bucket gets
            // < unit + (its last cell) products
            const u32 b = min(ex / pl.unit, pl.nb - 1u);
            s_map[threadIdx.x] = b;
            __syncthreads();
            if (f < pl.nf) {
                w.spill.fmap[pl.fbase + f] = b;
                if (v) atomicAdd(&s_bc[b], v);
                const u32 before = threadIdx.x ? s_map[threadIdx.x - 1] : prev_last;
                if (before != b) {  // first cell of bucket b (and the end of the bucket before)
                    w.spill.bstart[pl.bbase + b] = pl.pbase + ex;
                    w.spill.clo[pl.bbase + b] = s_lo[b] = rec.cmin + (f << pl.shift);
                    if (before != 0xFFFFFFFFu) w.spill.chi[pl.bbase + before] = s_hi[before] = rec.cmin + (f << pl.shift) - 1u;
                }
                if (f == pl.nf - 1) w.spill.chi[pl.bbase + b] = s_hi[b] = rec.cmax;
            }
            prev_last = s_map[255];
            run += total;
            __syncthreads();
        }
        // the buckets that outgrew the small table go on a work list: the launch with the big
        // table runs over that list only (its workgroups own most of a CU's LDS; launching one per
        // bucket just to find it small costs more than the reduction itself)
        // (no device-scope fence anywhere here: on a multi-XCD part it writes the L2 back)
        // (what a bucket needs of the table is its DISTINCT columns: at most its products, at most its column
        //  span -- hub rows pile thousands of products on a few hundred columns)
        __syncthreads();
        u32 mine = 0;
        for (u32 b = threadIdx.x; b < pl.nb; b += 256) {
            w.spill.bcount[pl.bbase + b] = s_bc[b];
            mine += spill_bucket_need(s_bc[b], s_lo[b], s_hi[b]) > kNumB2KMaxNnz ? 1u : 0u;
        }
        u32 n_big;
        u32 at = block_exclusive_scan<256>(mine, s_scan, &n_big);
        if (n_big) {  // one reservation per row: same-address atomics serialise (~12 ns each)
            if (threadIdx.x == 0) s_map[0] = atomicAdd(w.spill.big_count, n_big);
            __syncthreads();
            at += s_map[0];
            for (u32 b = threadIdx.x; b < pl.nb; b += 256)
                if (spill_bucket_need(s_bc[b], s_lo[b], s_hi[b]) > kNumB2KMaxNnz) w.spill.big_list[at++] = (u64(idx) << 32) | b;
        }
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(kGWalkThreads) void num_spill_scatter_kernel(ProductSrc<T> src, const u32* a_ro,
                                                                          RowWork w, int cls)
{
    SPECK_POISON();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using G = Block<kGWalkThreads>;
    const G g;
    T* m_av = reinterpret_cast<T*>(smem);
    u32* m_incl = reinterpret_cast<u32*>(m_av + kGWalkThreads);
    u32* scratch = m_incl + 2 * kGWalkThreads;
    u32* win = scratch + kGWalkThreads / 64 + 2;
    u32* hist = win + win_words<G>();
    u32* lbase = hist + kGMaxBuckets;
    unsigned short* cell2b = reinterpret_cast<unsigned short*>(lbase + kGMaxBuckets);
    RowMeta<T> meta{m_incl, m_incl + kGWalkThreads, m_av, win};
    if (block_void(w.st->capacity_miss)) return;
    src.rebase(a_ro);
    u32* pcol = w.spill.pcol[0];
    T* pval = static_cast<T*>(w.spill.pval[0]);
    const u32 count = min(w.st->num.count[cls], w.m);
    for (u32 idx = blockIdx.x; idx < count; idx += gridDim.x) {
        const RowRec rec = *class_rec_at(w.recs, w.m, cls, idx);
        u32 lo, hi;
        if (!spill_slice(rec, lo, hi)) continue;
        const GRowPlan pl = w.spill.plan[idx];
        __syncthreads();
        for (u32 f = threadIdx.x; f < pl.nf; f += kGWalkThreads) cell2b[f] = (unsigned short)w.spill.fmap[pl.fbase + f];
        for (u32 c0 = lo; c0 < hi; c0 += kGWalkThreads) {
            const u32 c1 = min(hi, c0 + (u32)kGWalkThreads);
            for (u32 b = threadIdx.x; b < pl.nb; b += kGWalkThreads) hist[b] = 0;
            __syncthreads();
            // (a) products of this chunk per bucket
            for_each_product<false>(g, src, c0, c1, meta, scratch,
                                    [&](const u32(&c)[kBatch], const T(&)[kBatch], u32 n) {
#pragma unroll
                                        for (int u = 0; u < kBatch; ++u)
                                            if ((u32)u < n)
                                                atomicAdd(&hist[cell2b[min((c[u] - rec.cmin) >> pl.shift, pl.nf - 1u)]], 1u);
                                    }, cls);
            // (b) one reservation per touched bucket; hist becomes the chunk-local fill count
            for (u32 b = threadIdx.x; b < pl.nb; b += kGWalkThreads) {
                const u32 cnt = hist[b];
                if (cnt) {
                    const u32 at = atomicAdd(&w.spill.bcursor[pl.bbase + b], cnt);
                    lbase[b] = (u32)(w.spill.bstart[pl.bbase + b] - pl.pbase) + at;
                    hist[b] = 0;
                }
            }
            __syncthreads();
            // (c) place (col, a*b)
            for_each_product<true>(g, src, c0, c1, meta, scratch,
                                   [&](const u32(&c)[kBatch], const T(&p)[kBatch], u32 n) {
#pragma unroll
                                       for (int u = 0; u < kBatch; ++u)
                                           if ((u32)u < n) {
                                               const u32 b = cell2b[min((c[u] - rec.cmin) >> pl.shift, pl.nf - 1u)];
                                               const u64 pos = pl.pbase + lbase[b] + atomicAdd(&hist[b], 1u);
                                               pcol[pos] = c[u];
                                               pval[pos] = p[u];
                                           }
                                   }, cls);
        }
    }
}

// reduce: two launches over all buckets.  The buckets aim at ~1 k products, so most of them fit
// the table of a NUM_B2K group (256 threads, 6 workgroups per CU); skewed columns fill some
// buckets beyond that: those take the NUM_B8K-sized launch, whose dense fallback handles anything.
// LDS: table | scan scratch; the dense fallback window aliases the table.
template <typename T, u32 CAP, int THREADS>
constexpr u32 num_spill_reduce_lds()
{
    return CAP * ((u32)sizeof(Acc<T>) + 4u) + (THREADS / 64 + 2 + 3) / 4 * 16;
}

// handles the buckets with N_LO < products <= N_HI by hashing (N_HI <= 2/3 CAP), and, if DENSE, the
// larger ones by dense column windows
template <typename T, u32 CAP, u32 W1, int THREADS, u32 N_LO, u32 N_HI, bool DENSE>
__global__ __launch_bounds__(THREADS) void num_spill_reduce_kernel(RowWork w, int cls)
{
    SPECK_POISON();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kGReduceThreads = THREADS;
    constexpr u32 kGDenseCols = CAP;
    using G = Block<THREADS>;
    const G g;
    Acc<T>* vals = reinterpret_cast<Acc<T>*>(smem);
    u32* keys = reinterpret_cast<u32*>(vals + CAP);
    u32* scratch = keys + CAP;
    u32* S = reinterpret_cast<u32*>(smem);
    static_assert(kGDenseCols * sizeof(Acc<T>) + 2 * (kGDenseCols / 32) * 4 <= CAP * (sizeof(Acc<T>) + 4),
                  "the dense fallback window aliases the table");
    static_assert(u64(N_HI) * 100 <= u64(CAP) * 85, "load factor <= 0.85");
    if (block_void(w.st->capacity_miss)) return;
    const u32* pcol = w.spill.pcol[0];
    const T* pval = static_cast<const T*>(w.spill.pval[0]);
    u32* ocol = w.spill.pcol[1];
    T* oval = static_cast<T*>(w.spill.pval[1]);
    const u32 count = w.st->num.count[cls];
    // the small launch walks (row, bucket) pairs; the big one the work list of oversized buckets
    const u32 outer_n = DENSE ? *w.spill.big_count : count;
    for (u32 it = blockIdx.x; it < outer_n; it += gridDim.x) {
        const u32 idx = DENSE ? (u32)(w.spill.big_list[it] >> 32) : it;
        const GRowPlan pl = w.spill.plan[idx];
        const u32 b_first = DENSE ? (u32)w.spill.big_list[it] : blockIdx.y;
        const u32 b_step = DENSE ? 0xFFFFFFFFu - b_first : gridDim.y;  // big: exactly one bucket
        for (u32 b = b_first; b < pl.nb; b += b_step) {
            const u32 n = w.spill.bcount[pl.bbase + b];
            if (n == 0) continue;  // empty (dcount stays 0)
            const u64 s0 = w.spill.bstart[pl.bbase + b];
            const u32 c_lo = w.spill.clo[pl.bbase + b], c_hi = w.spill.chi[pl.bbase + b];
            const u32 need = spill_bucket_need(n, c_lo, c_hi);
            if (need <= N_LO || (!DENSE && need > N_HI)) continue;  // the other launch's
            u32 distinct = 0;
            if (need <= N_HI) {
                u32 bits = table_bits(need, SPECK_LOAD_PCT);
                bits = min(max(bits, (u32)__builtin_ctz(kGReduceThreads)), (u32)__builtin_ctz(CAP));
                bits = (u32)__builtin_amdgcn_readfirstlane((int)bits);
                const u32 cap_row = 1u << bits;
                for (u32 i = threadIdx.x; i < cap_row; i += kGReduceThreads) {
                    keys[i] = kEmptyKey;
                    vals[i] = 0;
                }
                __syncthreads();
                for (u32 i0 = threadIdx.x; i0 < n; i0 += kBatch * kGReduceThreads) {
                    u32 c[kBatch];
                    Acc<T> p[kBatch];
                    u32 nv = 0;
#pragma unroll
                    for (int u = 0; u < kBatch; ++u) {
                        const u32 i = i0 + u * kGReduceThreads;
                        c[u] = kEmptyKey;
                        p[u] = 0;
                        if (i < n) {
                            c[u] = pcol[s0 + i];
                            p[u] = pval[s0 + i];
                            ++nv;
                        }
                    }
                    // (bounded: a bucket's table is sized by its products and its column span -- a column outside the range
                    //  the row's record holds, clamped into the last cell above, is not covered by that span)
                    (void)table_accumulate_batch<Acc<T>, true>(keys, vals, bits, c, p, nv);
                }
                __syncthreads();
                distinct = emit_bitmap_sorted<G, T, CAP, W1, N_HI>(g, keys, vals, S, scratch, cap_row, c_lo, c_hi,
                                                                   0u, ocol + s0, oval + s0, cls);
            } else if constexpr (DENSE) {
                // the bucket outgrew the table (skewed columns): dense windows over its span, the
                // products are re-read once per window
                Acc<T>* dv = reinterpret_cast<Acc<T>*>(smem);
                u32* bm = reinterpret_cast<u32*>(dv + kGDenseCols);
                u32* pref = bm + kGDenseCols / 32;
                for (u64 w0 = c_lo; w0 <= c_hi; w0 += kGDenseCols) {
                    const u64 left = u64(c_hi) - w0 + 1;
                    const u32 ncols = left < kGDenseCols ? (u32)left : kGDenseCols;
                    const u32 nwords = (ncols + 31) >> 5;
                    const u32 wbase = (u32)w0;
                    for (u32 i = threadIdx.x; i < ncols; i += kGReduceThreads) dv[i] = 0;
                    for (u32 i = threadIdx.x; i < nwords; i += kGReduceThreads) bm[i] = 0;
                    __syncthreads();
                    for (u32 i = threadIdx.x; i < n; i += kGReduceThreads) {
                        const u32 d = pcol[s0 + i] - wbase;
                        if (d < ncols) {
                            atomicAdd(&dv[d], (Acc<T>)pval[s0 + i]);
                            atomicOr(&bm[d >> 5], 1u << (d & 31));
                        }
                    }
                    __syncthreads();
                    const u32 total = bitmap_prefix(g, bm, pref, nwords, scratch);
                    for (u32 d = threadIdx.x; d < ncols; d += kGReduceThreads) {
                        const u32 word = bm[d >> 5];
                        if (word & (1u << (d & 31))) {
                            const u32 r = distinct + pref[d >> 5] + __popc(word & ((1u << (d & 31)) - 1u));
                            ocol[s0 + r] = wbase + d;
                            oval[s0 + r] = (T)dv[d];
                        }
                    }
                    distinct += total;
                    __syncthreads();
                }
            }
            if (threadIdx.x == 0) w.spill.dcount[pl.bbase + b] = distinct;
            __syncthreads();
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void num_spill_copy_kernel(RowWork w, u32* __restrict__ c_col,
                                                             T* __restrict__ c_val, int cls)
{
    SPECK_POISON();
    __shared__ u32 s_red[256 / 64 + 2];
    using G = Block<256>;
    const G g;
    if (block_void(w.st->capacity_miss)) return;
    const u32* ocol = w.spill.pcol[1];
    const T* oval = static_cast<const T*>(w.spill.pval[1]);
    const u32 count = min(w.st->num.count[cls], w.m);
    for (u32 idx = blockIdx.x; idx < count; idx += gridDim.x) {
        const RowRec rec = *class_rec_at(w.recs, w.m, cls, idx);
        const GRowPlan pl = w.spill.plan[idx];
        for (u32 b = blockIdx.y; b < pl.nb; b += gridDim.y) {
            const u32 n = w.spill.dcount[pl.bbase + b];
            if (n == 0) continue;  // uniform for the workgroup
            u32 before = 0;
            for (u32 j = threadIdx.x; j < b; j += 256) before += w.spill.dcount[pl.bbase + j];
            before = g.reduce_add(before, s_red);
            const u64 s0 = w.spill.bstart[pl.bbase + b];
            const size_t dst = size_t(rec.base) + before;
            // (a sequence without a symbolic pass: the row's room is what the previous identical call found -- nothing
            //  beyond it; the workgroup of the row's last bucket with entries compares the total below)
            const u32 fits = w.verify_numeric ? min(n, rec.nnz - min(rec.nnz, before)) : n;
            for (u32 i = threadIdx.x; i < fits; i += 256) {
                c_col[dst + i] = ocol[s0 + i];
                c_val[dst + i] = oval[s0 + i];
            }
        }
        if (w.verify_numeric && blockIdx.y == 0) {  // (uniform) the entries the buckets of the row hold NOW against its room
            u32 total = 0;
            for (u32 j = threadIdx.x; j < pl.nb; j += 256) total += w.spill.dcount[pl.bbase + j];
            total = g.reduce_add(total, s_red);
            if (threadIdx.x == 0 && total != rec.nnz) const_cast<DeviceStats*>(w.st)->capacity_miss = 1;
        }
    }
}

// ------------------------------------------------------------------ launchers

template <typename T>
u32 numeric_lds_bytes_t(int cls)
{
    switch (cls) {
        case NUM_DIRECT: return num_direct_lds<T, 256>();
        case NUM_G8: return 32 * num_esc_group_lds<T, 8>();
        case NUM_G16: return 16 * num_esc_group_lds<T, 16>();
        case NUM_R32: return 8 * num_escw_group_lds<T, 32>();
        case NUM_R64: return 4 * num_escw_group_lds<T, 64>();
        case NUM_W128: return 8 * num_group_lds<SubWave<32>, T, kNumW128Cap, 256>();
        case NUM_W512: return 4 * num_group_lds<SubWave<64>, T, kNumW512Cap, 256>();
        case NUM_W256: return 8 * num_group_lds<SubWave<32>, T, kNumW256Cap, 256>();
        case NUM_B2K: return num_group_lds<Block<256>, T, kNumB2KCap, 256>();
        case NUM_B8K: return num_group_lds<Block<512>, T, kNumB8KCap, 512>();
        case NUM_D1: return num_dense_lds<T, kNumD1Win, 256>();
        case NUM_D2: return num_dense_lds<T, kNumD2Cols, 1024>();
        case NUM_G: return num_spill_reduce_lds<T, kNumB8KCap, 512>();
        case NUM_NFCOPY: return 0;
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
        note_launch_status(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes),
                           "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
}

template <class G, typename T, u32 CAP, u32 W1, u32 NMAX, int MODE, int THREADS, u32 NLO = 0, bool VERIFY = false>
static void launch_num_hash(hipStream_t s, int cls, u32 count, const ProductSrc<T>& A, const u32* B,
                            const RowWork& w, u32* c_col, T* c_val, int cu_count)
{
    auto k = num_hash_kernel<G, T, CAP, W1, NMAX, MODE, THREADS, NLO, VERIFY>;
    const u32 lds = (THREADS / G::SIZE) * num_group_lds<G, T, CAP, THREADS>();
    set_dyn_lds(k, lds);
    SPECK_LAUNCH(k, dim3(grid_for(count, lds, THREADS, cu_count, THREADS / G::SIZE)),
                       dim3(THREADS), lds, s, A, B, w, c_col, c_val, cls);
}

static u32 g_b8k_full_first = 1;  // bit 0: complete calls, bit 1: the verifying launches of a reuse sequence
void set_b8k_full_first(u32 mask) { g_b8k_full_first = mask; }
static u32 g_spill_big_grid = 256;
void set_spill_big_grid(u32 blocks) { g_spill_big_grid = blocks ? blocks : 256u; }

template <typename T>
void launch_numeric_light(hipStream_t s, const u32* counts_hint, u32 mask, const CsrView<T>& Av,
                          const CsrView<T>& Bv, const RowWork& w, u32* c_col, T* c_val, int cu_count, bool exact,
                          hipEvent_t e0, hipEvent_t e1)
{
    constexpr int NS = 10;
    static const int slots[NS] = {NUM_D1, NUM_B2K, NUM_W512, NUM_W256, NUM_R64, NUM_R32, NUM_W128, NUM_G16, NUM_G8, NUM_DIRECT};
    static const u32 rows_per_block[NS] = {1, 1, 4, 8, 4, 8, 8, 16, 32, 256};
    bool tiny_only = true;
    for (int k = 0; k < 4; ++k)
        if ((mask >> slots[k] & 1u) && counts_hint[slots[k]]) tiny_only = false;
    if (w.verify_numeric || w.off_n) tiny_only = false;  // (only the big kernel has verifying bodies / moves the offsets)
    u32 lds = 0;
    for (int k = 0; k < NS; ++k)
        if (mask >> slots[k] & 1u) lds = std::max(lds, numeric_lds_bytes_t<T>(slots[k]));
    ClassGrid cg{};
    for (int k = 0; k < NS; ++k) {
        const bool on = (mask >> slots[k] & 1u) && counts_hint[slots[k]];
        cg.first[k + 1] = cg.first[k] + (on ? grid_for(counts_hint[slots[k]], lds, 256, cu_count, rows_per_block[k]) : 0u);
        cg.cnt[k] = exact ? counts_hint[slots[k]] : kNoCount;
    }
    // ... + the workgroups that move the staged row offsets (only the big kernel carries them)
    cg.first[NS + 1] = cg.first[NS];
    if (!tiny_only && w.off_n) cg.first[NS + 1] += std::min<u32>((w.off_n + 2047u) / 2048u, 512u);
    if (cg.first[NS + 1] == 0) {
        if (e0) (void)hipEventRecord(e0, s), (void)hipEventRecord(e1, s);  // (nothing to time: an empty interval)
        return;
    }
    const ProductSrc<T> src{w.b_sl, Av.data, Bv.col_ids, Bv.data, w.w_sl};
    bool with_esc = false;  // (a class of the mask without rows has no blocks: its body is never entered)
    for (int k = 0; k < NS; ++k)
        if ((kNumEscMask >> slots[k] & 1u) && cg.first[k + 1] != cg.first[k]) with_esc = true;
    if (w.verify_numeric)  // (pipeline.hip plans such a sequence only when this launch has workgroup / wave classes and no
                           //  register-class rows of its own)
        SPECK_LAUNCH_TIMED((num_light_kernel<T, false, true>), dim3(cg.first[NS + 1]), dim3(256), lds, s, e0, e1, src,
                           Av.row_offsets, w, c_col, c_val, cg);
    else if (!tiny_only && with_esc)
        SPECK_LAUNCH_TIMED((num_light_kernel<T, true>), dim3(cg.first[NS + 1]), dim3(256), lds, s, e0, e1, src, Av.row_offsets, w,
                           c_col, c_val, cg);
    else if (!tiny_only)
        SPECK_LAUNCH_TIMED((num_light_kernel<T, false>), dim3(cg.first[NS + 1]), dim3(256), lds, s, e0, e1, src, Av.row_offsets, w,
                           c_col, c_val, cg);
    else
        SPECK_LAUNCH_TIMED((num_tiny_kernel<T>), dim3(cg.first[NS]), dim3(256), lds, s, e0, e1, src, Av.row_offsets, w,
                           c_col, c_val, cg);
}

template <typename T>
void launch_numeric_first(hipStream_t s, u32 count, const CsrView<T>& Av, const CsrView<T>& Bv, const RowWork& w,
                          u32* counts, int cu_count, u32 wcols, hipEvent_t e0, hipEvent_t e1)
{
    if (count == 0) {
        if (e0) (void)hipEventRecord(e0, s), (void)hipEventRecord(e1, s);
        return;
    }
    const ProductSrc<T> src{w.b_sl, Av.data, Bv.col_ids, Bv.data, w.w_sl};
    wcols = wcols < 256u ? 256u : (wcols > kNumD1Cols ? kNumD1Cols : (wcols + 255u) & ~255u);
    const u32 lds = (wcols + 256) * (u32)sizeof(Acc<T>) + wcols +
                    (2 * 256 + 256 / 64 + 2 + win_words<Block<256>>() + 3) / 4 * 16;
    if (w.nf_pred_off) {
        set_dyn_lds((nf_dense_kernel<T, 256, true>), lds);
        SPECK_LAUNCH_TIMED((nf_dense_kernel<T, 256, true>), dim3(grid_for(count, lds, 256, cu_count, 1)), dim3(256), lds, s, e0,
                           e1, src, Av.row_offsets, w, counts, wcols);
        return;
    }
    set_dyn_lds((nf_dense_kernel<T, 256>), lds);
    SPECK_LAUNCH_TIMED((nf_dense_kernel<T, 256>), dim3(grid_for(count, lds, 256, cu_count, 1)), dim3(256), lds, s, e0, e1,
                       src, Av.row_offsets, w, counts, wcols);
}

template <typename T>
void launch_numeric(hipStream_t s, int cls, u32 count, const CsrView<T>& Av, const CsrView<T>& Bv,
                    const RowWork& w, u32* c_col, T* c_val, int cu_count)
{
    if (count == 0) return;
    // (A, B) below = (product source, A.row_offsets): the kernels rebase the per-entry arrays
    const ProductSrc<T> A{w.b_sl, Av.data, Bv.col_ids, Bv.data, w.w_sl};
    const u32* B = Av.row_offsets;
    const u32 lds = numeric_lds_bytes_t<T>(cls);
    switch (cls) {
        case NUM_DIRECT: {
            constexpr int TH = 256;
            SPECK_LAUNCH((num_direct_kernel<T, TH>), dim3(grid_for(count, lds, TH, cu_count, TH)),
                               dim3(TH), lds, s, A, B, w, c_col, c_val);
            break;
        }
        case NUM_G8:
            SPECK_LAUNCH((num_esc_kernel<T, 8>), dim3(grid_for(count, lds, 256, cu_count, 32)), dim3(256), lds, s, A, B,
                               w, c_col, c_val, cls);
            break;
        case NUM_G16:
            SPECK_LAUNCH((num_esc_kernel<T, 16>), dim3(grid_for(count, lds, 256, cu_count, 16)), dim3(256), lds, s, A, B,
                               w, c_col, c_val, cls);
            break;
        case NUM_R32:
            SPECK_LAUNCH((num_escw_kernel<T, 32>), dim3(grid_for(count, lds, 256, cu_count, 8)), dim3(256), lds, s, A, B,
                               w, c_col, c_val, cls);
            break;
        case NUM_R64:
            SPECK_LAUNCH((num_escw_kernel<T, 64>), dim3(grid_for(count, lds, 256, cu_count, 4)), dim3(256), lds, s, A, B,
                               w, c_col, c_val, cls);
            break;
        case NUM_W128:
            launch_num_hash<SubWave<32>, T, kNumW128Cap, 0, kNumW128MaxNnz, SORT_RANK, 256>(
                s, cls, count, A, B, w, c_col, c_val, cu_count);
            break;
        case NUM_W512:
            launch_num_hash<SubWave<64>, T, kNumW512Cap, kW512W1, kNumW512MaxNnz, SORT_BITMAP, 256>(
                s, cls, count, A, B, w, c_col, c_val, cu_count);
            break;
        case NUM_W256:
            launch_num_hash<SubWave<32>, T, kNumW256Cap, kW256W1, kNumW256MaxNnz, SORT_BITMAP, 256>(
                s, cls, count, A, B, w, c_col, c_val, cu_count);
            break;
        case NUM_B2K:
            launch_num_hash<Block<256>, T, kNumB2KCap, kB2KW1, kNumB2KMaxNnz, SORT_BITMAP, 256>(
                s, cls, count, A, B, w, c_col, c_val, cu_count);
            break;
        case NUM_B8K:
            // two launches over the class: the rows of the lower half fit a half-size table, whose
            // workgroups run two per CU (the full table owns 106 of a CU's 160 KiB); the full table goes first
            // (a handful of rows cannot fill the CUs anyway: one launch, one row's latency less)
            if (w.sliced) {  // in column slices of the 2 Ki table (num_sliced_body)
                const u32 sl = num_sliced_lds(num_group_lds<Block<256>, T, kNumB2KCap, 256>());
                if (w.verify_numeric)
                    SPECK_LAUNCH((num_sliced_kernel<T, true>), dim3(grid_for(count, sl, 256, cu_count, 1)), dim3(256), sl, s, A, B, w,
                                 c_col, c_val, cls);
                else
                    SPECK_LAUNCH((num_sliced_kernel<T, false>), dim3(grid_for(count, sl, 256, cu_count, 1)), dim3(256), sl, s, A, B, w,
                                 c_col, c_val, cls);
                break;
            }
            if (w.verify_numeric) {  // (a sequence without a symbolic pass: the verifying forms of the same launches)
                if (count * 2 < (u32)cu_count) {
                    launch_num_hash<Block<512>, T, kNumB8KCap, kB8KW1, kNumB8KMaxNnz, SORT_BITMAP, 512, 0, true>(
                        s, cls, count, A, B, w, c_col, c_val, cu_count);
                    break;
                }
                if (g_b8k_full_first & 2u)
                    launch_num_hash<Block<512>, T, kNumB8KCap, kB8KW1, kNumB8KMaxNnz, SORT_BITMAP, 512, kNumB8KHalfMaxNnz, true>(
                        s, cls, count, A, B, w, c_col, c_val, cu_count);
                launch_num_hash<Block<512>, T, kNumB8KCap / 2, kB8KW1, kNumB8KHalfMaxNnz, SORT_BITMAP, 512, 0, true>(
                    s, cls, count, A, B, w, c_col, c_val, cu_count);
                if (!(g_b8k_full_first & 2u))
                    launch_num_hash<Block<512>, T, kNumB8KCap, kB8KW1, kNumB8KMaxNnz, SORT_BITMAP, 512, kNumB8KHalfMaxNnz, true>(
                        s, cls, count, A, B, w, c_col, c_val, cu_count);
                break;
            }
            if (count * 2 < (u32)cu_count) {
                launch_num_hash<Block<512>, T, kNumB8KCap, kB8KW1, kNumB8KMaxNnz, SORT_BITMAP, 512>(
                    s, cls, count, A, B, w, c_col, c_val, cu_count);
                break;
            }
            // (the full table first: a workgroup of 106 KiB finds room on a CU only while the light launch beside it has
            //  not filled the chip -- webbase stand-in 1.176 -> 1.158 ms complete, three alternating pairs; the verifying
            //  launches of a reuse sequence keep the half table first: 0.788 against 0.868 ms.  Option b8k_full_first)
            if (g_b8k_full_first & 1u)
                launch_num_hash<Block<512>, T, kNumB8KCap, kB8KW1, kNumB8KMaxNnz, SORT_BITMAP, 512, kNumB8KHalfMaxNnz>(
                    s, cls, count, A, B, w, c_col, c_val, cu_count);
            launch_num_hash<Block<512>, T, kNumB8KCap / 2, kB8KW1, kNumB8KHalfMaxNnz, SORT_BITMAP, 512>(
                s, cls, count, A, B, w, c_col, c_val, cu_count);
            if (!(g_b8k_full_first & 1u))
                launch_num_hash<Block<512>, T, kNumB8KCap, kB8KW1, kNumB8KMaxNnz, SORT_BITMAP, 512, kNumB8KHalfMaxNnz>(
                    s, cls, count, A, B, w, c_col, c_val, cu_count);
            break;
        case NUM_D1: {
            auto k = num_dense_kernel<T, kNumD1Win, 256>;
            set_dyn_lds(k, lds);
            SPECK_LAUNCH(k, dim3(grid_for(count, lds, 256, cu_count, 1)), dim3(256), lds, s, A, B, w,
                               c_col, c_val, cls);
            break;
        }
        case NUM_D2: {
            if (w.verify_numeric) {
                auto kv = num_dense_kernel<T, kNumD2Cols, 1024, true>;
                set_dyn_lds(kv, lds);
                SPECK_LAUNCH(kv, dim3(grid_for(count, lds, 1024, cu_count, 1)), dim3(1024), lds, s, A, B,
                                   w, c_col, c_val, cls);
                break;
            }
            auto k = num_dense_kernel<T, kNumD2Cols, 1024>;
            set_dyn_lds(k, lds);
            SPECK_LAUNCH(k, dim3(grid_for(count, lds, 1024, cu_count, 1)), dim3(1024), lds, s, A, B,
                               w, c_col, c_val, cls);
            break;
        }
        case NUM_NFCOPY: {
            u32 blocks = (count + 3) / 4;  // four waves per workgroup, a row per wave
            const u32 cap = (u32)cu_count * 8u * 4u;
            if (blocks > cap) blocks = cap;
            SPECK_LAUNCH((nf_copy_kernel<T>), dim3(blocks ? blocks : 1), dim3(256), 0, s, w, c_col, c_val);
            break;
        }
        case NUM_G: {
            // same stream: the kernel boundaries are the grid-wide barriers between the steps
            const u32 rows = count < 8192u ? (count ? count : 1u) : 8192u;
            (void)hipMemsetAsync(w.spill.fcount, 0,
                                 (size_t(w.spill.cell_cap) + 3 * size_t(w.spill.bucket_cap) + 4) * sizeof(u32), s);
            SPECK_LAUNCH(num_spill_plan_kernel, dim3(1), dim3(1024), 0, s, w, cls);
            auto kc = num_spill_count_kernel<T>;
            SPECK_LAUNCH(kc, dim3(rows, kGParts), dim3(kGWalkThreads), (num_spill_walk_lds<T, false>()), s,
                               A, B, w, cls);
            SPECK_LAUNCH(num_spill_offsets_kernel, dim3(rows), dim3(256), 0, s, w, cls);
            auto ks = num_spill_scatter_kernel<T>;
            SPECK_LAUNCH(ks, dim3(rows, kGParts), dim3(kGWalkThreads), (num_spill_walk_lds<T, true>()), s,
                               A, B, w, cls);
            auto kr_small = num_spill_reduce_kernel<T, kNumB2KCap, kB2KW1, 256, 0, kNumB2KMaxNnz, false>;
            SPECK_LAUNCH(kr_small, dim3(rows, 32), dim3(256), (num_spill_reduce_lds<T, kNumB2KCap, 256>()), s, w,
                               cls);
            // (the big table takes a bucket up to a load of 0.85: beyond it only the dense windows are left, and a bucket
            //  that misses the 2/3 mark by a few entries would pay dozens of them over its column span)
            auto kr_big = num_spill_reduce_kernel<T, kNumB8KCap, kB8KW1, 512, kNumB2KMaxNnz, kNumB8KCap * 85 / 100, true>;
            set_dyn_lds(kr_big, lds);
            // (its workgroups stride over the list of oversized buckets, which only the device knows -- usually a handful.
            //  Each needs 106 KiB of LDS just to find that out: a grid of 2048 queued behind the NUM_B8K launches of the
            //  same phase for 0.3-0.5 ms on the webbase stand-in; option spill_big_grid)
            SPECK_LAUNCH(kr_big, dim3(g_spill_big_grid), dim3(512), lds, s, w, cls);
            SPECK_LAUNCH((num_spill_copy_kernel<T>), dim3(rows, 32), dim3(256), 0, s, w, c_col, c_val, cls);
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
                                           const CsrView<double>&, const RowWork&, u32*, double*, int, bool,
                                           hipEvent_t, hipEvent_t);
template void launch_numeric_light<float>(hipStream_t, const u32*, u32, const CsrView<float>&,
                                          const CsrView<float>&, const RowWork&, u32*, float*, int, bool, hipEvent_t,
                                          hipEvent_t);
template void launch_numeric_first<double>(hipStream_t, u32, const CsrView<double>&, const CsrView<double>&,
                                           const RowWork&, u32*, int, u32, hipEvent_t, hipEvent_t);
template void launch_numeric_first<float>(hipStream_t, u32, const CsrView<float>&, const CsrView<float>&,
                                          const RowWork&, u32*, int, u32, hipEvent_t, hipEvent_t);
template void launch_numeric<double>(hipStream_t, int, u32, const CsrView<double>&,
                                     const CsrView<double>&, const RowWork&, u32*, double*, int);
template void launch_numeric<float>(hipStream_t, int, u32, const CsrView<float>&,
                                    const CsrView<float>&, const RowWork&, u32*, float*, int);

}  // namespace speck
