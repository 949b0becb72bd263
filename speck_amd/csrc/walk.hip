// walk.hip -- the ONE-WALK complete call: rows of the register classes are walked ONCE, not counted and then computed.
//
// The reference walks every row of A twice by design -- a symbolic pass sizes C (source/GPU/Multiply.cu:488-575), the
// exclusive scan places the rows (:570), the numeric pass fills them (:835-1014) -- and so did every complete call of
// this library up to round 5; only a REUSE sequence, which knows where the previous identical call put every row,
// finished the rows of the register classes (esc.hpp, esc_wide.hpp: <= 256 products, sorted in registers, nothing sized
// by the row's nnz) in one walk.  What a row needs before it can be written is its PLACE in C, i.e. the nnz of all
// rows before it -- and that is what the look-back chain (chain.hpp) hands a workgroup inside ONE launch.  This kernel is
// the scan kernel (stages.hip: row offsets + numeric binning, a tile of contiguous rows per workgroup) with the numeric
// walk of the register-class rows inside:
//   0. a tile loads the per-row results of the analysis for its rows; rows of the register classes are listed by class
//   1. every wave takes items off that list -- one 64-lane row, two 32-lane rows, four 16-lane rows or eight 8-lane rows
//      -- and FINISHES them (num_esc_row / num_escw_row): the sorted, summed row goes to the row's slot of the scratch
//      pool (its slot offset is a prefix over the rows, written by the analysis like the slots of the numeric-first rows),
//      its nnz to LDS.  Rows of every other class were counted by their symbolic kernels before this launch.
//   2. the tile publishes (nnz, rows per numeric class, ...), takes the exclusive prefix over the tiles before it,
//   3. writes the row offsets, the numeric records of the rows other launches will compute, and MOVES the finished rows
//      from their slots to their places in C (also the numeric-first rows: no nf_copy launch).
// C must already be allocated: the caller hands over matOut's buffers and their capacity, every tile checks that what it
// places ends inside it, and a call whose nnz(C) outgrows the buffers is declared void (capacity_miss: the two-phase
// call re-runs and re-allocates -- the reference re-allocates exactly then, Multiply.cu:589-592).  Nothing of a previous
// call is read.
// Why slots and not registers: a row's place depends on every row before it IN THE TILE as well, so no row can leave
// before the tile's last row is done; a tile is tens of rows and an item is one dependent chain of gathers long, so
// the finished rows wait in memory (L2 / Infinity Cache: a tile's slots are written and read back by one workgroup
// within microseconds), not in the registers of waves that would sit idle.
#include <algorithm>
#include <type_traits>

#include "chain.hpp"
#include "device_common.hpp"
#include "launch.hpp"
#include "row_groups.hpp"
#include "esc.hpp"
#include "esc_rows.hpp"
#include "esc_wide.hpp"

namespace speck {

constexpr int kWalkThreads = 256, kWalkWaves = kWalkThreads / 64;
constexpr u32 kWalkStagedClasses = 5;  // list order: 64-lane rows, 32-lane, 16-lane, 8-lane rows, numeric-first rows

// LDS of one wave for the rows it finishes: the largest of the four classes' needs (8 x 8, 4 x 16, 2 x 32, 1 x 64 lanes)
template <typename T>
constexpr u32 walk_wave_lds()
{
    u32 b = 8u * num_esc_group_lds<T, 8>();
    b = b > 4u * num_esc_group_lds<T, 16>() ? b : 4u * num_esc_group_lds<T, 16>();
    b = b > 2u * num_escw_group_lds<T, 32>() ? b : 2u * num_escw_group_lds<T, 32>();
    b = b > num_escw_group_lds<T, 64>() ? b : num_escw_group_lds<T, 64>();
    return (b + 15u) / 16u * 16u;
}

// entries [0, n) of a finished row from its slot to its place in C, by the L lanes of a group (4 L entries in flight)
template <typename T, u32 L>
__device__ __forceinline__ void walk_move_row(u32 gl, const u32* __restrict__ s_col, const T* __restrict__ s_val,
                                              u32* __restrict__ d_col, T* __restrict__ d_val, u32 n)
{
    for (u32 j0 = gl; j0 < n; j0 += 4u * L) {
        u32 c[4];
        T v[4];
#pragma unroll
        for (u32 u = 0; u < 4; ++u) {
            const u32 j = j0 + u * L;
            c[u] = j < n ? s_col[j] : 0u;
            v[u] = j < n ? s_val[j] : T(0);
        }
#pragma unroll
        for (u32 u = 0; u < 4; ++u) {
            const u32 j = j0 + u * L;
            if (j < n) {
                d_col[j] = c[u];
                d_val[j] = v[u];
            }
        }
    }
}

extern __shared__ __attribute__((aligned(16))) unsigned char walk_lds[];  // kWalkWaves x walk_wave_lds<T>() bytes

template <typename T, int ITEMS>
__global__ __launch_bounds__(kWalkThreads, 7) void walk_kernel(ProductSrc<T> src, WalkArgs a, Chain chain)
{
    SPECK_POISON();
    constexpr u32 TRMAX = kWalkThreads * ITEMS;
    constexpr int NW = kWalkWaves;
    __shared__ u32 s_a0[TRMAX + 1], s_cmin[TRMAX], s_nnz[TRMAX], s_off[TRMAX];
    __shared__ u64 s_slot[TRMAX];
    __shared__ unsigned short s_list[TRMAX];
    __shared__ u32 s_cnt[ITEMS * NW][kWalkStagedClasses];   // staged rows per (sub-tile, wave, list class)
    __shared__ u32 s_start[kWalkStagedClasses + 1];
    __shared__ u32 s_mine[kChainWords];
    __shared__ u64 s_pref[kChainWords], s_tmp[2 * kChainWords + 2];
    __shared__ u32 s_scan[NW + 1];
    __shared__ u32 s_wcnt[ITEMS][NUM_CLASSES][NW];
    __shared__ u64 s_sum[NW], s_gops[NW];
    __shared__ u32 s_max[NW];
    __shared__ u64 s_bytes[kMaxClasses];
    __shared__ u32 s_run[kMaxClasses];
    const u32 t = threadIdx.x, lane = lane_id(), wid = t >> 6;
    src.rebase(a.a_ro);
    const u32 TR = a.tile_rows;  // <= TRMAX, a multiple of 64
    const u32 r0 = blockIdx.x * TR, nrows = min(TR, a.m - r0);
    // (a call an earlier kernel has declared void -- the pool does not hold the slots, a class nobody launched has rows --
    //  computes and moves nothing; its offsets still go through the chain so that the last tile can report)
    const bool void_call = block_void(a.st->capacity_miss);  // (one decision per workgroup: row_groups.hpp)
    T* const pool_val = static_cast<T*>(a.pool_val);
    T* const c_val = static_cast<T*>(a.c_val);

    // ---- 0. my rows: what the analysis found; the rows this tile finishes itself, listed by class
    u32 r_ops[ITEMS], r_cmax[ITEMS];
    u8 r_sym[ITEMS];
    if (t < kMaxClasses) s_bytes[t] = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const u32 lr = t + kWalkThreads * k;
        r_ops[k] = r_cmax[k] = 0;
        r_sym[k] = SYM_NONE;
        u32 lk = kWalkStagedClasses;  // list class of my row (none)
        if (lr < nrows) {
            const u32 row = r0 + lr;
            const u8 cls = a.cls_sym[row];
            r_sym[k] = cls;
            s_a0[lr] = a.a_ro[row];
            if (lr == nrows - 1) s_a0[nrows] = a.a_ro[row + 1];
            s_cmin[lr] = a.row_col_min[row];
            r_cmax[k] = a.row_col_max[row];
            r_ops[k] = a.row_ops[row];
            lk = cls == SYM_R64 ? 0u : cls == SYM_R32 ? 1u : cls == SYM_G16 ? 2u : cls == SYM_G8 ? 3u : cls == SYM_NF ? 4u : lk;
            s_slot[lr] = lk < kWalkStagedClasses ? a.nf_off[row] : 0ull;
            // rows of the other classes were counted before this launch; mine are counted as they are finished
            s_nnz[lr] = lk < 4u ? 0u : a.counts[row];
        }
        if (lr < TR) {  // (whole waves: TR is a multiple of 64)
#pragma unroll
            for (u32 q = 0; q < kWalkStagedClasses; ++q) {
                const u64 mask = __ballot(lk == q);
                if (lane == 0) s_cnt[k * NW + wid][q] = (u32)__popcll(mask);
            }
        } else if (lane == 0) {
#pragma unroll
            for (u32 q = 0; q < kWalkStagedClasses; ++q) s_cnt[k * NW + wid][q] = 0;
        }
    }
    __syncthreads();
    if (t == 0) {
        u32 run = 0;
        for (u32 q = 0; q < kWalkStagedClasses; ++q) {
            s_start[q] = run;
            for (int i = 0; i < ITEMS * NW; ++i) run += s_cnt[i][q];
        }
        s_start[kWalkStagedClasses] = run;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const u32 lr = t + kWalkThreads * k;
        if (lr >= TR) continue;
        const u8 cls = r_sym[k];
        const u32 lk = lr >= nrows ? kWalkStagedClasses
                                   : (cls == SYM_R64 ? 0u : cls == SYM_R32 ? 1u : cls == SYM_G16 ? 2u : cls == SYM_G8 ? 3u
                                                                                                  : cls == SYM_NF ? 4u : kWalkStagedClasses);
#pragma unroll
        for (u32 q = 0; q < kWalkStagedClasses; ++q) {
            const u64 mask = __ballot(lk == q);
            if (lk == q) {
                u32 pos = s_start[q] + (u32)__popcll(mask & lanemask_lt());
                for (int i = 0; i < k * NW + (int)wid; ++i) pos += s_cnt[i][q];
                s_list[pos] = (unsigned short)lr;
            }
        }
    }
    __syncthreads();

    // ---- 1. the rows of the register classes: finished here, into their slots
    const u32 n64 = s_start[1] - s_start[0], n32 = s_start[2] - s_start[1], n16 = s_start[3] - s_start[2],
              n8 = s_start[4] - s_start[3];
    const u32 it64 = n64, it32 = (n32 + 1u) >> 1, it16 = (n16 + 3u) >> 2, it8 = (n8 + 7u) >> 3;
    unsigned char* const my_stage = walk_lds + wid * walk_wave_lds<T>();
    // One loop per class (a wave takes every fourth item of each): with the four bodies in ONE loop the register allocator
    // kept the state of all of them alive side by side -- 127 VGPRs, four waves per SIMD; as functions of their own
    // (noinline) their LDS and global accesses became flat instructions (generic pointers), 139 us for the kernel.
    auto run_items = [&](u32 n_items, u32 cnt, u32 first, auto width_tag) {
        constexpr u32 W = decltype(width_tag)::value;
        constexpr u32 PER_WAVE = 64u / W;
        for (u32 it = wid; it < n_items; it += NW) {  // (wave-uniform)
            const u32 idx = it * PER_WAVE + lane / W;
            const bool active = idx < cnt;
            const u32 lr = active ? s_list[first + idx] : 0u;
            const u32 e0 = active ? s_a0[lr] : 0u, e1 = active ? s_a0[lr + 1] : 0u;
            const u64 slot = s_slot[lr];
            // (never past the end of the pool, whatever the analysis' total said)
            const u32 room = (active && slot < a.pool_cap) ? (u32)std::min<u64>(a.pool_cap - slot, 0xFFFFFFFFull) : 0u;
            u32* const o_col = a.pool_col + slot;
            T* const o_val = pool_val + slot;
            const SubWave<W> g;
            u32 all;
            if constexpr (W >= 32)
                all = num_escw_row<T, W>(g, my_stage + (lane / W) * num_escw_group_lds<T, W>(), src, e0, e1, s_cmin[lr], o_col, o_val, room);
            else
                all = num_esc_row<T, W>(g, my_stage + (lane / W) * num_esc_group_lds<T, W>(), src, e0, e1, o_col, o_val, room);
            if (active && (lane & (W - 1u)) == 0u) {
                s_nnz[lr] = all;
                if (all > room) const_cast<DeviceStats*>(a.st)->capacity_miss = 1;  // (cannot happen: nnz <= products <= slot)
            }
        }
    };
    if (!(void_call || (a.debug & 2u))) {
        run_items(it64, n64, s_start[0], std::integral_constant<u32, 64>{});
        run_items(it32, n32, s_start[1], std::integral_constant<u32, 32>{});
        run_items(it16, n16, s_start[2], std::integral_constant<u32, 16>{});
        run_items(it8, n8, s_start[3], std::integral_constant<u32, 8>{});
    }
    __syncthreads();

    // ---- 2. the tile's aggregate -> the chain -> what the tiles before me hold
    u32 c[ITEMS];
    u8 cls[ITEMS];
    u64 tsum = 0, g_ops = 0;
    u32 my_max = 0;
    // (rows of a class in my wave: ballots -- s_wcnt[sub-tile][class][wave], read again when the records are placed)
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const u32 lr = t + kWalkThreads * k;
        c[k] = 0;
        cls[k] = NUM_NONE;
        if (lr < nrows) {
            c[k] = s_nnz[lr];
            const u32 len_a = s_a0[lr + 1] - s_a0[lr];
            cls[k] = classify_numeric(len_a, r_ops[k], c[k], s_cmin[lr], r_cmax[k], a.cp);
            if (cls[k] == NUM_G) g_ops += r_ops[k];
            if (a.cp.want_bytes && cls[k] != NUM_NONE)
                atomicAdd((unsigned long long*)&s_bytes[cls[k]], (unsigned long long)numeric_row_bytes(len_a, r_ops[k], c[k], a.vsize));
            tsum += c[k];
            my_max = max(my_max, c[k]);
        }
#pragma unroll
        for (u32 q = 0; q < NUM_CLASSES; ++q) {
            const u32 n = (u32)__popcll(__ballot(cls[k] == q));
            if (lane == 0) s_wcnt[k][q][wid] = n;
        }
    }
    tsum = wave_reduce_add(tsum);
    g_ops = wave_reduce_add(g_ops);
    my_max = wave_reduce_max(my_max);
    if (lane == 0) {
        s_sum[wid] = tsum;
        s_gops[wid] = g_ops;
        s_max[wid] = my_max;
    }
    __syncthreads();
    if (t < kChainWords) {
        u32 v = 0;
        if (t < NUM_CLASSES)
            for (int k = 0; k < ITEMS; ++k)
                for (int w = 0; w < NW; ++w) v += s_wcnt[k][t][w];
        u64 sum = 0, gs = 0;
        u32 mx = 0;
        for (int w = 0; w < NW; ++w) {
            sum += s_sum[w];
            gs += s_gops[w];
            mx = max(mx, s_max[w]);
        }
        if (t == kCwTotLo) v = (u32)gs;
        if (t == kCwTotHi) v = (u32)(gs >> 32);
        if (t == kCwPfxLo) v = (u32)sum;
        if (t == kCwPfxHi) v = (u32)(sum >> 32);
        if (t == kCwMax) v = mx;
        s_mine[t] = v;
    }
    if (a.cp.want_bytes && a.bytes_acc && t < NUM_CLASSES && s_bytes[t])
        atomicAdd((unsigned long long*)&a.bytes_acc[kMaxClasses + t], (unsigned long long)s_bytes[t]);
    __syncthreads();
    const u32 nb = gridDim.x;
    chain_publish_own(chain, blockIdx.x, s_mine);
    const bool chain_ok = chain_exclusive(chain, blockIdx.x, nb, s_mine, s_pref, s_tmp);
    const u64 nnz_before = chain_u64(s_pref, kCwPfxLo, kCwPfxHi);
    const u64 tile_nnz = (u64(s_mine[kCwPfxHi]) << 32) + s_mine[kCwPfxLo];
    // every entry this tile places lies inside the caller's buffers (and inside u32 offsets)
    const bool fits = nnz_before + tile_nnz <= a.c_cap && nnz_before + tile_nnz <= 0xFFFFFFFFull;
    const bool last = blockIdx.x == nb - 1;
    if (last) {
        const u64 nnz_c = nnz_before + tile_nnz;
        const u64 g_total = chain_u64(s_pref, kCwTotLo, kCwTotHi) + (u64(s_mine[kCwTotHi]) << 32) + s_mine[kCwTotLo];
        if (t < kMaxClasses) {
            const u32 total = t < NUM_CLASSES ? (u32)s_pref[kCwClass + t] + s_mine[kCwClass + t] : 0u;
            a.st->num.count[t] = total;
            a.st->num.offset[t] = 0;
            if (total && !((a.cp.num_allowed >> t) & 1u)) a.st->capacity_miss = 1;
            if (t == NUM_G && a.expect_g_rows != ~0u && total != a.expect_g_rows) a.st->capacity_miss = 1;
        }
        if (t == 0) {
            a.st->nnz_c = nnz_c;
            a.st->max_row_nnz_c = max((u32)s_pref[kCwMax], s_mine[kCwMax]);
            if (nnz_c > 0xFFFFFFFFull) a.st->nnz_overflow = 1;
            if (nnz_c > a.c_cap) a.st->capacity_miss = 1;  // the caller's buffers do not hold this product: two-phase call
            a.st->g_products = g_total;
            if (a.expect_g != ~0ull && g_total != a.expect_g) a.st->capacity_miss = 1;
            a.st->walk_rows = a.m;
            if (!chain_ok || chain_error(chain)) {
                a.st->chain_error = 1;
                a.st->capacity_miss = 1;
            }
            a.offsets_out[a.m] = (u32)nnz_c;
            if (a.pred_off_out) a.pred_off_out[a.m] = (u32)nnz_c;
        }
    }
    if (!chain_ok) return;
    if (!fits) {
        if (t == 0) a.st->capacity_miss = 1;
        return;
    }

    // ---- 3. offsets, records of the rows other launches compute, and the finished rows to their places
    if (t < kMaxClasses) s_run[t] = (u32)s_pref[kCwClass + t];
    __syncthreads();
    u32 off_run = (u32)nnz_before;  // (uniform: entries of C before the current sub-tile)
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const u32 lr = t + kWalkThreads * k;
        u32 total;
        const u32 excl = block_exclusive_scan<kWalkThreads>(c[k], s_scan, &total);
        const u32 off = off_run + excl;
        if (lr < nrows) {
            s_off[lr] = off;
            a.offsets_out[r0 + lr] = off;
            if (a.pred_off_out) a.pred_off_out[r0 + lr] = off;
        }
        if (a.recs && cls[k] != NUM_NONE) {
            // my row's place in the list of its numeric class: rows of the class in the tiles before mine, in the
            // sub-tiles and waves before mine, and in the lanes before me (ascending rows inside every class, as the
            // scan kernel leaves them)
            const u32 q = cls[k];
            u32 pos = s_run[q];
            for (int kk = 0; kk < k; ++kk)
                for (int w = 0; w < NW; ++w) pos += s_wcnt[kk][q][w];
            for (u32 w = 0; w < wid; ++w) pos += s_wcnt[k][q][w];
            // (the lanes of my wave with the same class: one ballot per class that occurs in the wave)
            u64 todo = __ballot(true);
            u32 rank = 0;
            while (todo) {
                const u32 leader = (u32)__builtin_ctzll(todo);
                const u32 lq = (u32)__builtin_amdgcn_readlane((int)q, (int)leader);
                const u64 same = __ballot(q == lq);
                if (q == lq) rank = (u32)__popcll(same & lanemask_lt());
                todo &= ~same;
            }
            pos += rank;
            RowRec r;
            r.row = r0 + lr;
            r.a0 = s_a0[lr];
            r.a1 = s_a0[lr + 1];
            r.base = off;
            r.cmin = s_cmin[lr];
            r.cmax = r_cmax[k];
            r.ops = r_ops[k];
            r.nnz = c[k];
            if (pos < a.m) *class_rec_at(a.recs, a.m, q, pos) = r;
        }
        off_run += total;
    }
    __syncthreads();  // (s_off of every row, for the moves)
    if (void_call || (a.debug & 1u)) return;
    // the finished rows: slot -> C.  Register-class rows (<= 256 entries, most a few dozen) by 8-lane groups, the
    // numeric-first rows (dense windows: hundreds of entries) by waves.
    const u32 n_esc = s_start[4], n_all = s_start[5];
    for (u32 e = t >> 3; e < n_esc; e += kWalkThreads / 8) {
        const u32 lr = s_list[e];
        const u64 slot = s_slot[lr];
        walk_move_row<T, 8>(lane & 7u, a.pool_col + slot, pool_val + slot, a.c_col + s_off[lr], c_val + s_off[lr], s_nnz[lr]);
    }
    for (u32 e = n_esc + wid; e < n_all; e += NW) {
        const u32 lr = s_list[e];
        const u64 slot = s_slot[lr];
        const u32 n = slot + s_nnz[lr] <= a.pool_cap ? s_nnz[lr] : 0u;
        walk_move_row<T, 64>(lane, a.pool_col + slot, pool_val + slot, a.c_col + s_off[lr], c_val + s_off[lr], n);
    }
}

static u32 g_walk_tile_rows = 0, g_walk_debug = 0;
void set_walk_debug(u32 tile_rows, u32 flags) { g_walk_tile_rows = tile_rows, g_walk_debug = flags; }
u32 walk_tile_rows(u32 m)
{
    if (g_walk_tile_rows && u64(g_walk_tile_rows) * kChainMaxBlocks >= m) return g_walk_tile_rows;
    // tiles of 64 rows up to 4096 x 64 rows, then 128, 256, 512 (the chain holds kChainMaxBlocks workgroups)
    u32 tr = 64;
    while (u64(tr) * kChainMaxBlocks < m) tr <<= 1;
    return tr;
}
u32 walk_max_rows() { return 512u * kChainMaxBlocks; }

template <typename T>
void launch_walk(hipStream_t s, const WalkArgs& args, const ProductSrc<T>& src, const Chain& chain, hipEvent_t e0, hipEvent_t e1)
{
    WalkArgs a = args;
    a.tile_rows = walk_tile_rows(a.m);
    a.debug = g_walk_debug;
    const u32 tiles = (a.m + a.tile_rows - 1) / a.tile_rows;
    const u32 lds = kWalkWaves * walk_wave_lds<T>();
    if (a.tile_rows <= 256)
        SPECK_LAUNCH_TIMED((walk_kernel<T, 1>), dim3(tiles), dim3(kWalkThreads), lds, s, e0, e1, src, a, chain);
    else
        SPECK_LAUNCH_TIMED((walk_kernel<T, 2>), dim3(tiles), dim3(kWalkThreads), lds, s, e0, e1, src, a, chain);
}
template void launch_walk<double>(hipStream_t, const WalkArgs&, const ProductSrc<double>&, const Chain&, hipEvent_t, hipEvent_t);
template void launch_walk<float>(hipStream_t, const WalkArgs&, const ProductSrc<float>&, const Chain&, hipEvent_t, hipEvent_t);

}  // namespace speck
