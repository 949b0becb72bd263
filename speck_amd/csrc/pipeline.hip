// pipeline.hip -- host pipeline + C-ABI of the MI355X SpGEMM backend.
// Role of the reference's MultiplyspECKImplementation (source/GPU/Multiply.cu:51-1122),
// spECKConfig (include/spECKConfig.h) and dCSR helpers (source/dCSR.cpp), re-designed:
//   * one grow-only scratch arena per config (the reference cudaMallocs/frees every
//     scratch buffer inside each call, Multiply.cu:202-225,1056-1070)
//   * the launch sequence is STATIC: every kernel takes its row list and counts from a
//     device-side stats block and its grid depends only on rows(A), so
//       - the eager path needs ONE blocking read-back (nnz(C), to allocate C) instead of the
//         reference's 5-8, and
//       - a repeated call with the same buffers (the benchmark loop, Executor.cpp:59-72)
//         replays a captured hipGraph: one graph launch + one synchronisation per multiply.
//   * kernel classes run concurrently on separate streams between explicit fork/join events.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../include/speck_c_api.h"
#include "device_common.hpp"
#include "launch.hpp"

using namespace speck;

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            std::fprintf(stderr, "speck_amd: HIP error %s at %s:%d (%s)\n",                 \
                         hipGetErrorString(_e), __FILE__, __LINE__, #expr);                 \
            return (_e == hipErrorOutOfMemory) ? SPECK_ERR_OOM : SPECK_ERR_HIP;             \
        }                                                                                   \
    } while (0)

// Everything a captured launch sequence is specialised to.
struct GraphKey {
    const void* ptr[10] = {};
    u64 num[8] = {};
    bool operator==(const GraphKey& o) const
    {
        return std::memcmp(ptr, o.ptr, sizeof(ptr)) == 0 && std::memcmp(num, o.num, sizeof(num)) == 0;
    }
};

// What an eager call leaves behind for a replay of the SAME call (same buffers, same sizes) to verify instead of
// recompute: the row offsets of C, per scan tile / analysis block where its rows go in the class lists (launch.hpp,
// kPredTileWords), and the final statistics block.  One device allocation, carved.  The config keeps two: `pred`,
// rewritten by every eager call, and `gpred`, the snapshot a captured sequence owns -- an eager call on OTHER
// buffers in between must not change what that sequence compares against (and writes by).
struct Prediction {
    void* buf = nullptr;
    size_t bytes = 0;
    u32* off = nullptr;        // [rows + 1]
    u32* num_tile = nullptr;   // [scan_tiles(rows)][kPredTileWords]
    u32* sym_block = nullptr;  // [analysis_blocks(rows)][kPredBlockWords]
    DeviceStats* stats = nullptr;
    u32 rows = 0;
};

// What a replayed launch sequence is specialised to: the classes that were non-empty when the same inputs were last
// multiplied eagerly, and what the previous identical call lets it skip.
struct ReplayPlan {
    u32 num_mask, launch_mask;
    u32 num_counts[kMaxClasses];
    bool direct, fused, pred_scan, pred_sym;
    bool skip_scan;  // ... and so does every kernel that produces a row's nnz (RowWork::verify_counts): no scan kernel -- the
                     //   numeric launches read the records, offsets and class table the previous replay's scan left
    bool uncaptured;  // the launches of this sequence are enqueued at every call instead of captured into a graph (below)
    bool num_verify;  // ... and no symbolic pass for the rows of the hash / dense classes: the numeric light launch verifies
                      //   their nnz itself (RowWork::verify_numeric) -- the symbolic phase is the register-class rows alone
    bool overlap;  // the analysis only VERIFIES what the previous identical call left in the arena, on a stream of its
                   //   own beside the symbolic / scan / numeric launches (which read that: DESIGN.md 4.3)
    // ... and everything else of "the last eager call" the launches are sized from: a sequence that is enqueued anew at
    // every call (caller's stream, replay_uncaptured, profile_replay) must not pick up what a multiply of ANOTHER
    // problem left in the config since (a captured graph has it baked in)
    u32 sym_mask;
    u32 sym_counts[kMaxClasses];
    u64 g_products, nf_cap_entries;
    u32 nf_wcols;
};

struct speck_config {
    int device = 0;
    int sm = 0;                 // compute units
    int max_static_lds = 0;
    int max_dynamic_lds = 0;
    std::vector<hipStream_t> streams;  // 6, as in the reference (spECKConfig.h:23-26)
    hipEvent_t completeStart = nullptr, completeEnd = nullptr, individualStart = nullptr,
               individualEnd = nullptr;
    hipStream_t user_stream = nullptr;
    bool use_user_stream = false;

    // scratch arena (device), grow-only
    void* arena = nullptr;
    size_t arena_bytes = 0;
    DeviceStats* d_stats = nullptr;
    DeviceStats* h_stats = nullptr;  // pinned, mapped
    DeviceStats* h_stats_dev = nullptr;  // device address of h_stats
    u32* d_ticket = nullptr;             // completion ticket of the replayed sequence: device counter,
    u32* h_ticket = nullptr;             //   pinned host copy (the host spins on it),
    u32* h_ticket_dev = nullptr;         //   its device address
    u32 ticket_expected = 0;
    bool spin_wait = true;
    ClassifyParams cp{};
    int profile_kernels = 0;  // 1: HIP events around every launch; 2: around the phases only (no event between
                              //    the class launches of a phase: their spans are what a replayed sequence sees)
    bool profile_replay = false;  // option profile_replay: a profiled call that COULD be replayed runs the launches
                                  //    of the replayed sequence (uncaptured) instead of the eager path
    std::vector<hipEvent_t> kev;   // kernel event pool (timing)
    std::vector<hipStream_t> aux;  // one stream per kernel class: classes run concurrently
    std::vector<hipEvent_t> aux_done;
    hipEvent_t fork = nullptr;
    bool eager_speculate = true;  // option eager_speculate: an eager call that follows another one on this config sizes
                                  //   its symbolic phase (grids, launched classes, scratch pool, numeric-first window) from
                                  //   THAT call and runs analysis .. scan as one batch -- one read-back instead of two; the
                                  //   device checks every assumption (capacity_miss: the two-read-back sequence re-runs)
    bool spec_valid = false;      // such a call has completed (last_sym_* describe it)
    u64 spec_rows_a = 0, spec_rows_b = 0;
    int eager_spec_hits = 0, eager_spec_misses = 0;
    const u32* stage_off_src = nullptr;  // eager call in flight: the staged row offsets ride to C in the numeric light
    u32* stage_off_dst = nullptr;        //   launch (RowWork::off_src)
    u32 stage_off_n = 0;
    bool validate_inputs = true;  // eager path: B's rows strictly ascending and in range
    u32 epoch_counter = 0, check_epoch = 0;  // ... reported as the call's epoch (DeviceStats::b_bad_epoch); set while
                                             //   an eager call that checks is in flight
    bool concurrent_classes = true;
    u32 max_side_streams = 12;
    float split_min_us = 10.f; // both parts of a light launch must be at least this long to be launched apart
    float fork_min_us = 60.f;  // estimated duration from which a class launch gets its own stream
    bool merge_light = true;  // all 256-thread classes of a phase in one launch
    bool split_light = false; // ... in two back-to-back launches, by LDS / register need (since the register classes
                              //   sort without LDS tables the small rows are bound by their gathers, and ONE launch
                              //   overlaps them with the latency-bound wave / workgroup rows: -5 % on every stand-in)
    u32 xcd_aware = 10;       // class lists walked in per-XCD contiguous slices: bit 0 sub-wave hash classes,
                              //   bit 1 dense-window / bitmap classes, bit 2 workgroup hash classes, bit 3 the
                              //   register classes (measured: +3.5 % on the cant stand-in for bit 1, -9 % time of
                              //   the small-row launch on the mac_econ stand-in for bit 3 -- that launch is bound by
                              //   its gathers of B rows since it sorts in registers; bits 0 and 2 lose to load
                              //   imbalance)
    u32 last_sym_counts[kMaxClasses] = {}, last_num_counts[kMaxClasses] = {};

    // captured launch sequence of the last repeated call
    bool use_graph = true;
    bool graph_valid = false;
    GraphKey graph_key;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    u32 last_sym_mask = 0, last_num_mask = 0;  // non-empty classes of the last eager call
    u32 last_max_row_nnz = 0;                  // ... and its longest C row
    GraphKey last_key;                         // ... and what it ran on
    bool last_key_valid = false;
    int graph_replays = 0, graph_captures = 0, graph_misses = 0;
    void* gpool = nullptr;    // global-memory buffers of the NUM_G spill path, carved into `spill`
    size_t gpool_bytes = 0;
    void* nfpool = nullptr;   // scratch slots of the numeric-first rows: col_ids | values (grow-only)
    size_t nfpool_bytes = 0;
    u64 nf_cap_entries = 0;
    size_t nf_pool_max_bytes = 0;  // 0: half of the free device memory at allocation time
    int pool_fallbacks = 0;        // times a scratch-pool class was switched off because the pool did not fit
    // direct placement of the numeric-first rows by a replayed sequence: the row offsets of the last eager call
    // (the config's own copy -- C.row_offsets is the caller's to overwrite), and the C buffers of the capture
    Prediction pred, gpred;
    DeviceStats last_eager_stats{};  // final statistics block of the last eager call (what `pred` goes with)
    bool pred_valid = false;         // pred.off holds the offsets of the last eager call
    bool pred_tiles_valid = false;   // ... and pred.num_tile its tile tables,
    bool pred_fold_esc = false;      //     in the shape of a sequence that finishes the register-class rows early
    bool pred_scan = true;           // option pred_scan: a replayed sequence scans with launch_scan_predicted
    bool capture_pred_scan = false;  // set while such a sequence is being enqueued
    bool pred_sym = true;            // option pred_sym: ... and bins its rows for the symbolic phase inside the analysis
                                     //   kernel, at the list positions of the previous identical call (no scatter kernel)
    bool capture_pred_sym = false;
    bool nf_direct = true;           // option nf_direct
    bool capture_direct = false;     // set while a sequence with direct placement is being captured
    bool esc_fused = true;           // option esc_fused: such a sequence finishes the rows of the register classes in
                                     //   its symbolic phase
    bool capture_fused = false;      // set while a sequence that does so is being captured
    u32* capture_c_col = nullptr;
    void* capture_c_val = nullptr;
    bool graph_direct = false;       // the captured sequence places the numeric-first rows directly
    bool graph_fused = false;        // ... and finishes the rows of the register classes in its symbolic phase
    bool graph_pred_scan = false;    // ... and scans with the predicted kernel
    bool graph_pred_sym = false;     // ... and has no scatter kernel
    ReplayPlan graph_plan{};
    bool exec_dirty = false;         // other launches went onto the pipeline stream since graph_exec was last launched
    bool skip_scan = true;           // option skip_scan: a replayed sequence that follows a replay of itself has no scan kernel
    bool capture_skip_scan = false;  // set while such a sequence is being enqueued
    int num_verify = 1;              // option num_verify (0: never, 1: when it pays, 2: whenever possible): ... and no symbolic pass for its hash / dense rows (ReplayPlan::num_verify)
    bool capture_num_verify = false;
    bool capture_forked = false;     // option capture_forked (ReplayPlan::uncaptured)
    // The inputs the analysis depends on as the last WRITING analysis saw them (stages.hip, launch_snapshot_inputs): A's
    // column ids | B's row offsets | first and last column id of every row of B.  The verifier of a replayed sequence
    // compares the inputs with this copy instead of recomputing the analysis (option verify_inputs); grow-only, and like
    // the arena it belongs to whoever ran a writing analysis last (snap_for_arena travels with arena_key).
    u32* snap = nullptr;
    size_t snap_words = 0, snap_a_words = 0;
    bool verify_inputs = true;
    bool snap_pending = false;       // the verifier of the call in flight recomputes the analysis AND takes the copy
    bool snap_for_arena = false;     // the copy holds the inputs the arena's metadata was derived from (and verified against)
    std::function<int()> after_analysis;  // set by an eager call for the duration of its first batch (multiply_impl)
    bool gate_verifier = false;      // profiled pre-pass: the verifier's stream waits for the symbolic phase of the timed sequence
    bool arena_from_replay = false;  // the arena (numeric records, class table, statistics) was last written by a completed
                                     //   REPLAY of arena_key's problem: the layout a sequence without a scan reads
    bool overlap_analysis = true;    // option overlap_analysis: a replayed sequence runs its analysis as a verifier beside it
    bool capture_overlap = false;    // set while such a sequence is being enqueued
    bool graph_overlap = false;      // the captured sequence does so
    hipStream_t vstream = nullptr;   // the verifier's stream
    u32* h_verify = nullptr;         // its verdict (word 0) and its completion ticket (word 16): pinned, mapped
    u32* h_verify_dev = nullptr;
    u32* d_vticket = nullptr;
    u32 vticket_expected = 0;
    bool validate_in_flight = false; // the input check of an eager call is running on vstream (begin_validate)
    GraphKey arena_key;              // what the per-row / per-entry metadata in the arena (b_sl, row arrays, symbolic records,
    bool arena_key_valid = false;    //   class table) was last written for -- by a multiply that COMPLETED
    bool replay_uncaptured = false;  // option replay_uncaptured (debugging): enqueue the sequence instead of launching its graph
    u32 nf_wcols = kNumD1Cols;  // LDS window of the numeric-first kernel: the widest such row of the last analysis
    SpillBuffers spill{};
    u64 last_g_products = 0;  // what the spill pools of the captured sequence were sized for
    speck_stats last{};
};

namespace {

// polite busy-wait step of the host thread that waits for the completion ticket
inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    asm volatile("" ::: "memory");
#endif
}
constexpr std::chrono::microseconds kSpinBudget{2000};

hipStream_t main_stream(speck_config* c) { return c->use_user_stream ? c->user_stream : c->streams[0]; }

void drop_graph(speck_config* c)
{
    if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
    if (c->graph) (void)hipGraphDestroy(c->graph);
    c->graph_exec = nullptr;
    c->graph = nullptr;
    c->graph_valid = false;
}

int ensure_arena(speck_config* c, size_t bytes)
{
    if (bytes <= c->arena_bytes) return SPECK_OK;
    drop_graph(c);
    c->last_key_valid = false;
    c->arena_key_valid = false;
    if (c->arena) HIP_TRY(hipFree(c->arena));
    c->arena = nullptr;
    c->arena_bytes = 0;
    size_t want = bytes + bytes / 4 + (1 << 20);
    HIP_TRY(hipMalloc(&c->arena, want));
    c->arena_bytes = want;
    return SPECK_OK;
}

// room for the input snapshot of a problem (no room: the verifier recomputes the analysis instead)
void ensure_snap(speck_config* c, u64 nnz_a, u64 b_rows)
{
    const size_t a_words = (size_t(nnz_a) + 63) & ~size_t(63), need = a_words + 3 * size_t(b_rows) + 1;
    if (need > c->snap_words) {
        if (c->snap) (void)hipFree(c->snap);
        c->snap = nullptr;
        c->snap_words = 0;
        c->snap_for_arena = false;
        const size_t want = need + need / 8;
        if (hipMalloc(reinterpret_cast<void**>(&c->snap), want * sizeof(u32)) != hipSuccess) {
            (void)hipGetLastError();
            c->snap = nullptr;
            return;
        }
        c->snap_words = want;
    }
    if (c->snap_a_words != a_words) c->snap_for_arena = false;  // (another layout: whatever it holds is not this problem's)
    c->snap_a_words = a_words;
}

struct Carver {
    unsigned char* p;
    size_t used = 0;
    explicit Carver(void* base) : p(static_cast<unsigned char*>(base)) {}
    template <typename U>
    U* take(size_t n)
    {
        size_t bytes = (n * sizeof(U) + 255) & ~size_t(255);
        U* r = reinterpret_cast<U*>(p + used);
        used += bytes;
        return r;
    }
    static size_t need(size_t n, size_t elem) { return (n * elem + 255) & ~size_t(255); }
};

struct Scratch {
    u32 *row_ops, *row_max_ops, *row_col_min, *row_col_max;
    RowRec* recs;  // one 32-byte record per row, grouped by (numeric) kernel class
    RowRec* recs_sym;  // ... and by symbolic class: kept apart, so that the symbolic records of a call survive its scan
                       //   (a replayed sequence whose analysis only verifies reads those of the previous identical call)
    u8* cls;       // numeric class of every row
    u8* cls_sym;   // symbolic class of every row
    u32* a_ro_copy;  // A's row offsets as the analysis saw them
    BlockPartial* partials;
    uint2* b_sl;  // per A entry: (start, length) of the referenced B row (written by the analysis)
    uint2* w_sl;  // per A entry: its B entries inside the current column window (multi-window rows)
    u64* nf_off;           // per row: scratch slot of a numeric-first row
    u32* counts;           // per row (+1): nnz of the C row, written by the symbolic kernels
    u32* offsets;          // per row (+1): C.row_offsets of an eager call until C is known to be allocated
};

u32 partial_blocks(u32 m) { return std::max(analysis_blocks(m), scan_tiles(m)) + 2; }  // + PartialArrays padding

size_t scratch_bytes(u32 m, u64 nnz_a)
{
    size_t b = 2 * Carver::need(nnz_a, 8);
    b += 4 * Carver::need(m, 4);
    b += 2 * Carver::need(size_t(m) + 1, 4);
    b += Carver::need(m, 8);
    b += 2 * Carver::need(m, sizeof(RowRec));
    b += 2 * Carver::need(m, 1);
    b += Carver::need(size_t(m) + 1, 4);
    b += Carver::need(partial_blocks(m), sizeof(BlockPartial));
    return b + 4096;
}

Scratch carve(speck_config* c, u32 m, u64 nnz_a)
{
    Carver cv(c->arena);
    Scratch s;
    s.b_sl = cv.take<uint2>(nnz_a);
    s.w_sl = cv.take<uint2>(nnz_a);
    s.nf_off = cv.take<u64>(m);
    s.recs = cv.take<RowRec>(m);
    s.recs_sym = cv.take<RowRec>(m);
    s.cls_sym = cv.take<u8>(m);
    s.a_ro_copy = cv.take<u32>(size_t(m) + 1);
    s.row_ops = cv.take<u32>(m);
    s.row_max_ops = cv.take<u32>(m);
    s.row_col_min = cv.take<u32>(m);
    s.row_col_max = cv.take<u32>(m);
    s.cls = cv.take<u8>(m);
    s.counts = cv.take<u32>(size_t(m) + 1);
    s.offsets = cv.take<u32>(size_t(m) + 1);
    s.partials = cv.take<BlockPartial>(partial_blocks(m));
    return s;
}

// (re)allocate and carve a Prediction for `m` rows; false if the device has no room (the replay then predicts less)
bool ensure_pred(Prediction& p, u32 m)
{
    const size_t need = Carver::need(size_t(m) + 1, 4) + Carver::need(size_t(scan_tiles(m)) * kPredTileWords, 4) +
                        Carver::need(size_t(analysis_blocks(m)) * kPredBlockWords, 4) +
                        Carver::need(1, sizeof(DeviceStats));
    if (need > p.bytes) {
        if (p.buf) (void)hipFree(p.buf);
        p = Prediction{};
        if (hipMalloc(&p.buf, need) != hipSuccess) {
            (void)hipGetLastError();
            p.buf = nullptr;
            return false;
        }
        p.bytes = need;
    }
    Carver cv(p.buf);
    p.off = cv.take<u32>(size_t(m) + 1);
    p.num_tile = cv.take<u32>(size_t(scan_tiles(m)) * kPredTileWords);
    p.sym_block = cv.take<u32>(size_t(analysis_blocks(m)) * kPredBlockWords);
    p.stats = cv.take<DeviceStats>(1);
    p.rows = m;
    return true;
}

// scratch pool of the numeric-first rows: `entries` column ids followed by `entries` values
int ensure_nfpool(speck_config* c, u64 entries, size_t vsize)
{
    const size_t need = Carver::need(entries, 4) + Carver::need(entries, vsize) + 512;
    if (entries <= c->nf_cap_entries && need <= c->nfpool_bytes) return SPECK_OK;
    drop_graph(c);
    c->last_key_valid = false;
    if (c->nfpool) (void)hipFree(c->nfpool);
    c->nfpool = nullptr;
    c->nfpool_bytes = 0;
    c->nf_cap_entries = 0;
    const u64 cap = entries + entries / 8 + 1024;
    const size_t bytes = Carver::need(cap, 4) + Carver::need(cap, 8) + 512;
    // budget: an explicit cap (option nf_pool_max_mb) or half of what the device has free -- hidden scratch must
    // not be what makes a later allocation of C fail
    size_t free_b = 0, total_b = 0;
    size_t budget = c->nf_pool_max_bytes;
    if (!budget && hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = free_b / 2;
    if (budget && bytes > budget) return SPECK_ERR_OOM;
    if (hipMalloc(&c->nfpool, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return SPECK_ERR_OOM;
    }
    c->nfpool_bytes = bytes;
    c->nf_cap_entries = cap;
    return SPECK_OK;
}

RowWork make_work(speck_config* c, const Scratch& sc, const SpillBuffers& spill, bool symbolic_phase = false)
{
    RowWork w{};
    w.recs = symbolic_phase ? sc.recs_sym : sc.recs;
    w.st = c->d_stats;
    w.b_sl = sc.b_sl;
    w.spill = spill;
    w.nf_off = sc.nf_off;
    w.nf_col = static_cast<u32*>(c->nfpool);
    w.nf_val = c->nfpool ? static_cast<unsigned char*>(c->nfpool) + Carver::need(c->nf_cap_entries, 4) : nullptr;
    w.nf_cap = c->nfpool ? c->nf_cap_entries : 0;
    w.nf_pred_off = (c->capture_direct || c->capture_skip_scan) ? c->gpred.off : nullptr;
    w.nf_direct_col = c->capture_direct ? c->capture_c_col : nullptr;
    w.nf_direct_val = c->capture_direct ? c->capture_c_val : nullptr;
    w.w_sl = sc.w_sl;
    w.xcd_aware = c->xcd_aware;
    w.off_src = symbolic_phase ? nullptr : c->stage_off_src;
    w.off_dst = symbolic_phase ? nullptr : c->stage_off_dst;
    w.off_n = symbolic_phase ? 0u : c->stage_off_n;
    w.verify_counts = c->capture_skip_scan ? 1u : 0u;
    w.verify_numeric = c->capture_num_verify ? 1u : 0u;
    return w;
}

struct StageTimer {
    speck_config* c;
    bool on;
    hipStream_t s;
    StageTimer(speck_config* cfg, bool enable, hipStream_t st) : c(cfg), on(enable), s(st)
    {
        if (on) start();
    }
    void start() { (void)hipEventRecord(c->individualStart, s); }
    // returns ms since start() and restarts (reference: recordTimerVar/startTimerVar,
    // source/GPU/Multiply.cu:36-49)
    float lap()
    {
        if (!on) return 0.f;
        float ms = 0.f;
        (void)hipEventRecord(c->individualEnd, s);
        (void)hipEventSynchronize(c->individualEnd);
        (void)hipEventElapsedTime(&ms, c->individualStart, c->individualEnd);
        start();
        return ms;
    }
};

int check_inputs(const speck_dcsr* A, const speck_dcsr* B)
{
    if (!A || !B) return SPECK_ERR_INVALID;
    if (A->cols != B->rows) return SPECK_ERR_INVALID;
    // reference limits, source/GPU/Multiply.cu:57-66 (hash key / block-range packing);
    // kept so that every input the reference accepts is accepted and vice versa
    if (B->cols > (1ull << 27) || A->rows > (1ull << 27)) return SPECK_ERR_DIM_LIMIT;
    if (A->rows && !A->row_offsets) return SPECK_ERR_INVALID;
    if (B->rows && !B->row_offsets) return SPECK_ERR_INVALID;
    return SPECK_OK;
}

hipEvent_t kernel_event(speck_config* c, size_t i)
{
    while (c->kev.size() <= i) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        c->kev.push_back(e);
    }
    return c->kev[i];
}

// Launch one kernel per class in `mask`.  The classes are independent (disjoint rows), so each
// runs on its own stream between a fork and a join event on the pipeline stream: the
// latency-bound heavy-row kernels (few workgroups) overlap the throughput-bound small-row ones
// (the reference does the same with its 6 streams, source/GPU/Multiply.cu:494-553, but relies on
// legacy default-stream ordering; here the dependencies are explicit events).
struct ClassTiming {
    int cls;
    size_t ev;
};

// pseudo classes: the merged launches of the 256-thread classes -- the big-LDS ones and the small ones
constexpr int kLightBig = -1, kLightTiny = -2;

template <typename LaunchFn>
int run_classes(speck_config* c, hipStream_t s, const int* order, int n_order, u32 mask, u32 big_mask,
                u32 tiny_mask, const u32* counts, const float* ns_per_row, size_t* ev_idx,
                std::vector<ClassTiming>* timing, int exact_cls, LaunchFn&& launch)
{
    bool forked = false;
    size_t used = 0;
    auto active = [&](int cls) {
        if (cls == kLightBig) return (mask & big_mask) != 0;
        if (cls == kLightTiny) return (mask & tiny_mask) != 0;
        return (mask >> cls & 1u) != 0;
    };
    // Estimated duration of an item from the host-known row counts of the previous identical call
    // (isolated per-row costs, scripts/class_times.py).  An item earns a side stream only when it is
    // long enough to pay for the cross-queue dependency (~10-15 us per branch of a replayed graph):
    // the scircuit / mac_econ stand-ins run fastest on ONE stream, the webbase one with four.
    auto est_us = [&](int cls) -> float {
        if (!counts) return 1e9f;  // eager first call: counts unknown, keep every class apart
        u32 m = cls == kLightBig ? (mask & big_mask) : cls == kLightTiny ? (mask & tiny_mask) : (1u << cls);
        float us = 0.f;
        for (int k = 0; k < kMaxClasses; ++k)
            if (m >> k & 1u) us += counts[k] * ns_per_row[k] * 1e-3f;
        return us;
    };
    int n_big = 0;
    for (int i = 0; i < n_order; ++i)
        if (active(order[i]) && est_us(order[i]) >= c->fork_min_us) ++n_big;
    // The last active item stays on the pipeline stream: a fork costs its branch 10-20 us of
    // cross-queue latency, a join on an already finished branch almost nothing -- so a phase with
    // one kernel pays no event at all, and the merged light launch (usually the longest) starts
    // at once while the heavy-row kernels start late on their side streams and still finish first.
    int last_active = -1;
    for (int i = 0; i < n_order; ++i)
        if (active(order[i])) last_active = i;
    // ... and the last LONG one if there are several (the short ones queue on the pipeline stream too)
    if (n_big >= 2)
        for (int i = 0; i < n_order; ++i)
            if (active(order[i]) && est_us(order[i]) >= c->fork_min_us) last_active = i;
    // at most `max_side_streams` branches: every cross-queue dependency of a replayed graph costs
    // ~10 us, so further side items queue behind each other on the last side stream
    const size_t max_side = std::min<size_t>(c->aux.size(), c->max_side_streams);
    size_t touched = 0;  // side streams that carry work
    for (int i = 0; i < n_order; ++i) {
        const int cls = order[i];
        if (!active(cls)) continue;
        hipStream_t ks = s;
        if (c->concurrent_classes && max_side > 0 && i != last_active && n_big >= 2 &&
            est_us(cls) >= c->fork_min_us) {
            if (!forked) {
                HIP_TRY(hipEventRecord(c->fork, s));
                forked = true;
            }
            const size_t slot = std::min(used, max_side - 1);
            ks = c->aux[slot];
            if (slot >= touched) {
                HIP_TRY(hipStreamWaitEvent(ks, c->fork, 0));
                touched = slot + 1;
            }
            ++used;
        }
        const bool timed = c->profile_kernels == 1 && timing;
        // the light launches and the numeric-first one stamp their events themselves, with the kernel's own begin and
        // end (launch.hpp, SPECK_LAUNCH_TIMED); the others are bracketed by two event records
        const bool exact = timed && (cls == kLightBig || cls == kLightTiny || cls == exact_cls);
        hipEvent_t e0 = timed ? kernel_event(c, *ev_idx) : nullptr, e1 = timed ? kernel_event(c, *ev_idx + 1) : nullptr;
        if (timed && !exact) (void)hipEventRecord(e0, ks);
        launch(ks, cls, exact ? e0 : nullptr, exact ? e1 : nullptr);
        if (timed) {
            if (!exact) (void)hipEventRecord(e1, ks);
            timing->push_back({cls, *ev_idx});
            *ev_idx += 2;
        }
    }
    for (size_t k = 0; k < touched; ++k) {
        HIP_TRY(hipEventRecord(c->aux_done[k], c->aux[k]));
        HIP_TRY(hipStreamWaitEvent(s, c->aux_done[k], 0));
    }
    HIP_TRY(hipGetLastError());
    return SPECK_OK;
}

// isolated cost per row of every class (ns, MI355X, scripts/class_times.py on the four stand-ins)
constexpr float kSymNsPerRow[kMaxClasses] = {0.15f, 0.6f, 3.f, 5.f, 30.f, 1000.f, 8.5f, 1000.f, 12.f, 50000.f, 0.1f, 0.4f,
                                             0.4f, 0.8f, 0.05f, 0.f};
constexpr float kNumNsPerRow[kMaxClasses] = {0.1f, 0.3f, 2.f, 4.f, 12.f, 75.f, 12.f, 300.f, 1500.f, 2.5f, 1.5f, 0.2f,
                                             0.6f, 1.2f, 0.1f, 0.f};
constexpr u32 kAllSym = (1u << SYM_CLASSES) - 1u;
constexpr u32 kAllNum = (1u << NUM_CLASSES) - 1u;

u32 mask_of(const u32* counts, int n)
{
    u32 m = 0;
    for (int i = 0; i < n; ++i)
        if (counts[i]) m |= 1u << i;
    return m;
}

struct Timing {
    size_t ev = 0, ev_analysis = 0, ev_between = 0, ev_analysis_end = 0, ev_scan = 0, ev_num = 0;
    std::vector<ClassTiming> sym, num;
};

// analysis -> symbolic classes -> scan + numeric classification.  Nothing here needs a host
// decision: `sym_mask` only prunes kernels of classes known to be empty (eager path: all).
int enqueue_front(speck_config* c, hipStream_t s, const speck_dcsr* A_in, const speck_dcsr* B,
                  const Scratch& sc, u32* offsets_out, u32 vsize, u64 exact_nnz, u32 sym_mask, u32 num_mask,
                  bool classify_numeric, Timing* tm, const u32* sym_hint = nullptr,
                  DeviceStats* host_mirror = nullptr, u64 expect_g = ~0ull, u32 expect_g_rows = ~0u,
                  u32 parts = 3 /* 1: analysis + binning, 2: symbolic launches + scan */, u64 expect_nf = ~0ull,
                  const Prediction* pred_out = nullptr /* eager: what this call leaves for a replay */,
                  bool pred_fold_esc = false, bool hint_exact = true /* sym_hint holds THIS call's counts (list positions) */)
{
    const u32 m = (u32)A_in->rows;
    // A sequence whose analysis only verifies (capture_overlap) reads A's row offsets where the previous identical call
    // left them, like the rest of the structure-derived metadata: whatever the caller does to A.row_offsets while the
    // sequence runs, its kernels see ONE consistent structure, and the verifier says whether it is still A's.
    speck_dcsr a_stored = *A_in;
    if (c->capture_overlap) a_stored.row_offsets = sc.a_ro_copy;
    const speck_dcsr* const A = &a_stored;
    u32* const c_ro = sc.counts;  // the symbolic kernels count into scratch; the scan writes offsets_out
    ClassifyParams cp = c->cp;
    cp.sym_allowed = sym_mask;
    cp.num_allowed = num_mask;
    cp.esc16 = (c->cp.esc16 && B->cols <= (1ull << 26)) ? 1u : 0u;  // (column << 6 | product number) fits 32 bits
    cp.esc_fused = c->capture_fused ? 1u : 0u;
    const bool timed = c->profile_kernels && tm;
    if (parts & 1u) {
        if (timed) {
            tm->ev_analysis = tm->ev;
            (void)hipEventRecord(kernel_event(c, tm->ev++), s);
        }
        hipEvent_t between = nullptr;
        if (timed) {
            tm->ev_between = tm->ev;
            between = kernel_event(c, tm->ev++);
        }
        // (capture_overlap: no analysis IN the sequence -- launch_verifier puts it on a stream of its own, outside any
        //  graph: a fork / join inside the captured graph cost ~25 us of cross-queue hand-offs, more than it hid)
        if (!c->capture_overlap)
        launch_analysis(s, A->row_offsets, A->col_ids, B->row_offsets, B->col_ids, m, A->nnz, sc.row_ops,
                        sc.row_max_ops, sc.row_col_min, sc.row_col_max, sc.cls_sym, c_ro, sc.partials, sc.recs_sym,
                        c->d_stats, cp, sc.b_sl, between, sc.nf_off, expect_nf, (u32)B->rows,
                        pred_out ? pred_out->sym_block : nullptr, c->capture_pred_sym ? c->gpred.sym_block : nullptr,
                        c->gpred.stats, (u32)B->cols, B->nnz, c->check_epoch, sc.a_ro_copy);
        if (!c->capture_overlap) c->snap_for_arena = false;  // (a writing analysis: the copy of the inputs is not its)
        // (eager call: the input check of B goes onto its stream HERE -- behind the launch of the analysis, the head of
        //  the call's critical path, and beside that latency-bound kernel rather than beside the symbolic launches)
        if (c->after_analysis) {
            const int hrc = c->after_analysis();
            if (hrc != SPECK_OK) return hrc;
        }
        if (timed) {
            tm->ev_analysis_end = tm->ev;
            (void)hipEventRecord(kernel_event(c, tm->ev++), s);
        }
        HIP_TRY(hipGetLastError());
    }
    if (!(parts & 2u)) return SPECK_OK;
    const RowWork w = make_work(c, sc, SpillBuffers{}, true);
    // heaviest classes first: they have the longest tails
    u32 all_m[kMaxClasses];
    for (auto& x : all_m) x = m;  // no host-known counts: size every class for rows(A)
    const u32* hint = sym_hint ? sym_hint : all_m;
    static const int merged[7] = {SYM_GH, SYM_BM2, SYM_B32K, SYM_B16K, SYM_NF, kLightBig, kLightTiny};
    static const int separate[SYM_CLASSES] = {SYM_GH,  SYM_BM2, SYM_B32K, SYM_B16K, SYM_NF,  SYM_B4K, SYM_BM1,
                                              SYM_W1K, SYM_W256, SYM_R64, SYM_R32, SYM_W128, SYM_G16, SYM_G8, SYM_G4};
    // the launch's LDS size is the largest need among its classes and caps the waves per CU of all of
    // them: the 256-thread classes go in two launches, the big-LDS ones apart (split_light); the
    // first runs on a side stream next to the second
    // ... unless one of the two parts is next to empty: then a second launch only adds a boundary
    auto part_us = [](const u32* counts, const float* ns, u32 mask) {
        float us = 0.f;
        for (int k = 0; k < kMaxClasses; ++k)
            if (mask >> k & 1u) us += counts[k] * ns[k] * 1e-3f;
        return us;
    };
    bool split_sym = c->split_light;
    if (split_sym && sym_hint)
        split_sym = part_us(sym_hint, kSymNsPerRow, kSymLightMask & (1u << SYM_BM1)) >= c->split_min_us &&
                    part_us(sym_hint, kSymNsPerRow, kSymLightMask & ~(1u << SYM_BM1)) >= c->split_min_us;
    const u32 sym_big = split_sym ? (1u << SYM_BM1) : kSymLightMask;
    int rc = run_classes(c, s, c->merge_light ? merged : separate, c->merge_light ? 7 : (int)SYM_CLASSES, sym_mask,
                         kSymLightMask & sym_big, kSymLightMask & ~sym_big, sym_hint, kSymNsPerRow,
                         tm ? &tm->ev : nullptr, tm ? &tm->sym : nullptr, (int)SYM_NF,
                         [&](hipStream_t ks, int cls, hipEvent_t e0, hipEvent_t e1) {
                             if (cls == kLightBig || cls == kLightTiny) {
                                 const u32 part = cls == kLightBig ? sym_big : ~sym_big;
                                 launch_symbolic_light(ks, hint, sym_mask & kSymLightMask & part, A->row_offsets,
                                                       sc.b_sl, B->col_ids, w, c_ro, c->sm,
                                                       sym_hint != nullptr && hint_exact, c->capture_fused ? vsize : 0u, A->data,
                                                       B->data, e0, e1);
                             } else if (cls == SYM_NF) {
                                 // the numeric dense-window kernel, in the symbolic phase (numeric.hip)
                                 if (vsize == 8) {
                                     CsrView<double> Av{A->row_offsets, A->col_ids, static_cast<const double*>(A->data), m, (u32)A->cols};
                                     CsrView<double> Bv{B->row_offsets, B->col_ids, static_cast<const double*>(B->data), (u32)B->rows, (u32)B->cols};
                                     launch_numeric_first<double>(ks, hint[cls], Av, Bv, w, c_ro, c->sm, c->nf_wcols, e0, e1);
                                 } else {
                                     CsrView<float> Av{A->row_offsets, A->col_ids, static_cast<const float*>(A->data), m, (u32)A->cols};
                                     CsrView<float> Bv{B->row_offsets, B->col_ids, static_cast<const float*>(B->data), (u32)B->rows, (u32)B->cols};
                                     launch_numeric_first<float>(ks, hint[cls], Av, Bv, w, c_ro, c->sm, c->nf_wcols, e0, e1);
                                 }
                             } else
                                 launch_symbolic(ks, cls, hint[cls], A->row_offsets, sc.b_sl,
                                                 B->col_ids, w, c_ro, c->sm);
                         });
    if (rc != SPECK_OK) return rc;
    if (timed) {
        tm->ev_scan = tm->ev;
        (void)hipEventRecord(kernel_event(c, tm->ev++), s);
    }
    if (c->capture_skip_scan) {
        // no scan: every row's nnz was compared with the previous call's where it was produced (store_row_count, the
        // fused register-class bodies, the numeric-first kernel); the numeric launches read what that call's scan left
    } else if (c->capture_pred_scan)
        launch_scan_predicted(s, c_ro, offsets_out, m, A->row_offsets, sc.row_ops, sc.row_col_min, sc.row_col_max,
                              sc.recs, c->d_stats, cp, c->gpred.off, c->gpred.num_tile, c->gpred.stats,
                              (c->capture_pred_sym && !c->capture_overlap) ? sc.partials : nullptr, c->capture_overlap);
    else
        launch_scan(s, c_ro, offsets_out, m, A->row_offsets, sc.row_ops, sc.row_col_min, sc.row_col_max,
                    classify_numeric ? sc.cls : nullptr, sc.partials, sc.recs, c->d_stats, cp, vsize, exact_nnz,
                    host_mirror, expect_g, expect_g_rows, c->capture_direct ? c->gpred.off : nullptr,
                    pred_out ? pred_out->off : nullptr, pred_out ? pred_out->num_tile : nullptr, pred_fold_esc,
                    host_mirror ? c->d_ticket : nullptr, host_mirror ? c->h_ticket_dev : nullptr);
    if (timed) (void)hipEventRecord(kernel_event(c, tm->ev++), s);
    HIP_TRY(hipGetLastError());
    return SPECK_OK;
}

template <typename T>
int enqueue_back(speck_config* c, hipStream_t s, const speck_dcsr* A, const speck_dcsr* B,
                 const Scratch& sc, const u32* /*c_ro*/, u32* c_col, T* c_val, u32 num_mask,
                 const u32* counts /*host-known, or nullptr*/, Timing* tm)
{
    const u32 m = (u32)A->rows;
    CsrView<T> Av{c->capture_overlap ? sc.a_ro_copy : A->row_offsets, A->col_ids, static_cast<const T*>(A->data), m, (u32)A->cols};
    CsrView<T> Bv{B->row_offsets, B->col_ids, static_cast<const T*>(B->data), (u32)B->rows,
                  (u32)B->cols};
    const RowWork w = make_work(c, sc, c->spill);
    u32 all_m[kMaxClasses];
    for (auto& x : all_m) x = m;
    const u32* hint = counts ? counts : all_m;
    static const int merged[6] = {NUM_G, NUM_D2, NUM_B8K, NUM_NFCOPY, kLightBig, kLightTiny};
    static const int separate[NUM_CLASSES] = {NUM_G,  NUM_D2,   NUM_B8K, NUM_B2K, NUM_W256, NUM_NFCOPY, NUM_D1,
                                              NUM_W512, NUM_R64, NUM_R32, NUM_W128, NUM_G16, NUM_G8, NUM_G4, NUM_DIRECT};
    constexpr u32 kBigPart = (1u << NUM_D1) | (1u << NUM_B2K) | (1u << NUM_W512) | (1u << NUM_W256);
    bool split_num = c->split_light;
    if (split_num && counts) {
        auto part_us = [&](u32 mask) {
            float us = 0.f;
            for (int k = 0; k < kMaxClasses; ++k)
                if (mask >> k & 1u) us += counts[k] * kNumNsPerRow[k] * 1e-3f;
            return us;
        };
        split_num = part_us(kNumLightMask & kBigPart) >= c->split_min_us &&
                    part_us(kNumLightMask & ~kBigPart) >= c->split_min_us;
    }
    const u32 num_big = split_num ? kBigPart : kNumLightMask;
    return run_classes(c, s, c->merge_light ? merged : separate, c->merge_light ? 6 : (int)NUM_CLASSES, num_mask,
                       kNumLightMask & num_big, kNumLightMask & ~num_big, counts, kNumNsPerRow,
                       tm ? &tm->ev : nullptr, tm ? &tm->num : nullptr, -100,
                       [&](hipStream_t ks, int cls, hipEvent_t e0, hipEvent_t e1) {
                           if (cls == kLightBig || cls == kLightTiny) {
                               const u32 part = cls == kLightBig ? num_big : ~num_big;
                               launch_numeric_light<T>(ks, hint, num_mask & kNumLightMask & part, Av, Bv, w, c_col,
                                                       c_val, c->sm, counts != nullptr, e0, e1);
                           } else
                               launch_numeric<T>(ks, cls, hint[cls], Av, Bv, w, c_col, c_val, c->sm);
                       });
}

// Wait for everything enqueued on `s` behind a done_kernel (launch_done): the host spins on the ticket that kernel
// stores into pinned memory (bounded by TIME -- work that is still running after kSpinBudget is long enough for the
// wake-up latency of the blocking call not to matter -- then the blocking synchronisation).
int wait_ticket(speck_config* c, hipStream_t s)
{
    if (c->spin_wait) {
        const u32 want = ++c->ticket_expected;
        const auto t0 = std::chrono::steady_clock::now();
        bool seen = false;
        while (!(seen = __atomic_load_n(c->h_ticket, __ATOMIC_ACQUIRE) == want)) {
            for (int i = 0; i < 64; ++i) cpu_relax();
            if (std::chrono::steady_clock::now() - t0 > kSpinBudget) break;
        }
        if (!seen) HIP_TRY(hipStreamSynchronize(s));
    } else {
        HIP_TRY(hipStreamSynchronize(s));
    }
    c->ticket_expected = __atomic_load_n(c->h_ticket, __ATOMIC_ACQUIRE);
    return SPECK_OK;
}

// The statistics block of the call so far, on the host.  With the ticket: done_kernel mirrors the block into pinned
// memory and the host spins (a copy + blocking synchronisation costs 10-20 us of wake-up latency, and the eager path
// reads back twice).
int read_stats(speck_config* c, hipStream_t s)
{
    if (c->spin_wait) {
        launch_done(s, c->d_ticket, c->h_ticket_dev, c->d_stats, c->h_stats_dev);
        const int rc = wait_ticket(c, s);
        if (rc != SPECK_OK) return rc;
    } else {
        HIP_TRY(hipMemcpyAsync(c->h_stats, c->d_stats, sizeof(DeviceStats), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    // the input check of this (eager) call reports with the call's epoch
    if (c->check_epoch && c->h_stats->b_bad_epoch == c->check_epoch) c->h_stats->b_invalid = 1;
    return SPECK_OK;
}

// ... when the scan of the batch mirrors the block and stores the ticket itself (enqueue_front with a host mirror):
// no done_kernel, the host has the statistics while the scan's last blocks are still running
int await_scan_stats(speck_config* c, hipStream_t s)
{
    if (!c->spin_wait) return read_stats(c, s);
    const int rc = wait_ticket(c, s);
    if (rc != SPECK_OK) return rc;
    if (c->check_epoch && c->h_stats->b_bad_epoch == c->check_epoch) c->h_stats->b_invalid = 1;
    return SPECK_OK;
}

void publish_counts(speck_config* c)
{
    c->last.sum_products = c->h_stats->sum_products;
    c->last.max_row_ops = c->h_stats->max_row_ops;
    c->last.nnz_c = c->h_stats->nnz_c;
    c->last.max_row_nnz_c = c->h_stats->max_row_nnz_c;
    for (int i = 0; i < SPECK_NUM_SYM_BINS; ++i) {
        c->last.sym_bin_rows[i] = c->h_stats->sym.count[i];
        c->last.sym_bin_bytes[i] = c->h_stats->sym.bytes[i];
    }
    for (int i = 0; i < SPECK_NUM_NUM_BINS; ++i) {
        c->last.num_bin_rows[i] = c->h_stats->num.count[i];
        c->last.num_bin_bytes[i] = c->h_stats->num.bytes[i];
    }
}

template <typename T>
GraphKey make_key(speck_config* c, const speck_dcsr* A, const speck_dcsr* B, const speck_dcsr* C,
                  hipStream_t s)
{
    GraphKey k;
    k.ptr[0] = A->row_offsets; k.ptr[1] = A->col_ids; k.ptr[2] = A->data;
    k.ptr[3] = B->row_offsets; k.ptr[4] = B->col_ids; k.ptr[5] = B->data;
    k.ptr[6] = C->row_offsets; k.ptr[7] = C->col_ids; k.ptr[8] = C->data;
    k.ptr[9] = c->arena;
    k.num[0] = A->rows; k.num[1] = A->nnz; k.num[2] = B->rows; k.num[3] = B->cols; k.num[4] = C->nnz;
    k.num[5] = sizeof(T);
    k.num[6] = (u64(c->cp.sym_bitmap_ratio) << 32) | c->cp.num_dense_ratio;
    k.num[7] = (u64(c->cp.num_global_passes) << 32) | (u64(c->cp.sym_w128) << 5) | (u64(c->cp.sym_g8) << 4) | (u64(c->cp.num_g8) << 3) | (u64(c->cp.num_w256) << 2) |
               (u64(c->cp.want_bytes) << 1) |
               (c->concurrent_classes ? 1u : 0u);
    k.num[7] ^= reinterpret_cast<u64>(s);
    k.num[5] |= u64(c->cp.nf_min_ops) << 8;
    k.num[5] |= (u64(c->cp.esc32) << 40) | (u64(c->cp.esc64) << 41) | (u64(c->cp.esc16) << 42) | (u64(c->cp.esc4) << 43);
    k.num[4] |= u64(c->cp.gh_per_window) << 32;  // C->nnz fits 32 bits
    return k;
}


// The config's prediction (of the last eager call = this call: same key) becomes the sequence's own: device copy of
// the arrays, and the statistics block of that call in the shape the sequence classifies in.
int snapshot_prediction(speck_config* c, hipStream_t s, const ReplayPlan& p)
{
    if (!c->pred_valid) return SPECK_OK;
    if (!ensure_pred(c->gpred, c->pred.rows)) return SPECK_ERR_OOM;
    HIP_TRY(hipMemcpyAsync(c->gpred.buf, c->pred.buf, std::min(c->pred.bytes, c->gpred.bytes), hipMemcpyDeviceToDevice, s));
    // (the statistics of THAT call, kept by the eager path: the pinned mirror holds whatever ran last -- a replay of
    //  another problem on this config, for instance)
    DeviceStats ps = c->last_eager_stats;
    ps.capacity_miss = 0;
    std::memcpy(ps.num.count, p.num_counts, sizeof(ps.num.count));
    u32 run = 0;
    for (int k = 0; k < kMaxClasses; ++k) {
        ps.num.offset[k] = run;
        run += ps.num.count[k];
    }
    ps.num.offset[kMaxClasses] = run;
    if (p.fused)
        for (int k = 0; k < kMaxClasses; ++k)
            if (kNumEscMask >> k & 1u) {
                ps.num.bytes[NUM_NFCOPY] += ps.num.bytes[k];
                ps.num.bytes[k] = 0;
            }
    // (a BLOCKING copy, behind the device-to-device one: `ps` is on the stack, and an asynchronous copy from pageable
    //  memory makes the runtime lock those pages behind the caller's back -- a later copy of other pageable memory,
    //  e.g. the caller downloading C, then ended in a GPU memory fault during the next replay: found by the stress run)
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipMemcpy(c->gpred.stats, &ps, sizeof(ps), hipMemcpyHostToDevice));
    return SPECK_OK;
}

ReplayPlan plan_replay(const speck_config* c, bool arena_replay_ok = false)
{
    // (The replayed sequence classifies exactly like the eager one.  Round 2 re-classified an under-filled NUM_B8K
    //  class into NUM_B2K at a load of 0.85 here; since the workgroup classes take rows up to that load in every
    //  call -- device_common.hpp, SPECK_LOAD_PCT -- there is nothing left to fold.)
    ReplayPlan p;
    p.num_mask = c->last_num_mask;
    std::memcpy(p.num_counts, c->last_num_counts, sizeof(p.num_counts));
    p.sym_mask = c->last_sym_mask;
    std::memcpy(p.sym_counts, c->last_sym_counts, sizeof(p.sym_counts));
    p.g_products = c->last_g_products;
    p.nf_cap_entries = c->nf_cap_entries;
    p.nf_wcols = c->nf_wcols;
    // Numeric-first rows: the eager call wrote them to scratch slots and copied them after the scan (nothing else
    // knows where a row goes before the scan).  The replayed sequence knows where they WENT: it writes each row
    // straight to the offset the previous identical call gave it, provided its fresh nnz is the same, and the scan
    // checks every fresh offset against that prediction -- no slot, no copy launch.
    // The same knowledge lets the rows of the register classes (NUM_G8 / NUM_G16: products sorted in registers,
    // nothing sized by the nnz) be finished in the SYMBOLIC phase: one walk of the row instead of two.  The numeric
    // phase then accounts for them as rows that are already in place.
    constexpr u32 kEscNum = kNumEscMask;
    p.fused = c->esc_fused && c->nf_direct && c->pred_valid && (p.num_mask & kEscNum) != 0 &&
              c->cp.sym_g8 == c->cp.num_g8 && c->merge_light;  // (the fused body lives in the merged light launch)
    if (p.fused) {
        for (int k = 0; k < kMaxClasses; ++k)
            if (kEscNum >> k & 1u) {
                p.num_counts[NUM_NFCOPY] += p.num_counts[k];
                p.num_counts[k] = 0;
            }
        p.num_mask = (p.num_mask & ~kEscNum) | (1u << NUM_NFCOPY);
    }
    p.direct = c->nf_direct && c->pred_valid && (p.num_mask >> NUM_NFCOPY & 1u);
    p.launch_mask = p.direct ? (p.num_mask & ~(1u << NUM_NFCOPY)) : p.num_mask;
    // Every row offset predicted, every tile table known (in the shape this sequence classifies in): the scan is one
    // kernel that verifies instead of two that fold (launch_scan_predicted).  Rows waiting for the copy launch need
    // their records: not with those.
    const bool shape_ok = c->pred_fold_esc == p.fused || !(c->last_num_mask & kEscNum);  // (no such rows: one shape)
    p.pred_scan = c->pred_scan && c->pred_valid && c->pred_tiles_valid && shape_ok &&
                  (p.direct || !(p.num_mask >> NUM_NFCOPY & 1u));
    // ... and so is the symbolic binning, inside the analysis kernel (no scatter kernel, its totals folded by the
    // predicted scan).  Rows that need a scratch slot from the scatter's prefix (global key sets; numeric-first rows
    // that are not placed directly) keep the scatter kernel.
    p.pred_sym = c->pred_sym && p.pred_scan && !(c->last_sym_mask >> SYM_GH & 1u) &&
                 (!(c->last_sym_mask >> SYM_NF & 1u) || p.direct);
    // ... and then nothing downstream needs what the analysis WRITES any more: the previous identical call left all of it
    // in the arena.  The analysis becomes a verifier beside the sequence (its own stream, joined in front of the ticket).
    // (on the library's own pipeline stream only: beside a CALLER's stream the verifier would not be ordered behind the
    //  work that produces the inputs there)
    p.overlap = c->overlap_analysis && p.pred_sym && c->vstream != nullptr && !c->use_user_stream;
    // ... and when the arena was last written by a replay of THIS sequence, its scan has nothing left to do either: the
    // offsets, classes and records it would produce are a function of the rows' nnz and of the analysis' quantities --
    // all verified where they are produced.  C.row_offsets is still rewritten in every call (from the sequence's copy of
    // the offsets, by extra workgroups of the numeric light launch: needs that launch).
    constexpr u32 kBigLight = (1u << NUM_D1) | (1u << NUM_B2K) | (1u << NUM_W512) | (1u << NUM_W256);
    // (... or has no rows that are finished early at all: every row then goes through the numeric light launch)
    const bool early_ok = (p.fused && p.direct) || (p.num_mask & (kEscNum | (1u << NUM_NFCOPY))) == 0;
    // In such a sequence the symbolic pass of a hash / dense row has ONE reader left: the comparison of its count with the
    // previous call's.  The numeric bodies can make that comparison themselves -- they count what their table holds before
    // they sort it -- if they stay inside a table that was sized by a nnz that may no longer hold (bounded probing) and
    // inside the row's room in C (numeric.hip, VERIFY; the spill chain of NUM_G sizes everything from what it counts in
    // the same call and compares in its copy kernel).  Then those rows are walked ONCE, as the register-class rows are: the
    // symbolic phase of the sequence is the fused launch of the register classes alone (nothing at all for an input
    // without such rows) -- and the heavy symbolic classes that kept a sequence from dropping its scan (below) are gone.
    // Only when it PAYS: the verifying bodies cost the numeric light launch 5-15 % (one more compare in its probing loop,
    // the hottest loop of the library), while the symbolic pass of a FEW hash rows beside many register-class rows hides
    // inside the fused launch (scircuit / mac_econ stand-ins: that launch got 1.5 us shorter, the numeric one 3 us
    // longer).  So: when the hash / dense rows are what the symbolic phase spends its time on (isolated per-row costs, as
    // for the stream forks) -- the nlpkkt stand-in, whose symbolic launch was a third of its multiply: 26.4 -> 19.4 ms.
    // (option num_verify = 2: whenever possible)
    float us_hash = 0.f, us_esc = 0.f;
    for (int k = 0; k < kMaxClasses; ++k)
        ((kSymEscMask >> k & 1u) ? us_esc : us_hash) += p.sym_counts[k] * kSymNsPerRow[k] * 1e-3f;
    const bool want_verify = c->num_verify && (p.fused || !(p.launch_mask & kEscNum)) && !(p.sym_mask >> SYM_NF & 1u) &&
                             (c->num_verify >= 2 || us_hash > us_esc);
    const u32 eff_sym = want_verify ? (p.sym_mask & kSymEscMask) : p.sym_mask;
    // (Only for sequences whose symbolic phase is the ONE light launch: with heavy symbolic classes on side streams the
    //  join of that phase would be followed by the fork of the numeric phase with no kernel in between, and a captured
    //  graph of that shape crashed the host inside the runtime every second run -- webbase stand-in, round 4; the same
    //  sequence enqueued launch by launch did not.  Those sequences keep their scan: it is 3 % of their multiply.)
    p.skip_scan = c->skip_scan && arena_replay_ok && p.overlap && early_ok && c->merge_light && !c->split_light &&
                  (p.launch_mask & kBigLight) != 0 && (eff_sym & ~kSymLightMask) == 0;
    p.num_verify = want_verify && p.skip_scan;
    // A sequence without a scan whose numeric phase FORKS (heavy classes on side streams: webbase stand-in) is not
    // captured: the executable graph of that sequence crashed the host inside the runtime in every second process that had
    // multiplied other problems on the config before (scripts/repro_standins.py; a segmentation fault inside
    // hipGraphLaunch / instantiate, not in any kernel -- the same family as the join / fork shape above), while the same
    // launches enqueued one by one at every call never did (eight processes in a row, and the four of
    // tests/test_gpu_parity.py::test_standins_take_turns_on_one_config_in_fresh_processes).  It costs such a multiply
    // nothing measurable (webbase stand-in 0.874-0.882 ms either way).  (option capture_forked = 1: capture anyway)
    p.uncaptured = p.skip_scan && (p.launch_mask & ~(kNumLightMask | (1u << NUM_NFCOPY))) != 0 && !c->capture_forked;
    return p;
}

// front + back + ticket of a replayed sequence, into a stream under capture -- or, with `tm`, straight onto the
// stream with HIP events around the launches (speck_config_profile_kernels + option profile_replay: the launches
// the replay consists of, timed one by one)
template <typename T>
int enqueue_replay(speck_config* c, hipStream_t s, const speck_dcsr* A, const speck_dcsr* B, const speck_dcsr* C,
                   const Scratch& sc, const ReplayPlan& p, Timing* tm, size_t* ev_num_end)
{
    c->capture_fused = p.fused;
    c->capture_direct = p.direct;
    c->capture_pred_scan = p.pred_scan;
    c->capture_pred_sym = p.pred_sym;
    c->capture_overlap = p.overlap;
    c->capture_skip_scan = p.skip_scan;
    c->capture_num_verify = p.num_verify;
    if (p.skip_scan) {  // C.row_offsets <- the sequence's own copy of the offsets (numeric light launch)
        c->stage_off_src = c->gpred.off;
        c->stage_off_dst = C->row_offsets;
        c->stage_off_n = (u32)A->rows + 1u;
    }
    c->capture_c_col = C->col_ids;
    c->capture_c_val = C->data;
    struct Reset {
        speck_config* c;
        u32 wcols;
        ~Reset()
        {
            c->capture_direct = c->capture_fused = c->capture_pred_scan = c->capture_pred_sym = c->capture_overlap = false;
            if (c->capture_skip_scan) c->stage_off_src = nullptr, c->stage_off_dst = nullptr, c->stage_off_n = 0;
            c->capture_skip_scan = c->capture_num_verify = false;
            c->nf_wcols = wcols;
        }
    } reset{c, c->nf_wcols};
    c->nf_wcols = p.nf_wcols;
    // (num_verify: the symbolic phase is the register classes alone -- the plan keeps every class, for the form of the
    //  sequence that runs when the arena is not this problem's)
    u32 sym_mask = p.sym_mask, sym_counts[kMaxClasses];
    std::memcpy(sym_counts, p.sym_counts, sizeof(sym_counts));
    if (p.num_verify) {
        sym_mask &= kSymEscMask;
        for (int k = 0; k < kMaxClasses; ++k)
            if (!(kSymEscMask >> k & 1u)) sym_counts[k] = 0;
    }
    int rc = enqueue_front(c, s, A, B, sc, C->row_offsets, (u32)sizeof(T), C->nnz, sym_mask,
                           p.num_mask, true, tm, sym_counts, nullptr,
                           p.g_products, p.num_counts[NUM_G], 3u, p.nf_cap_entries);
    if (rc != SPECK_OK) return rc;
    if (tm && c->gate_verifier) {  // (profiled pre-pass of a long sequence: the verifier's stream starts here, multiply_impl)
        HIP_TRY(hipEventRecord(c->fork, s));
        HIP_TRY(hipStreamWaitEvent(c->vstream, c->fork, 0));
        // (... and a moment later, as next to the graph: a verifier dispatched TOGETHER with the numeric launch takes half of
        //  the chip's wave slots before that launch's resident workgroups have them, and keeps them -- the nlpkkt stand-in's
        //  light launch then measured 20.0 ms against 18.7 in a trace of the graph)
        launch_delay(c->vstream, 30);
    }
    if (tm) {
        tm->ev_num = tm->ev;
        (void)hipEventRecord(kernel_event(c, tm->ev++), s);
    }
    rc = enqueue_back<T>(c, s, A, B, sc, C->row_offsets, C->col_ids, static_cast<T*>(C->data),
                         p.launch_mask, p.num_counts, tm);
    if (rc != SPECK_OK) return rc;
    if (tm) {
        *ev_num_end = tm->ev;
        (void)hipEventRecord(kernel_event(c, tm->ev++), s);
    }
    // no copy node: the last kernel mirrors the (final) statistics block into pinned host memory, then stores
    // the completion ticket
    launch_done(s, c->d_ticket, c->h_ticket_dev, c->d_stats, c->h_stats_dev);
    return SPECK_OK;
}

// Capture the sequence into one graph, specialised to `key`.
template <typename T>
int capture_graph(speck_config* c, hipStream_t s, const speck_dcsr* A, const speck_dcsr* B,
                  const speck_dcsr* C, const Scratch& sc, const GraphKey& key, bool arena_replay_ok = false)
{
    drop_graph(c);
    ReplayPlan plan = plan_replay(c, arena_replay_ok);
    if (snapshot_prediction(c, s, plan) != SPECK_OK) {  // no room for the sequence's own copy: predict nothing
        c->pred_valid = c->pred_tiles_valid = false;
        plan = plan_replay(c);
    }
    c->graph_plan = plan;
    c->graph_direct = plan.direct;
    c->graph_fused = plan.fused;
    c->graph_pred_scan = plan.pred_scan;
    c->graph_pred_sym = plan.pred_sym;
    c->graph_overlap = plan.overlap;
    if (c->use_user_stream || c->replay_uncaptured || plan.uncaptured) {
        // the launches of this sequence are enqueued one by one at every call (multiply_impl): only the plan and the
        // sequence's copy of the prediction are kept -- a caller's stream is never put into capture mode
        c->graph_key = key;
        c->graph_valid = true;
        ++c->graph_captures;
        return SPECK_OK;
    }
    HIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    const int rc = enqueue_replay<T>(c, s, A, B, C, sc, plan, nullptr, nullptr);
    hipGraph_t g = nullptr;
    hipError_t e2 = hipStreamEndCapture(s, &g);
    if (rc != SPECK_OK || e2 != hipSuccess || !g) {
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        return SPECK_ERR_HIP;
    }
    hipGraphExec_t ge = nullptr;
    if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        return SPECK_ERR_HIP;
    }
    c->graph = g;
    c->graph_exec = ge;
    c->graph_key = key;
    c->graph_valid = true;
    c->exec_dirty = false;
    ++c->graph_captures;
    return SPECK_OK;
}

// per-launch / per-phase times of the last call from its events (speck_stats)
void publish_kernel_times(speck_config* c, const Timing& tm, size_t ev_num_end)
{
    auto ms = [&](size_t a) {
        float v = 0.f;
        (void)hipEventElapsedTime(&v, c->kev[a], c->kev[a + 1]);
        return v;
    };
    (void)hipEventElapsedTime(&c->last.analysis_ms, c->kev[tm.ev_analysis], c->kev[tm.ev_analysis_end]);
    c->last.scan_ms = ms(tm.ev_scan);
    // phases: end of the analysis launches -> start of the scan (all symbolic branches joined);
    // before the first numeric launch -> after the last join
    (void)hipEventElapsedTime(&c->last.sym_phase_ms, c->kev[tm.ev_analysis_end], c->kev[tm.ev_scan]);
    (void)hipEventElapsedTime(&c->last.num_phase_ms, c->kev[tm.ev_num], c->kev[ev_num_end]);
    for (const auto& ct : tm.sym) {
        if (ct.cls == kLightBig) c->last.sym_light_ms = ms(ct.ev);
        else if (ct.cls == kLightTiny) c->last.sym_tiny_ms = ms(ct.ev);
        else c->last.sym_bin_ms[ct.cls] = ms(ct.ev);
    }
    for (const auto& ct : tm.num) {
        if (ct.cls == kLightBig) c->last.num_light_ms = ms(ct.ev);
        else if (ct.cls == kLightTiny) c->last.num_tiny_ms = ms(ct.ev);
        else c->last.num_bin_ms[ct.cls] = ms(ct.ev);
    }
    c->last.kernel_events_valid = 1;
}

// The analysis of a replayed sequence as a VERIFIER (ReplayPlan::overlap): on its own stream, enqueued by the host
// right behind the sequence, so that it runs beside it.  It compares what it computes from A and B as they are now with
// what the previous identical call left in the arena -- which is what the sequence's kernels read -- and reports to
// pinned host memory.  wait_verifier: the verdict, once that stream is idle (it is, long before the sequence's ticket).
int launch_verifier(speck_config* c, const speck_dcsr* A, const speck_dcsr* B, const Scratch& sc)
{
    __atomic_store_n(c->h_verify, 0u, __ATOMIC_RELEASE);
    if (c->verify_inputs && c->snap && c->snap_for_arena) {
        // the arena's metadata is a function of inputs that are still what the writing analysis saw: four streams compared
        launch_verify_inputs(c->vstream, A->row_offsets, sc.a_ro_copy, (u32)A->rows, A->col_ids, c->snap, A->nnz, B->row_offsets,
                             B->col_ids, (u32)B->rows, c->snap + c->snap_a_words, c->h_verify_dev);
        launch_ticket(c->vstream, c->d_vticket, c->h_verify_dev + 16);
        HIP_TRY(hipGetLastError());
        return SPECK_OK;
    }
    ClassifyParams cp = c->cp;
    cp.sym_allowed = cp.num_allowed = 0xFFFFFFFFu;
    cp.esc16 = (c->cp.esc16 && B->cols <= (1ull << 26)) ? 1u : 0u;  // (as enqueue_front classifies)
    cp.esc_fused = 0;
    launch_analysis(c->vstream, A->row_offsets, A->col_ids, B->row_offsets, B->col_ids, (u32)A->rows, A->nnz, sc.row_ops,
                    sc.row_max_ops, sc.row_col_min, sc.row_col_max, sc.cls_sym, sc.counts, sc.partials, sc.recs_sym,
                    c->d_stats, cp, sc.b_sl, nullptr, sc.nf_off, ~0ull, (u32)B->rows, nullptr, nullptr, nullptr,
                    (u32)B->cols, B->nnz, 0u, sc.a_ro_copy, c->h_verify_dev);
    // ... and behind it the copy of the inputs it has just verified the arena against (they do not change while the call
    // is in flight): the verifiers of the next replays compare with that.  (The FIRST replay of a problem pays the
    // recomputing verifier once; the eager call that precedes it pays nothing.)
    if (c->verify_inputs && c->snap) {
        launch_snapshot_inputs(c->vstream, A->row_offsets, A->col_ids, c->snap, A->nnz, B->row_offsets, B->col_ids, (u32)B->rows,
                               c->snap + c->snap_a_words);
        c->snap_pending = true;
    }
    launch_ticket(c->vstream, c->d_vticket, c->h_verify_dev + 16);
    HIP_TRY(hipGetLastError());
    return SPECK_OK;
}
int wait_verifier(speck_config* c, bool* changed)
{
    // (the verifier finishes long before the sequence: by the time the host looks its ticket is there -- no API call)
    const u32 want = ++c->vticket_expected;
    const auto t0 = std::chrono::steady_clock::now();
    bool seen = false;
    while (!(seen = __atomic_load_n(c->h_verify + 16, __ATOMIC_ACQUIRE) == want)) {
        for (int i = 0; i < 64; ++i) cpu_relax();
        if (std::chrono::steady_clock::now() - t0 > kSpinBudget) break;
    }
    if (!seen) HIP_TRY(hipStreamSynchronize(c->vstream));
    c->vticket_expected = __atomic_load_n(c->h_verify + 16, __ATOMIC_ACQUIRE);
    *changed = __atomic_load_n(c->h_verify, __ATOMIC_ACQUIRE) != 0u;
    return SPECK_OK;
}

// The input check of an eager call, beside it on the verifier's stream (stages.hip: validate_b_kernel).
int begin_validate(speck_config* c, const speck_dcsr* B)
{
    __atomic_store_n(c->h_verify, 0u, __ATOMIC_RELEASE);
    if (c->use_user_stream) {
        // the caller's stream may still be producing B: the check goes behind what is queued there now
        HIP_TRY(hipEventRecord(c->fork, c->user_stream));
        HIP_TRY(hipStreamWaitEvent(c->vstream, c->fork, 0));
    }
    launch_validate_b(c->vstream, B->row_offsets, B->col_ids, (u32)B->rows, (u32)B->cols, B->nnz, c->h_verify_dev);
    launch_ticket(c->vstream, c->d_vticket, c->h_verify_dev + 16);
    HIP_TRY(hipGetLastError());
    c->validate_in_flight = true;
    return SPECK_OK;
}
// ... its verdict (waits for the check if it is still running: it is not, by the time anybody asks)
int finish_validate(speck_config* c, bool* b_invalid)
{
    *b_invalid = false;
    if (!c->validate_in_flight) return SPECK_OK;
    c->validate_in_flight = false;
    bool any = false;
    const int rc = wait_verifier(c, &any);
    if (rc != SPECK_OK) return rc;
    *b_invalid = (__atomic_load_n(c->h_verify, __ATOMIC_ACQUIRE) & 4u) != 0;
    return SPECK_OK;
}

template <typename T>
int multiply_impl(speck_config* c, const speck_dcsr* A, const speck_dcsr* B, speck_dcsr* C,
                  speck_timings* t)
{
    if (!c || !C) return SPECK_ERR_INVALID;
    int rc = check_inputs(A, B);
    if (rc != SPECK_OK) return rc;
    HIP_TRY(hipSetDevice(c->device));
    speck_timings local_t{};
    if (!t) t = &local_t;
    c->last = speck_stats{};

    // reference: source/GPU/Multiply.cu:67-70
    if (A->nnz == 0 || B->nnz == 0) {
        C->nnz = 0;
        return SPECK_OK;
    }
    const u32 m = (u32)A->rows;
    hipStream_t s = main_stream(c);

    if (t->measureCompleteTime) HIP_TRY(hipEventRecord(c->completeStart, s));
    auto finish_complete = [&]() -> int {
        if (t->measureCompleteTime) {
            // reference: cudaDeviceSynchronize + complete event, Multiply.cu:1082-1085
            HIP_TRY(hipEventRecord(c->completeEnd, s));
            HIP_TRY(hipEventSynchronize(c->completeEnd));
            HIP_TRY(hipEventElapsedTime(&t->complete, c->completeStart, c->completeEnd));
        }
        return SPECK_OK;
    };
    StageTimer st(c, t->measureAll != 0, s);
    struct RestoreFlag {
        int& flag;
        int value;
        ~RestoreFlag() { flag = value; }
    } restore_profile{c->profile_kernels, c->profile_kernels};
    if (t->measureAll) c->profile_kernels = 1;  // per-stage times come from per-launch events

    rc = ensure_arena(c, scratch_bytes(m, A->nnz));
    if (rc != SPECK_OK) return rc;
    Scratch sc = carve(c, m, A->nnz);
    c->snap_pending = false;
    if (c->verify_inputs && c->overlap_analysis && c->use_graph) ensure_snap(c, A->nnz, B->rows);

    // ------------------------------------------------------------------ replay path
    // Same buffers as a previous call, C already allocated for the expected nnz: replay the
    // captured launch sequence.  The device checks the two assumptions baked into it (nnz(C)
    // unchanged, no row in a class that was pruned); on a miss the eager path below re-runs.
    const bool c_ready = C->rows == A->rows && C->row_offsets && C->col_ids && C->data && C->nnz > 0;
    if (c->use_graph && c_ready && c->profile_kernels && c->profile_replay && !t->measureAll) {
        // the launches a replay of this call consists of, straight onto the stream with events around them
        const GraphKey key = make_key<T>(c, A, B, C, s);
        if (c->last_key_valid && c->last_key == key) {
            c->exec_dirty = true;
            if (c->graph_valid && !(c->graph_key == key)) drop_graph(c);  // (its copy of the prediction is rewritten below)
            const bool arena_mine = c->arena_key_valid && c->arena_key == key;
            ReplayPlan plan = plan_replay(c, arena_mine && c->arena_from_replay);
            if (snapshot_prediction(c, s, plan) != SPECK_OK) {
                c->pred_valid = c->pred_tiles_valid = false;
                plan = plan_replay(c);
            }
            // (a verifying analysis needs the metadata of THIS problem in the arena)
            if (!arena_mine) plan.overlap = plan.skip_scan = plan.num_verify = false;
            c->arena_key_valid = false;
            Timing tm;
            size_t ev_num_end = 0;
            // (the verifier FIRST here: enqueuing the launches one by one with events around them takes the host longer
            //  than the first of them runs -- launched behind them the verifier would run beside the LAST launch, not
            //  beside the first as it does next to the graph)
            // (the same sequence once WITHOUT events directly in front: the timed launches then start on a chip that is
            //  busy and warm, as every replay of the graph behind another one does -- after the idle gap of a host-side
            //  synchronisation the first launch of the sequence measured ~5 % longer than its average in a kernel trace)
            rc = enqueue_replay<T>(c, s, A, B, C, sc, plan, nullptr, nullptr);
            if (rc != SPECK_OK) return rc;
            // (... and the verifier not before the TIMED sequence starts: with launches of milliseconds -- nlpkkt stand-in --
            //  the host has both sequences enqueued long before the first has run, and the verifier would spend itself
            //  beside the untimed one: the timed light launch measured 16.9 ms against 18.7 in a trace of the graph)
            //  (only then: beside a sequence of tens of microseconds the host launches the verifier ~30 us AFTER the graph,
            //   and a verifier that starts WITH the timed sequence made its first launch 5 % longer than a trace of the
            //   graph shows it -- scircuit stand-in 45.5 against 43.2 us)
            //  (... and behind the symbolic phase of the timed sequence: next to the graph the host launches the verifier ~30 us
            //   after the graph -- the short fused launch of such an input is through by then)
            c->gate_verifier = plan.overlap && c->last_eager_stats.sum_products >= (1ull << 29);
            rc = enqueue_replay<T>(c, s, A, B, C, sc, plan, &tm, &ev_num_end);
            c->gate_verifier = false;
            if (rc != SPECK_OK) return rc;
            // (BEHIND the sequence, as the graph path does: launched in front of it the verifier ran beside the first launch
            //  of the sequence from its start and made that launch ~2 us longer than it is inside the graph -- the launch
            //  durations of this path are the ones the bench line and scripts/check_launch_ms.py quote)
            if (plan.overlap) {
                rc = launch_verifier(c, A, B, sc);
                if (rc != SPECK_OK) return rc;
            }
            HIP_TRY(hipStreamSynchronize(s));
            c->ticket_expected = __atomic_load_n(c->h_ticket, __ATOMIC_ACQUIRE);
            bool changed = false;
            if (plan.overlap) {
                rc = wait_verifier(c, &changed);
                if (rc != SPECK_OK) return rc;
            }
            if (!changed && !c->h_stats->capacity_miss && !c->h_stats->nnz_overflow && c->h_stats->nnz_c == C->nnz) {
                c->arena_key = key;
                c->arena_key_valid = true;
                c->arena_from_replay = true;
                if (c->snap_pending) c->snap_for_arena = true;  // (the copy of the inputs this call's verifier took: launch_verifier)
                publish_counts(c);
                publish_kernel_times(c, tm, ev_num_end);
                c->last.replayed = 0;  // (not served by the graph: graph_replays does not count it)
                c->last.nf_direct = plan.direct ? 1 : 0;
                c->last.esc_fused = plan.fused ? 1 : 0;
                c->last.pred_stages = (plan.pred_scan ? 1 : 0) | (plan.pred_sym ? 2 : 0) | (plan.overlap ? 4 : 0) | (plan.skip_scan ? 8 : 0) |
                                      (plan.num_verify ? 16 : 0);
                return finish_complete();
            }
        }
    }
    if (c->use_graph && c_ready && !c->profile_kernels && !t->measureAll) {
        const GraphKey key = make_key<T>(c, A, B, C, s);
        bool have = c->graph_valid && c->graph_key == key;
        const bool replay_layout = c->arena_key_valid && c->arena_key == key && c->arena_from_replay;
        if (!have && c->last_key_valid && c->last_key == key)
            have = capture_graph<T>(c, s, A, B, C, sc, key, replay_layout) == SPECK_OK;
        // the sequence has replayed once: from now on it needs no scan kernel (captured anew, once)
        else if (have && !c->graph_plan.skip_scan && replay_layout && c->last_key_valid && c->last_key == key &&
                 plan_replay(c, true).skip_scan)
            have = capture_graph<T>(c, s, A, B, C, sc, key, true) == SPECK_OK;
        if (have && c->exec_dirty && !(c->replay_uncaptured || c->use_user_stream || c->graph_plan.uncaptured)) {
            // (see below) a fresh executable for a graph that other launches have passed; no executable: eager path
            if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
            c->graph_exec = nullptr;
            if (hipGraphInstantiate(&c->graph_exec, c->graph, nullptr, nullptr, 0) != hipSuccess) {
                (void)hipGetLastError();
                drop_graph(c);
                have = false;
            }
            c->exec_dirty = false;
        }
        if (have) {
            // Launching the SAME executable graph again after other launches went onto the same stream in between
            // (an eager multiply of another problem on this config) ended in GPU memory faults on this runtime --
            // gone with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, i.e. the runtime's pre-recorded launch packets; found by
            // the interleaved stress (tests/tools/stress_gpu.py interleave=K).  So: an executable that has seen other
            // work on the pipeline stream since its last launch is instantiated afresh from the captured graph (a few
            // hundred us, once per switch between problems); on a CALLER's stream, whose traffic the library cannot
            // see, the launches of the sequence are enqueued one by one instead (option replay_uncaptured does the same
            // everywhere: +1..6 % per multiply).
            // A sequence whose analysis only VERIFIES reads the metadata the previous multiply of THIS problem left in the
            // arena.  If something else has used the arena since (another problem on this config, a stage entry point) the
            // same sequence runs with a writing analysis in front instead -- enqueued, not from the graph -- and leaves
            // the arena as the next replay needs it.
            const bool arena_ok = c->arena_key_valid && c->arena_key == key;
            bool overlapped = c->graph_overlap, skipped = c->graph_plan.skip_scan;
            const bool self_verified = c->graph_plan.num_verify;
            c->arena_key_valid = false;  // (until this call has completed)
            if ((c->graph_overlap && !arena_ok) || (c->graph_plan.skip_scan && !replay_layout)) {
                ReplayPlan p2 = c->graph_plan;
                p2.overlap = p2.skip_scan = p2.num_verify = false;
                overlapped = skipped = false;
                c->exec_dirty = true;
                rc = enqueue_replay<T>(c, s, A, B, C, sc, p2, nullptr, nullptr);
                if (rc != SPECK_OK) return rc;
            } else if (c->replay_uncaptured || c->use_user_stream || c->graph_plan.uncaptured) {
                rc = enqueue_replay<T>(c, s, A, B, C, sc, c->graph_plan, nullptr, nullptr);
                if (rc != SPECK_OK) return rc;
            } else {
                HIP_TRY(hipGraphLaunch(c->graph_exec, s));
            }
            // (behind the sequence, while it runs: the host would only spin otherwise)
            if (overlapped) {
                rc = launch_verifier(c, A, B, sc);
                if (rc != SPECK_OK) return rc;
            }
            // the last node of the sequence stores a ticket into pinned memory
            rc = wait_ticket(c, s);
            if (rc != SPECK_OK) return rc;
            bool changed = false;
            if (overlapped) {
                rc = wait_verifier(c, &changed);
                if (rc != SPECK_OK) return rc;
            }
            if (!changed && !c->h_stats->capacity_miss && !c->h_stats->nnz_overflow && c->h_stats->nnz_c == C->nnz) {
                c->arena_key = key;
                c->arena_key_valid = true;
                c->arena_from_replay = true;
                if (c->snap_pending) c->snap_for_arena = true;  // (the copy of the inputs this call's verifier took: launch_verifier)
                ++c->graph_replays;
                publish_counts(c);
                c->last.replayed = 1;
                c->last.nf_direct = c->graph_direct ? 1 : 0;
                c->last.esc_fused = c->graph_fused ? 1 : 0;
                c->last.pred_stages = (c->graph_pred_scan ? 1 : 0) | (c->graph_pred_sym ? 2 : 0) | (overlapped ? 4 : 0) | (skipped ? 8 : 0) |
                                      ((skipped && self_verified) ? 16 : 0);
                return finish_complete();
            }
            ++c->graph_misses;  // inputs changed under the same pointers: fall through
            drop_graph(c);
        }
    }

    // ------------------------------------------------------------------ eager path
    c->exec_dirty = true;  // (a captured sequence of another problem must not be launched again as it is)
    c->arena_key_valid = false;  // (the arena is rewritten: it is this problem's once the call has completed)
    // INIT: C.row_offsets reuse rule (Multiply.cu:156-165)
    u32* c_ro = nullptr;
    bool own_ro = false;
    if (C->rows == A->rows && C->row_offsets != nullptr) {
        c_ro = C->row_offsets;
    } else {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c_ro), (size_t(m) + 1) * sizeof(u32)));
        own_ro = true;
    }
    auto fail = [&](int code) {
        if (own_ro) (void)hipFree(c_ro);
        return code;
    };
    t->init = st.lap();

    // ANALYSIS + binning + SYMBOLIC + SCAN (Multiply.cu:239-575) -- one read-back
    Timing tm;
    u32 sym_now[kMaxClasses];
    const u32* sym_known = nullptr;  // rows per symbolic class, once a read-back of this call has them
    // what this call leaves behind for a replay of itself (Prediction): its row offsets and tile tables, written by
    // the scan kernel next to its other outputs.  No room on the device: the replay predicts nothing.
    c->pred_valid = c->pred_tiles_valid = false;
    const bool keep_pred = (c->nf_direct || c->pred_scan) && c->use_graph && ensure_pred(c->pred, m);
    // (the shape of the tile tables: will that replay finish the register-class rows in its symbolic phase?)
    const bool fold_esc = keep_pred && c->esc_fused && c->nf_direct && c->cp.sym_g8 == c->cp.num_g8 && c->merge_light;
    // (the scan mirrors the statistics and stores the ticket itself: await_scan_stats)
    const bool early_stats = c->spin_wait && !c->profile_kernels;
    auto front = [&](u32 parts) {
        // (the offsets go to scratch: C.row_offsets -- possibly the caller's reused buffer -- is written only once
        //  nothing can fail any more)
        return enqueue_front(c, s, A, B, sc, sc.offsets, (u32)sizeof(T), ~0ull, kAllSym, kAllNum, true, &tm,
                             parts == 2u ? sym_known : nullptr, early_stats ? c->h_stats_dev : nullptr, ~0ull, ~0u, parts, ~0ull,
                             keep_pred ? &c->pred : nullptr, fold_esc);
    };
    // ONE batch, sized from the previous eager call on this config (same shapes): the classes that call had rows in
    // (a row in any other class raises capacity_miss: publish_bins), grids from its counts, its scratch pool and
    // numeric-first window (checked by the scatter / numeric-first kernels).  Saves the first of the two read-backs --
    // a ticket, a spin and the launch latency behind it, ~15 us of a 170 us multiply.
    // The input check -- B's rows strictly ascending and in range: one coalesced pass over B.col_ids -- runs BESIDE the
    // call on the verifier's stream (it rode in the analysis launch until round 4 and cost that launch 13 us); its
    // verdict is looked at with the statistics of the scan, before anything of C is written.  Until then the kernels
    // stay inside their tables and windows whatever B holds.  A's column ids are checked -- and clamped -- by the analysis.
    struct ValidateGuard {  // (no way out of this call leaves the check running: the next call reuses its verdict word)
        speck_config* c;
        ~ValidateGuard()
        {
            bool ignored;
            (void)finish_validate(c, &ignored);
        }
    } validate_guard{c};
    // (launched BEHIND the analysis of the call, not in front of it: two launches on another stream cost the host
    //  5-10 us that the analysis -- the head of the call's critical path -- then starts later; enqueue_front calls it)
    bool validate_started = false;
    auto start_validate = [&]() -> int {
        if (!c->validate_inputs || validate_started) return SPECK_OK;
        validate_started = true;
        return begin_validate(c, B);
    };
    struct HookGuard {
        speck_config* c;
        ~HookGuard() { c->after_analysis = nullptr; }
    } hook_guard{c};
    c->after_analysis = start_validate;
    auto b_is_invalid = [&](bool* bad) { return finish_validate(c, bad); };
    bool speculated = false;
    u32 spec_counts[kMaxClasses];
    if (c->eager_speculate && c->spec_valid && c->spec_rows_a == A->rows && c->spec_rows_b == B->rows &&
        (c->cp.nf_min_ops || c->cp.gh_per_window)) {
        u32 mask = 0;
        for (int k = 0; k < kMaxClasses; ++k) {
            spec_counts[k] = c->last_sym_counts[k] ? c->last_sym_counts[k] + c->last_sym_counts[k] / 4 + 64 : 0;
            if (spec_counts[k]) mask |= 1u << k;
        }
        // (the light launch is one kernel whatever it holds: every class of it may have rows)
        for (int k = 0; k < kMaxClasses; ++k)
            if ((kSymLightMask >> k & 1u) && !spec_counts[k]) spec_counts[k] = 256;
        mask |= kSymLightMask;
        rc = enqueue_front(c, s, A, B, sc, sc.offsets, (u32)sizeof(T), ~0ull, mask, kAllNum, true, &tm, spec_counts,
                           early_stats ? c->h_stats_dev : nullptr, ~0ull, ~0u, 3u, c->nf_cap_entries,
                           keep_pred ? &c->pred : nullptr, fold_esc, false);
        if (rc == SPECK_OK) rc = early_stats ? await_scan_stats(c, s) : read_stats(c, s);
        if (rc != SPECK_OK) return fail(rc);
        if (c->h_stats->a_invalid) return fail(SPECK_ERR_INVALID);
        bool b_bad = false;
        rc = b_is_invalid(&b_bad);
        if (rc != SPECK_OK) return fail(rc);
        if (b_bad) return fail(SPECK_ERR_UNSORTED);
        speculated = !c->h_stats->capacity_miss;
        ++(speculated ? c->eager_spec_hits : c->eager_spec_misses);
        c->last.eager_speculated = speculated ? 1 : -1;
    }
    if (!speculated) {
    // analysis + binning
    rc = front(1u);
    if (rc != SPECK_OK) return fail(rc);
    if (c->cp.nf_min_ops || c->cp.gh_per_window) {
        // numeric-first rows (and the global key sets of SYM_GH rows) need their scratch pool before the
        // symbolic phase: one more read-back
        // (the replayed sequence has none: the pool of the previous identical call is checked on the device)
        rc = read_stats(c, s);
        if (rc != SPECK_OK) return fail(rc);
        if (c->h_stats->a_invalid) return fail(SPECK_ERR_INVALID);
        // No room (or no budget) for the pool: first the global key sets go (those rows take the multi-window
        // bitmap, which needs no memory), then the numeric-first rows (they take the two-phase path) -- for this
        // config from now on; the rows are classified again.
        while (c->h_stats->nf_entries) {
            rc = ensure_nfpool(c, c->h_stats->nf_entries, sizeof(T));
            if (rc != SPECK_ERR_OOM) break;
            if (c->cp.gh_per_window && c->h_stats->sym.count[SYM_GH]) c->cp.gh_per_window = 0;
            else if (c->cp.nf_min_ops && c->h_stats->sym.count[SYM_NF]) c->cp.nf_min_ops = 0;
            else break;
            ++c->pool_fallbacks;
            rc = front(1u);
            if (rc == SPECK_OK) rc = read_stats(c, s);
            if (rc != SPECK_OK) break;
        }
        if (rc != SPECK_OK) return fail(rc);
        c->nf_wcols = c->h_stats->nf_max_range ? c->h_stats->nf_max_range : kNumD1Cols;
        // the read-back also buys right-sized grids, fork decisions and list positions for the symbolic launches
        std::memcpy(sym_now, c->h_stats->sym.count, sizeof(sym_now));
        sym_known = sym_now;
    }
    // (with both scratch-pool classes off the symbolic kernels run before the host has seen the verdict of the
    //  check: they stay inside their tables and windows whatever B holds, and nothing of C is written before the
    //  read-back below)
    rc = front(2u);
    if (rc != SPECK_OK) return fail(rc);
    rc = early_stats ? await_scan_stats(c, s) : read_stats(c, s);
    if (rc != SPECK_OK) return fail(rc);
    if (c->h_stats->a_invalid) return fail(SPECK_ERR_INVALID);
    bool b_bad = false;
    rc = b_is_invalid(&b_bad);
    if (rc != SPECK_OK) return fail(rc);
    if (b_bad) return fail(SPECK_ERR_UNSORTED);
    }  // !speculated
    t->countProducts = 0.f;
    t->loadBalanceCounting = 0.f;
    t->globalMapsCounting = 0.f;
    t->spGEMMCounting = st.lap();  // analysis + binning + symbolic + scan are one async batch
    publish_counts(c);
    if (c->h_stats->nnz_overflow) return fail(SPECK_ERR_NNZ_OVERFLOW);
    const u64 nnz_c = c->h_stats->nnz_c;

    if (c->h_stats->sum_products == 0) {
        // reference: Multiply.cu:256-261 -> matOut.alloc(rows, cols, 0, false)
        if (own_ro) (void)hipFree(c_ro);
        speck_dcsr_free(C);
        C->rows = A->rows;
        C->cols = B->cols;
        C->nnz = 0;
        c->last_key_valid = false;
        return finish_complete();
    }

    // Spill pool of the NUM_G rows FIRST: every failure up to here leaves C untouched (header contract).
    const u32 num_mask = mask_of(c->h_stats->num.count, NUM_CLASSES);
    if (num_mask >> NUM_G & 1u) {
        // global-memory spill buffers of the heavy rows (role of the reference's global maps,
        // Multiply.cu:357-427), grow-only: per-row plan, per-bucket counters, and two product
        // pools sized from the EXACT product count of the NUM_G rows (numeric.hip, NUM_G)
        const u64 pg = c->h_stats->g_products;
        const u32 rows_g = c->h_stats->num.count[NUM_G];
        const u64 buckets = pg / kGBucketTarget + rows_g + 16;
        const u64 cells = u64(kGCellsPerBucket) * buckets;
        if (cells > 0x7FFFFFFFull) return fail(SPECK_ERR_OOM);
        const size_t need = Carver::need(rows_g, sizeof(GRowPlan)) + Carver::need(cells + 3 * buckets + 4, 4) +
                            Carver::need(buckets, 8) +
                            Carver::need(cells, 4) + Carver::need(buckets, 8) + 2 * Carver::need(buckets, 4) +
                            2 * Carver::need(pg, 4) + 2 * Carver::need(pg, sizeof(T)) + 4096;
        if (need > c->gpool_bytes) {
            drop_graph(c);
            if (c->gpool) (void)hipFree(c->gpool);
            c->gpool = nullptr;
            c->gpool_bytes = 0;
            if (hipMalloc(&c->gpool, need + need / 8) != hipSuccess) {
                (void)hipGetLastError();
                return fail(SPECK_ERR_OOM);
            }
            c->gpool_bytes = need + need / 8;
        }
        Carver cv(c->gpool);
        SpillBuffers sp{};
        sp.plan = cv.take<GRowPlan>(rows_g);
        // fcount | bcount | bcursor | dcount are cleared by ONE memset: keep them back to back
        u32* counters = cv.take<u32>(cells + 3 * buckets + 4);
        sp.fcount = counters;
        sp.bcount = counters + cells;
        sp.bcursor = sp.bcount + buckets;
        sp.dcount = sp.bcursor + buckets;
        sp.big_count = sp.dcount + buckets;
        sp.big_list = cv.take<u64>(buckets);
        sp.fmap = cv.take<u32>(cells);
        sp.bstart = cv.take<u64>(buckets);
        sp.clo = cv.take<u32>(buckets);
        sp.chi = cv.take<u32>(buckets);
        sp.pcol[0] = cv.take<u32>(pg);
        sp.pcol[1] = cv.take<u32>(pg);
        sp.pval[0] = cv.take<T>(pg);
        sp.pval[1] = cv.take<T>(pg);
        sp.bucket_cap = (u32)buckets;
        sp.cell_cap = (u32)cells;
        if (sp.plan != c->spill.plan || sp.pcol[0] != c->spill.pcol[0] || sp.pval[1] != c->spill.pval[1] ||
            sp.bucket_cap != c->spill.bucket_cap)
            drop_graph(c);  // a captured sequence holds the old layout
        c->spill = sp;
    }
    t->globalMapsNumeric = st.lap();
    // ALLOC C: only when nnz changed (Multiply.cu:589-602)
    void* c_val = C->data;
    u32* c_col = C->col_ids;
    if (C->nnz != nnz_c || !c_val || !c_col) {
        void* nv = nullptr;
        u32* nc = nullptr;
        hipError_t e1 = hipMalloc(&nv, std::max<size_t>(nnz_c, 1) * sizeof(T));
        hipError_t e2 = e1 == hipSuccess
                            ? hipMalloc(reinterpret_cast<void**>(&nc), std::max<size_t>(nnz_c, 1) * 4)
                            : e1;
        if (e1 != hipSuccess || e2 != hipSuccess) {
            if (nv) (void)hipFree(nv);
            (void)hipGetLastError();
            return fail(SPECK_ERR_OOM);
        }
        if (C->data) (void)hipFree(C->data);
        if (C->col_ids) (void)hipFree(C->col_ids);
        if (C->row_offsets && C->row_offsets != c_ro) (void)hipFree(C->row_offsets);
        c_val = nv;
        c_col = nc;
    } else if (C->row_offsets && C->row_offsets != c_ro) {
        (void)hipFree(C->row_offsets);
    }
    // publish (Multiply.cu:1116-1121); from here C owns c_ro
    C->rows = A->rows;
    C->cols = B->cols;
    C->nnz = nnz_c;
    C->data = c_val;
    C->col_ids = c_col;
    C->row_offsets = c_ro;
    own_ro = false;
    // the staged offsets -> C.row_offsets: by extra workgroups of the numeric light launch when there is one with the
    // big kernel (launch_numeric_light), else by a copy of its own
    constexpr u32 kBigLight = (1u << NUM_D1) | (1u << NUM_B2K) | (1u << NUM_W512) | (1u << NUM_W256);
    const bool ride = c->merge_light && (num_mask & kBigLight) != 0;
    struct Unstage {
        speck_config* c;
        ~Unstage() { c->stage_off_src = nullptr, c->stage_off_dst = nullptr, c->stage_off_n = 0; }
    } unstage{c};
    if (ride) {
        c->stage_off_src = sc.offsets;
        c->stage_off_dst = c_ro;
        c->stage_off_n = m + 1;
    } else
        HIP_TRY(hipMemcpyAsync(c_ro, sc.offsets, (size_t(m) + 1) * sizeof(u32), hipMemcpyDeviceToDevice, s));
    t->allocC = st.lap();
    t->loadBalanceNumeric = 0.f;


    // NUMERIC (Multiply.cu:835-1014) + in-kernel sort (Multiply.cu:1028-1043)
    c->last_g_products = c->h_stats->g_products;
    if (c->profile_kernels) {
        tm.ev_num = tm.ev;
        (void)hipEventRecord(kernel_event(c, tm.ev++), s);
    }
    rc = enqueue_back<T>(c, s, A, B, sc, c_ro, c_col, static_cast<T*>(c_val), num_mask,
                         c->h_stats->num.count, &tm);
    if (rc != SPECK_OK) return rc;
    const size_t ev_num_end = tm.ev;
    if (c->profile_kernels) (void)hipEventRecord(kernel_event(c, tm.ev++), s);
    // The reference may return before its kernels finish when measureCompleteTime is off
    // (Multiply.cu:1082-1085) and relies on blocking streams to order later copies.  The
    // pipeline streams here are non-blocking, so the call always returns with C complete.
    if (c->spin_wait && !c->profile_kernels) {
        launch_done(s, c->d_ticket, c->h_ticket_dev, c->d_stats, c->h_stats_dev);
        rc = wait_ticket(c, s);
        if (rc != SPECK_OK) return rc;
    } else {
        HIP_TRY(hipStreamSynchronize(s));
    }
    t->spGEMMNumeric = st.lap();
    t->sorting = 0.f;  // sorting is fused into the numeric kernels
    t->cleanup = 0.f;  // nothing to free: the arena persists

    // remember what this call ran on: an identical next call is captured and replayed
    c->last_sym_mask = mask_of(c->h_stats->sym.count, SYM_CLASSES);
    c->last_num_mask = num_mask;
    c->last_max_row_nnz = c->h_stats->max_row_nnz_c;
    std::memcpy(c->last_sym_counts, c->h_stats->sym.count, sizeof(c->last_sym_counts));
    std::memcpy(c->last_num_counts, c->h_stats->num.count, sizeof(c->last_num_counts));
    c->last_key = make_key<T>(c, A, B, C, s);
    c->last_key_valid = true;
    c->arena_key = c->last_key;
    c->arena_key_valid = true;
    c->arena_from_replay = false;
    c->spec_valid = true;
    c->spec_rows_a = A->rows;
    c->spec_rows_b = B->rows;
    // ... and where every row went: the scan kernel wrote the prediction (offsets, tile tables) as it went
    c->pred_valid = c->pred_tiles_valid = keep_pred;
    c->pred_fold_esc = fold_esc;
    c->last_eager_stats = *c->h_stats;

    rc = finish_complete();
    if (rc != SPECK_OK) return rc;

    if (c->profile_kernels) {
        HIP_TRY(hipStreamSynchronize(s));
        auto span = [&](size_t a, size_t b) {
            float v = 0.f;
            (void)hipEventElapsedTime(&v, c->kev[a], c->kev[b]);
            return v;
        };
        publish_kernel_times(c, tm, ev_num_end);
        if (t->measureAll) {
            // the reference's eleven stage fields (Timings.h:7-18, filled at Multiply.cu:227-1073) from
            // the kernel events: stages that are fused here report under the field of their role
            t->countProducts = span(tm.ev_analysis, tm.ev_between);          // readOperations
            t->loadBalanceCounting = span(tm.ev_between, tm.ev_analysis_end);  // symbolic binning (scatter)
            t->spGEMMCounting = c->last.sym_phase_ms;                          // symbolic launches
            t->loadBalanceNumeric = c->last.scan_ms;                           // scan + numeric binning
            t->spGEMMNumeric = c->last.num_phase_ms;                           // numeric launches ...
            t->sorting = 0.f;                                                  // ... which sort in-kernel
        }
    }
    if (t->measureAll) {
        // same table the reference prints (Multiply.cu:1097-1113)
        std::printf("spECK     initial mallocs = %f ms\n", t->init);
        std::printf("spECK  count computations = %f ms\n", t->countProducts);
        std::printf("spECK       load-balancer = %f ms\n", t->loadBalanceCounting);
        std::printf("spECK      GlobalMaps Cnt = %f ms\n", t->globalMapsCounting);
        std::printf("spECK     counting kernel = %f ms\n", t->spGEMMCounting);
        std::printf("spECK        malloc mat C = %f ms\n", t->allocC);
        std::printf("spECK   num load-balancer = %f ms\n", t->loadBalanceNumeric);
        std::printf("spECK     init GlobalMaps = %f ms\n", t->globalMapsNumeric);
        std::printf("spECK      numeric kernel = %f ms\n", t->spGEMMNumeric);
        std::printf("spECK      Sorting kernel = %f ms\n", t->sorting);
        std::printf("spECK             cleanup = %f ms\n", t->cleanup);
        std::printf("--------------------------------------------------------------\n");
    }
    return SPECK_OK;
}

// speck_dcsr_copy, entirely on the device: the source may be a row-range view whose offsets are absolute -- its first
// offset is read HERE, not on the host (no device-to-host copy anywhere in the conversion).
// words: col_ids as u32, values as u32 pairs / singles (vwords = value_size / 4)
__global__ __launch_bounds__(256) void dcsr_copy_kernel(const u32* __restrict__ s_ro, const u32* __restrict__ s_col,
                                                        const u32* __restrict__ s_val, u32* __restrict__ d_ro,
                                                        u32* __restrict__ d_col, u32* __restrict__ d_val, u64 rows, u64 nnz,
                                                        u32 vwords)
{
    const u32 base = rows ? s_ro[0] : 0u;
    const u64 tid = u64(blockIdx.x) * 256 + threadIdx.x, nth = u64(gridDim.x) * 256;
    for (u64 i = tid; i <= rows; i += nth) d_ro[i] = rows ? s_ro[i] - base : 0u;
    for (u64 i = tid; i < nnz; i += nth) d_col[i] = s_col[u64(base) + i];
    const u64 nv = nnz * vwords;
    const u32* sv = s_val + u64(base) * vwords;
    for (u64 i = tid; i < nv; i += nth) d_val[i] = sv[i];
}

}  // namespace

extern "C" {

int speck_config_create(int device, speck_config** out)
{
    if (!out) return SPECK_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device >= n) {
        (void)hipGetLastError();
        return SPECK_ERR_NO_DEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    auto* c = new speck_config();
    c->device = device;
    c->sm = prop.multiProcessorCount;
    c->max_static_lds = (int)prop.sharedMemPerBlock;
    c->max_dynamic_lds = (int)std::max(prop.sharedMemPerBlockOptin, prop.sharedMemPerBlock);
    for (int i = 0; i < 6; ++i) {
        hipStream_t s;
        HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        c->streams.push_back(s);
    }
    HIP_TRY(hipEventCreate(&c->completeStart));
    HIP_TRY(hipEventCreate(&c->completeEnd));
    HIP_TRY(hipEventCreate(&c->individualStart));
    HIP_TRY(hipEventCreate(&c->individualEnd));
    for (int i = 0; i < kMaxClasses; ++i) {
        hipStream_t s;
        hipEvent_t e;
        // (a HIGH-priority stream for the NUM_G chain -- seven short kernels that queue for LDS behind the long rows of
        //  the other launches -- was tried in round 4: the webbase stand-in lost 5 %, and the mere existence of such a
        //  stream cost the scircuit stand-in, which never uses it, 55 %: 0.088 -> 0.138 ms.  Not done.)
        HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->aux.push_back(s);
        c->aux_done.push_back(e);
    }
    HIP_TRY(hipEventCreateWithFlags(&c->fork, hipEventDisableTiming));
    {
        // the verifier yields to the sequence it runs beside: lowest stream priority (the sequence's light launches
        // lost 5-8 % to it at equal priority)
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        HIP_TRY(hipStreamCreateWithPriority(&c->vstream, hipStreamNonBlocking, least));
    }
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_verify), 128, hipHostMallocMapped));
    std::memset(c->h_verify, 0, 128);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_vticket), sizeof(u32)));
    HIP_TRY(hipMemset(c->d_vticket, 0, sizeof(u32)));
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->h_verify_dev), c->h_verify, 0));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_stats), sizeof(DeviceStats)));
    // (b_bad_epoch is the one word no kernel zeroes: recycled memory of a destroyed config must not hold an epoch
    //  this config is going to use)
    HIP_TRY(hipMemset(c->d_stats, 0, sizeof(DeviceStats)));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_stats), sizeof(DeviceStats), hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->h_stats_dev), c->h_stats, 0));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_ticket), sizeof(u32)));
    HIP_TRY(hipMemset(c->d_ticket, 0, sizeof(u32)));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_ticket), 64, hipHostMallocMapped));
    *c->h_ticket = 0;
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->h_ticket_dev), c->h_ticket, 0));
    c->cp.sym_bitmap_ratio = 32;
    c->cp.num_dense_ratio = 16;
    c->cp.num_global_passes = 4;  // heavy rows: dense windows up to 64 Ki columns, else global spill
    c->cp.nf_min_ops = 512;   // numeric-first for narrow rows with at least this many products (0 = off; 256 costs the
                              // scircuit stand-in 5 %, 1024 leaves the boundary rows of the cant one a launch of their own)
    c->cp.gh_per_window = 8192;  // global key set for rows with fewer products per 1 Mi-column bitmap window (0 = off)
    c->cp.esc16 = 1;       // rows of <= 64 products from <= 16 entries: 16 lanes per row, in registers (esc.hpp)
    c->cp.esc4 = 0;        // rows of <= 16 products from <= 4 entries: 4 lanes per row, carved out of the 8-lane class.  OFF:
                           //   measured round 4 -- webbase stand-in -1.5 % time, but mac_econ +2 % (its fused launch 52.4 ->
                           //   56.0 us although 55 % of its 8-lane rows qualify) and scircuit +1 %: splitting the list of small
                           //   rows in two costs the B-row locality of neighbouring rows more than the half-size network saves
    c->cp.esc32 = 1;       // rows of <= 128 products from <= 32 entries: 32 lanes per row, in registers (esc_wide.hpp)
    c->cp.esc64 = 1;       // rows of <= 256 products from <= 64 entries: a wave per row, in registers
    c->cp.num_g8 = 1;      // rows of <= 32 products from <= 8 entries: 8 lanes per row, in registers
    c->cp.sym_g8 = 1;      // rows of <= 25 products: 8 lanes per row
    c->cp.sym_w128 = 1;    // rows of 52..102 products: 16 lanes per row
    c->cp.num_w256 = 1;    // rows of 86..170 entries: 32 lanes per row
    c->cp.want_bytes = 0;
    *out = c;
    return SPECK_OK;
}

int speck_config_destroy(speck_config* c)
{
    if (!c) return SPECK_ERR_INVALID;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    drop_graph(c);
    for (auto s : c->streams) (void)hipStreamDestroy(s);
    (void)hipEventDestroy(c->completeStart);
    (void)hipEventDestroy(c->completeEnd);
    (void)hipEventDestroy(c->individualStart);
    (void)hipEventDestroy(c->individualEnd);
    for (auto e : c->kev) (void)hipEventDestroy(e);
    for (auto s : c->aux) (void)hipStreamDestroy(s);
    for (auto e : c->aux_done) (void)hipEventDestroy(e);
    if (c->fork) (void)hipEventDestroy(c->fork);
    if (c->vstream) (void)hipStreamDestroy(c->vstream);
    if (c->h_verify) (void)hipHostFree(c->h_verify);
    if (c->d_vticket) (void)hipFree(c->d_vticket);
    if (c->arena) (void)hipFree(c->arena);
    if (c->snap) (void)hipFree(c->snap);
    if (c->gpool) (void)hipFree(c->gpool);
    if (c->nfpool) (void)hipFree(c->nfpool);
    if (c->pred.buf) (void)hipFree(c->pred.buf);
    if (c->gpred.buf) (void)hipFree(c->gpred.buf);
    if (c->d_stats) (void)hipFree(c->d_stats);
    if (c->h_stats) (void)hipHostFree(c->h_stats);
    if (c->h_ticket) (void)hipHostFree(c->h_ticket);
    if (c->d_ticket) (void)hipFree(c->d_ticket);
    delete c;
    return SPECK_OK;
}

int speck_config_info(const speck_config* c, int* sm, int* max_static_lds, int* max_dynamic_lds)
{
    if (!c) return SPECK_ERR_INVALID;
    if (sm) *sm = c->sm;
    if (max_static_lds) *max_static_lds = c->max_static_lds;
    if (max_dynamic_lds) *max_dynamic_lds = c->max_dynamic_lds;
    return SPECK_OK;
}

int speck_config_handles(const speck_config* c, void* streams6[6], void* events4[4])
{
    if (!c) return SPECK_ERR_INVALID;
    if (streams6)
        for (int i = 0; i < 6; ++i) streams6[i] = c->streams[i];
    if (events4) {
        events4[0] = c->completeStart;
        events4[1] = c->completeEnd;
        events4[2] = c->individualStart;
        events4[3] = c->individualEnd;
    }
    return SPECK_OK;
}

int speck_config_set_stream(speck_config* c, void* hip_stream)
{
    if (!c) return SPECK_ERR_INVALID;
    c->user_stream = static_cast<hipStream_t>(hip_stream);
    c->use_user_stream = hip_stream != nullptr;
    drop_graph(c);  // (a sequence is tied to the stream it was made for, and to HOW it is replayed there)
    return SPECK_OK;
}

int speck_config_set_option(speck_config* c, const char* name, int64_t value)
{
    if (!c || !name) return SPECK_ERR_INVALID;
    const std::string n(name);
    if (n == "sym_bitmap_ratio") c->cp.sym_bitmap_ratio = (u32)value;
    else if (n == "num_dense_ratio") c->cp.num_dense_ratio = (u32)value;
    else if (n == "num_global_passes") c->cp.num_global_passes = (u32)value;
    else if (n == "gh_per_window") {
        c->cp.gh_per_window = (u32)value;
        drop_graph(c);
        c->last_key_valid = false;
    } else if (n == "nf_min_ops") {
        c->cp.nf_min_ops = (u32)value;
        drop_graph(c);
        c->last_key_valid = false;
    }
    else if (n == "nf_pool_max_mb") c->nf_pool_max_bytes = size_t(value) << 20;
    else if (n == "profile_replay") c->profile_replay = value != 0;
    else if (n == "replay_uncaptured") {
        c->replay_uncaptured = value != 0;
        drop_graph(c);
    }
    else if (n == "analysis_wide_rows") {
        set_analysis_wide_rows((u32)value);
        drop_graph(c);
        c->last_key_valid = false;
    }
    else if (n == "eager_speculate") c->eager_speculate = value != 0;
    else if (n == "capture_forked") {
        c->capture_forked = value != 0;
        drop_graph(c);
    }
    else if (n == "verify_inputs") {
        c->verify_inputs = value != 0;
        c->snap_for_arena = false;
        drop_graph(c);
    }
    else if (n == "num_verify") {
        c->num_verify = (int)value;
        drop_graph(c);
    }
    else if (n == "skip_scan") {
        c->skip_scan = value != 0;
        drop_graph(c);
    }
    else if (n == "overlap_analysis") {
        c->overlap_analysis = value != 0;
        drop_graph(c);
    }
    else if (n == "pred_sym") {
        c->pred_sym = value != 0;
        drop_graph(c);
        c->last_key_valid = false;
    }
    else if (n == "pred_scan") {
        c->pred_scan = value != 0;
        drop_graph(c);
        c->last_key_valid = false;
    }
    else if (n == "esc_fused") {
        c->esc_fused = value != 0;
        drop_graph(c);
        c->last_key_valid = false;
    }
    else if (n == "nf_direct") {
        c->nf_direct = value != 0;
        drop_graph(c);
    }
    else if (n == "sym_w128") {
        c->cp.sym_w128 = value != 0;
        drop_graph(c);
        c->last_key_valid = false;
    }
    else if (n == "sym_g8") {
        c->cp.sym_g8 = value != 0;
        drop_graph(c);
        c->last_key_valid = false;
    }
    else if (n == "esc16") {
        c->cp.esc16 = value != 0;
        drop_graph(c);
        c->last_key_valid = false;
    }
    else if (n == "esc32" || n == "esc64" || n == "esc4") {
        (n == "esc32" ? c->cp.esc32 : (n == "esc64" ? c->cp.esc64 : c->cp.esc4)) = value != 0;
        drop_graph(c);
        c->last_key_valid = false;
    }
    else if (n == "num_g8") {
        c->cp.num_g8 = value != 0;
        drop_graph(c);
        c->last_key_valid = false;
    }
    else if (n == "num_w256") {
        c->cp.num_w256 = value != 0;
        drop_graph(c);
        c->last_key_valid = false;
    }
    else if (n == "collect_bytes") c->cp.want_bytes = value != 0;        // per-class byte model
    else if (n == "concurrent_classes") c->concurrent_classes = value != 0;
    else if (n == "validate_inputs") c->validate_inputs = value != 0;
    else if (n == "spin_wait") {
        c->spin_wait = value != 0;
        drop_graph(c);
    }
    else if (n == "fork_min_us") {
        c->fork_min_us = (float)value;
        drop_graph(c);
    }
    else if (n == "split_min_us") {
        c->split_min_us = (float)value;
        drop_graph(c);
    }
    else if (n == "max_side_streams") {
        c->max_side_streams = (u32)value;
        drop_graph(c);
    }
    else if (n == "use_graph") {
        // (an eager call leaves a prediction behind only for a config that may replay: the call that follows the switch
        //  is a complete one)
        if (c->use_graph != (value != 0)) {
            drop_graph(c);
            c->last_key_valid = false;
        }
        c->use_graph = value != 0;
    }
    else if (n == "grid_rounds_block") {
        set_grid_rounds((u32)value, 0);
        drop_graph(c);
    }
    else if (n == "grid_rounds_sub") {
        set_grid_rounds(0, (u32)value);
        drop_graph(c);
    }
    else if (n == "spill_big_grid") {
        set_spill_big_grid((u32)value);
        drop_graph(c);
    }
    else if (n == "tiny_threads") {
        set_tiny_threads((int)value);
        drop_graph(c);
    }
    else if (n == "split_light") {
        c->split_light = value != 0;
        drop_graph(c);
    }
    else if (n == "xcd_aware") {
        c->xcd_aware = (u32)value;
        drop_graph(c);
    }
    else if (n == "merge_light") {
        c->merge_light = value != 0;
        drop_graph(c);
    }
    else return SPECK_ERR_INVALID;
    return SPECK_OK;
}

int speck_config_profile_kernels(speck_config* c, int enable)
{
    if (!c) return SPECK_ERR_INVALID;
    c->profile_kernels = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
    return SPECK_OK;
}

int speck_last_stats(const speck_config* c, speck_stats* out)
{
    if (!c || !out) return SPECK_ERR_INVALID;
    *out = c->last;
    out->numeric_reruns = c->graph_misses;
    out->graph_replays = c->graph_replays;
    out->graph_captures = c->graph_captures;
    out->pool_fallbacks = c->pool_fallbacks;
    out->scratch_pool_bytes = c->nfpool_bytes;
    return SPECK_OK;
}

int speck_multiply_f64(speck_config* c, const speck_dcsr* A, const speck_dcsr* B, speck_dcsr* C,
                       speck_timings* t)
{
    return multiply_impl<double>(c, A, B, C, t);
}

int speck_multiply_f32(speck_config* c, const speck_dcsr* A, const speck_dcsr* B, speck_dcsr* C,
                       speck_timings* t)
{
    return multiply_impl<float>(c, A, B, C, t);
}

int speck_analysis(speck_config* c, const speck_dcsr* A, const speck_dcsr* B, uint32_t* d_row_ops,
                   uint32_t* d_row_max_ops, uint32_t* d_row_col_min, uint32_t* d_row_col_max,
                   uint64_t* h_sum_products, uint32_t* h_max_row_ops)
{
    if (!c) return SPECK_ERR_INVALID;
    int rc = check_inputs(A, B);
    if (rc != SPECK_OK) return rc;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = main_stream(c);
    c->exec_dirty = true;  // (stage entry point: launches on the pipeline stream)
    c->arena_key_valid = false;  // (... and writes the arena)
    const u32 m = (u32)A->rows;
    if (m == 0 || A->nnz == 0 || B->nnz == 0) {
        if (h_sum_products) *h_sum_products = 0;
        if (h_max_row_ops) *h_max_row_ops = 0;
        if (m) {
            if (d_row_ops) HIP_TRY(hipMemsetAsync(d_row_ops, 0, size_t(m) * 4, s));
            if (d_row_max_ops) HIP_TRY(hipMemsetAsync(d_row_max_ops, 0, size_t(m) * 4, s));
            if (d_row_col_min) HIP_TRY(hipMemsetAsync(d_row_col_min, 0xFF, size_t(m) * 4, s));
            if (d_row_col_max) HIP_TRY(hipMemsetAsync(d_row_col_max, 0, size_t(m) * 4, s));
            HIP_TRY(hipStreamSynchronize(s));
        }
        return SPECK_OK;
    }
    rc = ensure_arena(c, scratch_bytes(m, A->nnz));
    if (rc != SPECK_OK) return rc;
    Scratch sc = carve(c, m, A->nnz);  // partials come from the arena, row arrays from the caller
    HIP_TRY(hipMemsetAsync(c->d_stats, 0, sizeof(DeviceStats), s));
    launch_analysis(s, A->row_offsets, A->col_ids, B->row_offsets, B->col_ids, m, A->nnz, d_row_ops,
                    d_row_max_ops, d_row_col_min, d_row_col_max, nullptr, nullptr, sc.partials, sc.recs,
                    c->d_stats, [&] {
                        ClassifyParams cp = c->cp;
                        cp.sym_allowed = cp.num_allowed = 0xFFFFFFFFu;
                        return cp;
                    }(), nullptr, nullptr, nullptr, ~0ull, (u32)B->rows);
    HIP_TRY(hipGetLastError());
    rc = read_stats(c, s);
    if (rc != SPECK_OK) return rc;
    if (c->h_stats->a_invalid) return SPECK_ERR_INVALID;
    if (h_sum_products) *h_sum_products = c->h_stats->sum_products;
    if (h_max_row_ops) *h_max_row_ops = c->h_stats->max_row_ops;
    return SPECK_OK;
}

int speck_symbolic(speck_config* c, const speck_dcsr* A, const speck_dcsr* B, uint32_t* d_row_offsets,
                   uint64_t* h_nnz_c)
{
    if (!c || !d_row_offsets) return SPECK_ERR_INVALID;
    int rc = check_inputs(A, B);
    if (rc != SPECK_OK) return rc;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = main_stream(c);
    c->exec_dirty = true;  // (stage entry point: launches on the pipeline stream)
    c->arena_key_valid = false;  // (... and writes the arena)
    const u32 m = (u32)A->rows;
    if (A->nnz == 0 || B->nnz == 0) {
        HIP_TRY(hipMemsetAsync(d_row_offsets, 0, (size_t(m) + 1) * 4, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (h_nnz_c) *h_nnz_c = 0;
        return SPECK_OK;
    }
    rc = ensure_arena(c, scratch_bytes(m, A->nnz));
    if (rc != SPECK_OK) return rc;
    Scratch sc = carve(c, m, A->nnz);
    const u32 nf_was = c->cp.nf_min_ops, gh_was = c->cp.gh_per_window;
    c->cp.nf_min_ops = 0;  // structure only: no values here, every row through a symbolic kernel
    c->cp.gh_per_window = 0;  // ... one that needs no scratch pool
    rc = enqueue_front(c, s, A, B, sc, d_row_offsets, 8, ~0ull, kAllSym, kAllNum, false, nullptr);
    c->cp.nf_min_ops = nf_was;
    c->cp.gh_per_window = gh_was;
    if (rc != SPECK_OK) return rc;
    rc = read_stats(c, s);
    if (rc != SPECK_OK) return rc;
    c->last_key_valid = false;
    if (c->h_stats->nnz_overflow) return SPECK_ERR_NNZ_OVERFLOW;
    if (h_nnz_c) *h_nnz_c = c->h_stats->nnz_c;
    return SPECK_OK;
}

int speck_partition_rows(speck_config* c, const speck_dcsr* A, const speck_dcsr* B, int parts,
                         uint64_t* h_bounds)
{
    if (!c || !h_bounds || parts <= 0) return SPECK_ERR_INVALID;
    int rc = check_inputs(A, B);
    if (rc != SPECK_OK) return rc;
    HIP_TRY(hipSetDevice(c->device));
    const u32 m = (u32)A->rows;
    rc = ensure_arena(c, scratch_bytes(m, A->nnz));
    if (rc != SPECK_OK) return rc;
    Scratch sc = carve(c, m, A->nnz);
    u64 P = 0;
    rc = speck_analysis(c, A, B, sc.row_ops, nullptr, nullptr, nullptr, &P, nullptr);
    if (rc != SPECK_OK) return rc;
    std::vector<u32> ops(m);
    if (m) HIP_TRY(hipMemcpy(ops.data(), sc.row_ops, size_t(m) * 4, hipMemcpyDeviceToHost));
    // cost per row: products plus a per-row constant so that empty rows still spread
    h_bounds[0] = 0;
    u64 total = P + m;
    u64 run = 0;
    int next = 1;
    for (u32 i = 0; i < m && next < parts; ++i) {
        run += u64(ops[i]) + 1;
        while (next < parts && run * parts >= total * next) h_bounds[next++] = i + 1;
    }
    while (next <= parts) h_bounds[next++] = m;
    return SPECK_OK;
}

int speck_dcsr_alloc(speck_dcsr* m, uint64_t rows, uint64_t cols, uint64_t nnz, int alloc_offsets,
                     size_t value_size)
{
    if (!m) return SPECK_ERR_INVALID;
    speck_dcsr_free(m);
    m->rows = rows;
    m->cols = cols;
    m->nnz = nnz;
    HIP_TRY(hipMalloc(&m->data, std::max<size_t>(nnz, 1) * value_size));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->col_ids), std::max<size_t>(nnz, 1) * 4));
    if (alloc_offsets) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->row_offsets), (rows + 1) * 4));
    return SPECK_OK;
}

int speck_dcsr_free(speck_dcsr* m)
{
    if (!m) return SPECK_ERR_INVALID;
    if (m->col_ids) (void)hipFree(m->col_ids);
    if (m->data) (void)hipFree(m->data);
    if (m->row_offsets) (void)hipFree(m->row_offsets);
    m->col_ids = nullptr;
    m->data = nullptr;
    m->row_offsets = nullptr;
    m->nnz = 0;
    m->rows = 0;
    return SPECK_OK;
}

int speck_dcsr_upload(speck_dcsr* dst, uint64_t rows, uint64_t cols, uint64_t nnz,
                      const uint32_t* h_row_offsets, const uint32_t* h_col_ids, const void* h_data,
                      size_t value_size)
{
    return speck_dcsr_upload_padded(dst, rows, cols, nnz, h_row_offsets, h_col_ids, h_data, value_size, 0u);
}

int speck_dcsr_upload_padded(speck_dcsr* dst, uint64_t rows, uint64_t cols, uint64_t nnz, const uint32_t* h_row_offsets,
                             const uint32_t* h_col_ids, const void* h_data, size_t value_size, uint32_t padding)
{
    // reference: dst.alloc(rows + padding, cols, nnz + 8 * padding), then rows / nnz of the source (dCSR.cpp:53-54)
    int rc = speck_dcsr_alloc(dst, rows + padding, cols, nnz + 8ull * padding, 1, value_size);
    if (rc != SPECK_OK) return rc;
    dst->rows = rows;
    dst->nnz = nnz;
    if (nnz) {
        HIP_TRY(hipMemcpy(dst->data, h_data, nnz * value_size, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dst->col_ids, h_col_ids, nnz * 4, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(dst->row_offsets, h_row_offsets, (rows + 1) * 4, hipMemcpyHostToDevice));
    if (padding) {  // dCSR.cpp:59-64
        HIP_TRY(hipMemset(static_cast<char*>(dst->data) + nnz * value_size, 0, 8ull * padding * value_size));
        HIP_TRY(hipMemset(dst->col_ids + nnz, 0, 8ull * padding * 4));
        HIP_TRY(hipMemset(dst->row_offsets + rows + 1, 0, size_t(padding) * 4));
    }
    return SPECK_OK;
}

int speck_dcsr_copy(speck_dcsr* dst, const speck_dcsr* src, size_t value_size, uint32_t padding)
{
    if (!dst || !src || dst == src) return SPECK_ERR_INVALID;
    if (src->rows && !src->row_offsets) return SPECK_ERR_INVALID;
    if (dst->data && dst->data == src->data) return SPECK_ERR_INVALID;  // alloc frees dst first (dCSR.cpp:28)
    const uint64_t rows = src->rows, nnz = src->nnz;
    int rc = speck_dcsr_alloc(dst, rows + padding, src->cols, nnz + 8ull * padding, 1, value_size);
    if (rc != SPECK_OK) return rc;
    dst->rows = rows;
    dst->nnz = nnz;
    const u64 work = std::max<u64>(rows + 1, nnz * (value_size / 4));
    hipLaunchKernelGGL(dcsr_copy_kernel, dim3((unsigned)std::min<u64>((work + 255) / 256, 8192)), dim3(256), 0, nullptr,
                       src->row_offsets, src->col_ids, static_cast<const u32*>(src->data), dst->row_offsets, dst->col_ids,
                       static_cast<u32*>(dst->data), rows, nnz, (u32)(value_size / 4));
    HIP_TRY(hipGetLastError());
    if (padding) {
        HIP_TRY(hipMemsetAsync(static_cast<char*>(dst->data) + nnz * value_size, 0, 8ull * padding * value_size, nullptr));
        HIP_TRY(hipMemsetAsync(dst->col_ids + nnz, 0, 8ull * padding * 4, nullptr));
        HIP_TRY(hipMemsetAsync(dst->row_offsets + rows + 1, 0, size_t(padding) * 4, nullptr));
    }
    HIP_TRY(hipStreamSynchronize(nullptr));
    return SPECK_OK;
}

int speck_dcsr_download(const speck_dcsr* src, uint32_t* h_row_offsets, uint32_t* h_col_ids,
                        void* h_data, size_t value_size)
{
    if (!src) return SPECK_ERR_INVALID;
    if (src->nnz) {
        if (h_data) HIP_TRY(hipMemcpy(h_data, src->data, src->nnz * value_size, hipMemcpyDeviceToHost));
        if (h_col_ids) HIP_TRY(hipMemcpy(h_col_ids, src->col_ids, src->nnz * 4, hipMemcpyDeviceToHost));
    }
    if (h_row_offsets && src->row_offsets)
        HIP_TRY(hipMemcpy(h_row_offsets, src->row_offsets, (src->rows + 1) * 4, hipMemcpyDeviceToHost));
    return SPECK_OK;
}

int speck_dcsr_update(speck_dcsr* dst, const uint32_t* h_row_offsets, const uint32_t* h_col_ids,
                      const void* h_data, size_t value_size)
{
    if (!dst) return SPECK_ERR_INVALID;
    if (dst->nnz) {
        if (h_data) HIP_TRY(hipMemcpy(dst->data, h_data, dst->nnz * value_size, hipMemcpyHostToDevice));
        if (h_col_ids) HIP_TRY(hipMemcpy(dst->col_ids, h_col_ids, dst->nnz * 4, hipMemcpyHostToDevice));
    }
    if (h_row_offsets && dst->row_offsets)
        HIP_TRY(hipMemcpy(dst->row_offsets, h_row_offsets, (dst->rows + 1) * 4, hipMemcpyHostToDevice));
    return SPECK_OK;
}

const char* speck_status_string(int status)
{
    switch (status) {
        case SPECK_OK: return "ok";
        case SPECK_ERR_INVALID: return "invalid argument";
        case SPECK_ERR_DIM_LIMIT: return "matrix dimension above the 2^27 limit";
        case SPECK_ERR_HIP: return "HIP runtime error";
        case SPECK_ERR_OOM: return "out of device memory";
        case SPECK_ERR_NNZ_OVERFLOW: return "nnz(C) exceeds 2^32-1 (u32 row_offsets)";
        case SPECK_ERR_NO_DEVICE: return "no such HIP device";
        case SPECK_ERR_IO: return "I/O error";
        case SPECK_ERR_UNSORTED: return "a row of B is not strictly ascending (or holds a column >= cols)";
        case SPECK_ERR_COMM: return "multi-GPU exchange failed (RCCL / shared-memory transport)";
    }
    return "unknown";
}

const char* speck_version(void) { return "speck_amd 0.4 (gfx950)"; }

}  // extern "C"
