// pipeline.hip -- host pipeline + C-ABI of the MI355X SpGEMM backend.
// Role of the reference's MultiplyspECKImplementation (source/GPU/Multiply.cu:51-1122),
// spECKConfig (include/spECKConfig.h) and dCSR helpers (source/dCSR.cpp), re-designed:
//   * one grow-only scratch arena per config (the reference cudaMallocs/frees every
//     scratch buffer inside each call, Multiply.cu:202-225,1056-1070)
//   * two blocking read-backs per call (after binning, after the scan) instead of 5-8
//   * every kernel takes its row list + counts from a device-side stats block
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
    DeviceStats* h_stats = nullptr;  // pinned
    ClassifyParams cp{};
    bool profile_kernels = false;
    std::vector<hipEvent_t> kev;  // kernel event pool (timing)
    std::vector<hipStream_t> aux;  // one stream per kernel class: classes run concurrently
    std::vector<hipEvent_t> aux_done;
    hipEvent_t fork = nullptr;
    bool concurrent_classes = true;
    speck_stats last{};
};

namespace {

hipStream_t main_stream(speck_config* c) { return c->use_user_stream ? c->user_stream : c->streams[0]; }

int ensure_arena(speck_config* c, size_t bytes)
{
    if (bytes <= c->arena_bytes) return SPECK_OK;
    if (c->arena) HIP_TRY(hipFree(c->arena));
    c->arena = nullptr;
    c->arena_bytes = 0;
    size_t want = bytes + bytes / 4 + (1 << 20);
    HIP_TRY(hipMalloc(&c->arena, want));
    c->arena_bytes = want;
    return SPECK_OK;
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
    u32 *row_ops, *row_max_ops, *row_col_min, *row_col_max, *bin_rows;
    u8* cls;
    u64* tile_sums;
    BlockPartial* partials;
    u32* blk_base;
};

u32 partial_blocks(u32 m) { return std::max(analysis_blocks(m), scan_tiles(m)); }

size_t scratch_bytes(u32 m)
{
    size_t b = 0;
    b += 5 * Carver::need(m, 4);
    b += Carver::need(m, 1);
    b += Carver::need(scan_scratch_bytes(m), 1);
    b += Carver::need(partial_blocks(m), sizeof(BlockPartial));
    b += Carver::need(size_t(partial_blocks(m)) * kMaxClasses, 4);
    return b + 4096;
}

Scratch carve(speck_config* c, u32 m)
{
    Carver cv(c->arena);
    Scratch s;
    s.row_ops = cv.take<u32>(m);
    s.row_max_ops = cv.take<u32>(m);
    s.row_col_min = cv.take<u32>(m);
    s.row_col_max = cv.take<u32>(m);
    s.bin_rows = cv.take<u32>(m);
    s.cls = cv.take<u8>(m);
    s.tile_sums = reinterpret_cast<u64*>(cv.take<u8>(scan_scratch_bytes(m)));
    s.partials = cv.take<BlockPartial>(partial_blocks(m));
    s.blk_base = cv.take<u32>(size_t(partial_blocks(m)) * kMaxClasses);
    return s;
}

struct StageTimer {
    speck_config* c;
    bool on;
    hipStream_t s;
    StageTimer(speck_config* cfg, bool enable, hipStream_t st) : c(cfg), on(enable), s(st)
    {
        if (on) start();
    }
    void start()
    {
        (void)hipEventRecord(c->individualStart, s);
    }
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

// Runs analysis (+ optional symbolic classification/binning) and reads the stats back.
int run_analysis(speck_config* c, hipStream_t s, const speck_dcsr* A, const speck_dcsr* B,
                 const Scratch& sc, bool classify, u32* counts)
{
    const u32 m = (u32)A->rows;
    HIP_TRY(hipMemsetAsync(c->d_stats, 0, sizeof(DeviceStats), s));
    launch_analysis(s, A->row_offsets, A->col_ids, B->row_offsets, B->col_ids, m, A->nnz, sc.row_ops,
                    sc.row_max_ops, sc.row_col_min, sc.row_col_max, classify ? sc.cls : nullptr,
                    counts, sc.partials, sc.blk_base, sc.bin_rows, c->d_stats, c->cp);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(c->h_stats, c->d_stats, sizeof(DeviceStats), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SPECK_OK;
}

// Launch one kernel per non-empty class.  The classes are independent (disjoint rows), so each
// runs on its own stream between a fork and a join event on the pipeline stream: the
// latency-bound heavy-row kernels (few workgroups) overlap the throughput-bound small-row ones
// (the reference does the same with its 6 streams, source/GPU/Multiply.cu:494-553, but relies on
// legacy default-stream ordering; here the dependencies are explicit events).
struct ClassTiming {
    int cls;
    size_t ev;
};

template <typename LaunchFn>
int run_classes(speck_config* c, hipStream_t s, const int* order, int n_order, const u32* counts,
                size_t* ev_idx, std::vector<ClassTiming>* timing, LaunchFn&& launch)
{
    bool forked = false;
    size_t used = 0;
    for (int i = 0; i < n_order; ++i) {
        const int cls = order[i];
        const u32 cnt = counts[cls];
        if (!cnt) continue;
        hipStream_t ks = s;
        if (c->concurrent_classes && used < c->aux.size()) {
            if (!forked) {
                HIP_TRY(hipEventRecord(c->fork, s));
                forked = true;
            }
            ks = c->aux[used];
            HIP_TRY(hipStreamWaitEvent(ks, c->fork, 0));
        }
        if (c->profile_kernels) (void)hipEventRecord(kernel_event(c, *ev_idx), ks);
        launch(ks, cls, cnt);
        if (c->profile_kernels) {
            (void)hipEventRecord(kernel_event(c, *ev_idx + 1), ks);
            timing->push_back({cls, *ev_idx});
            *ev_idx += 2;
        }
        if (ks != s) {
            HIP_TRY(hipEventRecord(c->aux_done[used], ks));
            HIP_TRY(hipStreamWaitEvent(s, c->aux_done[used], 0));
            ++used;
        }
    }
    HIP_TRY(hipGetLastError());
    return SPECK_OK;
}

int run_symbolic_kernels(speck_config* c, hipStream_t s, const speck_dcsr* A, const speck_dcsr* B,
                         const Scratch& sc, u32* counts, size_t* ev_idx,
                         std::vector<ClassTiming>* timing)
{
    RowWork w{sc.bin_rows, sc.row_ops, sc.row_col_min, sc.row_col_max, c->d_stats};
    // heaviest classes first: they have the longest tails
    static const int order[SYM_CLASSES] = {SYM_BM2, SYM_B32K, SYM_B16K, SYM_B4K,
                                           SYM_BM1, SYM_W1K,  SYM_W256, SYM_G16};
    return run_classes(c, s, order, SYM_CLASSES, c->h_stats->sym.count, ev_idx, timing,
                       [&](hipStream_t ks, int cls, u32 cnt) {
                           launch_symbolic(ks, cls, cnt, A->row_offsets, A->col_ids, B->row_offsets,
                                           B->col_ids, w, counts, c->sm);
                       });
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

    if (t->measureCompleteTime) {
        HIP_TRY(hipEventRecord(c->completeStart, s));
    }
    StageTimer st(c, t->measureAll != 0, s);

    // ---- INIT: scratch from the arena; C.row_offsets reuse rule (Multiply.cu:156-165)
    rc = ensure_arena(c, scratch_bytes(m));
    if (rc != SPECK_OK) return rc;
    Scratch sc = carve(c, m);
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

    // ---- ANALYSIS + symbolic binning (Multiply.cu:239-252, 279-345)
    size_t ev = 0;
    if (c->profile_kernels) (void)hipEventRecord(kernel_event(c, ev++), s);
    rc = run_analysis(c, s, A, B, sc, true, c_ro);
    if (rc != SPECK_OK) return fail(rc);
    // (event pair 0/1 brackets analysis+binning; recorded after the sync is harmless)
    if (c->profile_kernels) (void)hipEventRecord(kernel_event(c, ev++), s);
    const u64 P = c->h_stats->sum_products;
    c->last.sum_products = P;
    c->last.max_row_ops = c->h_stats->max_row_ops;
    t->countProducts = st.lap();

    if (P == 0) {
        // reference: Multiply.cu:256-261 -> matOut.alloc(rows, cols, 0, false)
        if (own_ro) (void)hipFree(c_ro);
        speck_dcsr_free(C);
        C->rows = A->rows;
        C->cols = B->cols;
        C->nnz = 0;
        if (t->measureCompleteTime) {
            HIP_TRY(hipEventRecord(c->completeEnd, s));
            HIP_TRY(hipEventSynchronize(c->completeEnd));
            HIP_TRY(hipEventElapsedTime(&t->complete, c->completeStart, c->completeEnd));
        }
        return SPECK_OK;
    }
    t->loadBalanceCounting = st.lap();
    t->globalMapsCounting = 0.f;

    // ---- SYMBOLIC (Multiply.cu:488-554)
    std::vector<ClassTiming> sym_timing, num_timing;
    rc = run_symbolic_kernels(c, s, A, B, sc, c_ro, &ev, &sym_timing);
    if (rc != SPECK_OK) return fail(rc);
    for (int i = 0; i < SPECK_NUM_SYM_BINS; ++i) {  // h_stats is overwritten by the next read-back
        c->last.sym_bin_rows[i] = c->h_stats->sym.count[i];
        c->last.sym_bin_bytes[i] = c->h_stats->sym.bytes[i];
    }

    // ---- SCAN + numeric classification/binning (Multiply.cu:570-575, 615-682)
    const size_t ev_scan = ev;
    if (c->profile_kernels) (void)hipEventRecord(kernel_event(c, ev++), s);
    launch_scan(s, c_ro, m, sc.tile_sums, A->row_offsets, sc.row_ops, sc.row_col_min, sc.row_col_max,
                sc.cls, sc.partials, sc.blk_base, sc.bin_rows, c->d_stats, c->cp, (u32)sizeof(T));
    if (c->profile_kernels) (void)hipEventRecord(kernel_event(c, ev++), s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(c->h_stats, c->d_stats, sizeof(DeviceStats), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    t->spGEMMCounting = st.lap();
    if (c->h_stats->nnz_overflow) return fail(SPECK_ERR_NNZ_OVERFLOW);
    const u64 nnz_c = c->h_stats->nnz_c;
    c->last.nnz_c = nnz_c;
    c->last.max_row_nnz_c = c->h_stats->max_row_nnz_c;

    // ---- ALLOC C: only when nnz changed (Multiply.cu:589-602)
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
    t->allocC = st.lap();
    t->loadBalanceNumeric = 0.f;
    t->globalMapsNumeric = 0.f;

    // ---- NUMERIC (Multiply.cu:835-1014) + in-kernel sort (Multiply.cu:1028-1043)
    CsrView<T> Av{A->row_offsets, A->col_ids, static_cast<const T*>(A->data), m, (u32)A->cols};
    CsrView<T> Bv{B->row_offsets, B->col_ids, static_cast<const T*>(B->data), (u32)B->rows,
                  (u32)B->cols};
    RowWork w{sc.bin_rows, sc.row_ops, sc.row_col_min, sc.row_col_max, c->d_stats};
    static const int order[NUM_CLASSES] = {NUM_G,   NUM_D2,   NUM_B8K,  NUM_B2K,   NUM_D1,
                                           NUM_W512, NUM_W128, NUM_G16, NUM_DIRECT};
    rc = run_classes(c, s, order, NUM_CLASSES, c->h_stats->num.count, &ev, &num_timing,
                     [&](hipStream_t ks, int cls, u32 cnt) {
                         launch_numeric<T>(ks, cls, cnt, Av, Bv, w, c_ro, c_col, static_cast<T*>(c_val),
                                           nnz_c, c->d_stats, c->sm);
                     });
    if (rc != SPECK_OK) return rc;
    if (t->measureAll) {
        HIP_TRY(hipStreamSynchronize(s));
    }
    t->spGEMMNumeric = st.lap();
    t->sorting = 0.f;  // sorting is fused into the numeric kernels
    t->cleanup = 0.f;  // nothing to free: the arena persists

    if (t->measureCompleteTime) {
        // reference: cudaDeviceSynchronize + complete event, Multiply.cu:1082-1085
        HIP_TRY(hipEventRecord(c->completeEnd, s));
        HIP_TRY(hipEventSynchronize(c->completeEnd));
        HIP_TRY(hipEventElapsedTime(&t->complete, c->completeStart, c->completeEnd));
    }

    // ---- stats for the harness
    for (int i = 0; i < SPECK_NUM_NUM_BINS; ++i) {
        c->last.num_bin_rows[i] = c->h_stats->num.count[i];
        c->last.num_bin_bytes[i] = c->h_stats->num.bytes[i];
    }
    if (c->profile_kernels) {
        HIP_TRY(hipStreamSynchronize(s));
        auto ms = [&](size_t a) {
            float v = 0.f;
            (void)hipEventElapsedTime(&v, c->kev[a], c->kev[a + 1]);
            return v;
        };
        c->last.analysis_ms = ms(0);
        c->last.scan_ms = ms(ev_scan);
        for (const auto& ct : sym_timing) c->last.sym_bin_ms[ct.cls] = ms(ct.ev);
        for (const auto& ct : num_timing) c->last.num_bin_ms[ct.cls] = ms(ct.ev);
        c->last.kernel_events_valid = 1;
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
        HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->aux.push_back(s);
        c->aux_done.push_back(e);
    }
    HIP_TRY(hipEventCreateWithFlags(&c->fork, hipEventDisableTiming));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_stats), sizeof(DeviceStats)));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_stats), sizeof(DeviceStats), hipHostMallocDefault));
    c->cp.sym_bitmap_ratio = 32;
    c->cp.num_dense_ratio = 16;
    c->cp.num_global_passes = 0xFFFFFFFFu;
    c->cp.want_bytes = 0;
    *out = c;
    return SPECK_OK;
}

int speck_config_destroy(speck_config* c)
{
    if (!c) return SPECK_ERR_INVALID;
    (void)hipSetDevice(c->device);
    for (auto s : c->streams) (void)hipStreamDestroy(s);
    (void)hipEventDestroy(c->completeStart);
    (void)hipEventDestroy(c->completeEnd);
    (void)hipEventDestroy(c->individualStart);
    (void)hipEventDestroy(c->individualEnd);
    for (auto e : c->kev) (void)hipEventDestroy(e);
    for (auto s : c->aux) (void)hipStreamDestroy(s);
    for (auto e : c->aux_done) (void)hipEventDestroy(e);
    if (c->fork) (void)hipEventDestroy(c->fork);
    if (c->arena) (void)hipFree(c->arena);
    if (c->d_stats) (void)hipFree(c->d_stats);
    if (c->h_stats) (void)hipHostFree(c->h_stats);
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

int speck_config_set_stream(speck_config* c, void* hip_stream)
{
    if (!c) return SPECK_ERR_INVALID;
    c->user_stream = static_cast<hipStream_t>(hip_stream);
    c->use_user_stream = hip_stream != nullptr;
    return SPECK_OK;
}

int speck_config_set_option(speck_config* c, const char* name, int64_t value)
{
    if (!c || !name) return SPECK_ERR_INVALID;
    const std::string n(name);
    if (n == "sym_bitmap_ratio") c->cp.sym_bitmap_ratio = (u32)value;
    else if (n == "num_dense_ratio") c->cp.num_dense_ratio = (u32)value;
    else if (n == "num_global_passes") c->cp.num_global_passes = (u32)value;
    else if (n == "collect_bytes") c->cp.want_bytes = value != 0;        // per-class byte model
    else if (n == "concurrent_classes") c->concurrent_classes = value != 0;
    else return SPECK_ERR_INVALID;
    return SPECK_OK;
}

int speck_config_profile_kernels(speck_config* c, int enable)
{
    if (!c) return SPECK_ERR_INVALID;
    c->profile_kernels = enable != 0;
    return SPECK_OK;
}

int speck_last_stats(const speck_config* c, speck_stats* out)
{
    if (!c || !out) return SPECK_ERR_INVALID;
    *out = c->last;
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
    rc = ensure_arena(c, scratch_bytes(m));
    if (rc != SPECK_OK) return rc;
    Scratch sc = carve(c, m);  // partials / blk_base come from the arena, row arrays from the caller
    sc.row_ops = d_row_ops;
    sc.row_max_ops = d_row_max_ops;
    sc.row_col_min = d_row_col_min;
    sc.row_col_max = d_row_col_max;
    rc = run_analysis(c, s, A, B, sc, false, nullptr);
    if (rc != SPECK_OK) return rc;
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
    const u32 m = (u32)A->rows;
    if (A->nnz == 0 || B->nnz == 0) {
        HIP_TRY(hipMemsetAsync(d_row_offsets, 0, (size_t(m) + 1) * 4, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (h_nnz_c) *h_nnz_c = 0;
        return SPECK_OK;
    }
    rc = ensure_arena(c, scratch_bytes(m));
    if (rc != SPECK_OK) return rc;
    Scratch sc = carve(c, m);
    rc = run_analysis(c, s, A, B, sc, true, d_row_offsets);
    if (rc != SPECK_OK) return rc;
    size_t ev = 0;
    const bool prof = c->profile_kernels;
    c->profile_kernels = false;
    std::vector<ClassTiming> timing;
    rc = run_symbolic_kernels(c, s, A, B, sc, d_row_offsets, &ev, &timing);
    c->profile_kernels = prof;
    if (rc != SPECK_OK) return rc;
    launch_scan(s, d_row_offsets, m, sc.tile_sums, A->row_offsets, sc.row_ops, sc.row_col_min,
                sc.row_col_max, nullptr, sc.partials, sc.blk_base, sc.bin_rows, c->d_stats, c->cp, 8);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(c->h_stats, c->d_stats, sizeof(DeviceStats), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
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
    const u32 m = (u32)A->rows;
    rc = ensure_arena(c, scratch_bytes(m));
    if (rc != SPECK_OK) return rc;
    Scratch sc = carve(c, m);
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
    int rc = speck_dcsr_alloc(dst, rows, cols, nnz, 1, value_size);
    if (rc != SPECK_OK) return rc;
    if (nnz) {
        HIP_TRY(hipMemcpy(dst->data, h_data, nnz * value_size, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dst->col_ids, h_col_ids, nnz * 4, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(dst->row_offsets, h_row_offsets, (rows + 1) * 4, hipMemcpyHostToDevice));
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
    }
    return "unknown";
}

const char* speck_version(void) { return "speck_amd 0.1 (gfx950)"; }

}  // extern "C"
