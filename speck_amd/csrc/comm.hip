// comm.hip -- row-sharded SpGEMM across the GPUs of one node: the ONE exchange step of the path, a gatherv of
// the C shards to a root rank (the reference is single-GPU, source/Executor.cpp:25).
// One process per GPU.  Rank p multiplies the row range [b_p, b_{p+1}) of A (a view, speck_partition_rows) with
// a replicated B; its shard has LOCAL row offsets.  The exchange:
//   sizes    ncclAllGather of (rows_p, nnz_p)                    -> displacements r_off / n_off on every rank
//   gatherv  grouped ncclSend / ncclRecv (RCCL has no gatherv; every peer -> root transfer rides one xGMI link)
//            of row_offsets[0 .. rows_p), col_ids, data into the root's concatenated buffers at r_off[p] / n_off[p]
//   rebase   the root adds n_off[p] to the offsets it received from rank p (one small kernel) and closes the
//            array with the total nnz
// A PLAN keeps the sizes, the root's output buffers (one set per slot) and an event per slot, so that a repeated
// exchange posts its transfers and returns: they run on the plan's own stream under the next multiply.
// Transports: RCCL over xGMI (librccl is dlopen'ed: the library has no link-time dependency on it, and a process
// that already holds PyTorch's copy shares it), and a host-staged one through POSIX shared memory for ranks
// that cannot form an RCCL communicator (several ranks on ONE GPU: RCCL rejects duplicate devices) -- the same
// plan / displacement / rebase code on both.
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/speck_c_api.h"
#include "comm_layout.hpp"

namespace {

#define COMM_HIP(expr)                                                                            \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            std::fprintf(stderr, "speck_amd: HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); \
            return _e == hipErrorOutOfMemory ? SPECK_ERR_OOM : SPECK_ERR_HIP;                     \
        }                                                                                         \
    } while (0)

// ---- RCCL, resolved at run time
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl* rccl()
{
    static Rccl r;
    static bool tried = false;
    if (tried) return r.handle ? &r : nullptr;
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
    }
    if (!r.handle) return nullptr;
    bool ok = true;
    auto sym = [&](const char* n) {
        void* p = dlsym(r.handle, n);
        if (!p) ok = false;
        return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) {
        dlclose(r.handle);
        r.handle = nullptr;
        return nullptr;
    }
    return &r;
}
#define COMM_NCCL(expr)                                                                                   \
    do {                                                                                                  \
        ncclResult_t _r = (expr);                                                                         \
        if (_r != ncclSuccess) {                                                                          \
            std::fprintf(stderr, "speck_amd: RCCL error %s at %s:%d\n", rccl()->GetErrorString(_r), __FILE__, __LINE__); \
            return SPECK_ERR_COMM;                                                                        \
        }                                                                                                 \
    } while (0)

// ---- host-staged transport: a control block + one data segment per (rank, slot) in POSIX shared memory
constexpr int kMaxRanks = 64, kMaxSlots = 4;
struct ShmControl {
    std::atomic<uint32_t> arrived;
    std::atomic<uint32_t> bar_count, bar_gen;
    std::atomic<uint64_t> sizes[kMaxRanks][3];         // rows, nnz, the plan id the rank is creating
    std::atomic<uint64_t> sizes_gen[kMaxRanks];
    std::atomic<uint64_t> ready[kMaxRanks][kMaxSlots];  // generation the rank's segment of a slot holds
    std::atomic<uint64_t> taken[kMaxRanks][kMaxSlots];  // generation the root has consumed
};
struct Segment {
    void* p = nullptr;
    size_t bytes = 0;
    std::string name;
    bool owner = false;
};
bool map_segment(Segment& s, const std::string& name, size_t bytes, bool create)
{
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(60);
    int fd = -1;
    while (true) {
        fd = create ? shm_open(name.c_str(), O_CREAT | O_RDWR, 0600) : shm_open(name.c_str(), O_RDWR, 0600);
        if (fd >= 0) {
            struct stat stt;
            if (create) {
                if (ftruncate(fd, (off_t)bytes) != 0) {
                    close(fd);
                    return false;
                }
                break;
            }
            if (fstat(fd, &stt) == 0 && (size_t)stt.st_size >= bytes) break;  // the creator has sized it
            close(fd);
            fd = -1;
        }
        if (std::chrono::steady_clock::now() > deadline) return false;
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    void* p = mmap(nullptr, bytes ? bytes : 1, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return false;
    s.p = p;
    s.bytes = bytes;
    s.name = name;
    s.owner = create;
    return true;
}
void unmap_segment(Segment& s)
{
    if (s.p) munmap(s.p, s.bytes ? s.bytes : 1);
    if (s.owner && !s.name.empty()) shm_unlink(s.name.c_str());
    s = Segment{};
}
template <typename Pred>
bool spin_until(Pred&& done, int seconds = 120)
{
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(seconds);
    for (int i = 0; !done(); ++i) {
        if (i > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if ((i & 1023) == 0 && std::chrono::steady_clock::now() > deadline) return false;
    }
    return true;
}

constexpr char kShmMagic[8] = {'S', 'P', 'K', 'H', 'O', 'S', 'T', '1'};

}  // namespace

struct speck_comm {
    int nranks = 1, rank = 0, transport = SPECK_TRANSPORT_RCCL, device = 0;
    ncclComm_t nccl = nullptr;
    hipStream_t stream = nullptr;
    uint64_t* d_sizes = nullptr;  // RCCL: 3 (mine: rows, nnz, plan id) + 3 * nranks (everyone's)
    uint64_t* h_sizes = nullptr;  // ... and their PINNED host mirror: asynchronous copies never touch pageable memory
                                  //   (the runtime locks such pages behind the caller's back; a later copy of other
                                  //   pageable memory then faulted on the GPU -- found with the replay stress)
    std::string shm_token;
    Segment ctl;
    uint64_t sizes_gen = 0;
    uint64_t next_plan_id = 1;    // plans are created collectively: every rank counts them per COMMUNICATOR (a process-wide
                                  //   counter disagrees as soon as one rank owns a second communicator or made a plan the
                                  //   others did not); the sizes exchange carries the id and every rank checks it
    ShmControl* control() const { return static_cast<ShmControl*>(ctl.p); }
};

struct speck_gather_plan {
    speck_comm* comm = nullptr;
    int root = 0, slots = 1;
    size_t vsize = 8;
    uint64_t cols = 0;
    speck::GatherLayout lay;
    std::vector<speck_dcsr> out;          // root: concatenated buffers, one set per slot
    std::vector<hipEvent_t> done;
    std::vector<char> pending;
    std::vector<uint64_t> gen;            // host-staged transport: generation of each slot
    std::vector<Segment> mine;            // ... my segment of each slot
    std::vector<std::vector<Segment>> theirs;  // ... root: every rank's segment of each slot
    uint64_t* d_off = nullptr;            // root: r_off | n_off on the device (rebase kernel)
    uint32_t* d_zero_ro = nullptr;        // rows(mine) zero offsets: what a shard WITHOUT products stands for -- the multiply
                                          //   leaves such a C with nnz = 0 and no row_offsets at all (Multiply.cu:67-70, 256-261)
    uint64_t id = 0;
};

namespace {

// out.row_offsets holds, per rank segment, LOCAL offsets: add the rank's displacement; close with the total.
__global__ __launch_bounds__(256) void rebase_offsets_kernel(uint32_t* __restrict__ ro, const uint64_t* __restrict__ r_off,
                                                             const uint64_t* __restrict__ n_off, int nranks, int skip_rank)
{
    const uint64_t total_rows = r_off[nranks];
    for (uint64_t r = uint64_t(blockIdx.x) * 256 + threadIdx.x; r <= total_rows; r += uint64_t(gridDim.x) * 256) {
        if (r == total_rows) {
            ro[r] = (uint32_t)n_off[nranks];
            continue;
        }
        int lo = 0, hi = nranks - 1;  // the rank whose row range holds r (empty ranges are skipped by the search)
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (r_off[mid] <= r) lo = mid; else hi = mid - 1;
        }
        (void)skip_rank;
        ro[r] += (uint32_t)n_off[lo];
    }
}

uint64_t segment_bytes(uint64_t rows, uint64_t nnz, size_t vsize) { return rows * 4 + nnz * 4 + nnz * vsize + 64; }

// all ranks of a host-staged communicator (generation counter: reusable)
int shm_barrier(speck_comm* c)
{
    ShmControl* k = c->control();
    const uint32_t bg = k->bar_gen.load();
    if (k->bar_count.fetch_add(1) + 1 == (uint32_t)c->nranks) {
        k->bar_count.store(0);
        k->bar_gen.fetch_add(1);
    } else if (!spin_until([&] { return k->bar_gen.load() != bg; }))
        return SPECK_ERR_COMM;
    return SPECK_OK;
}

int exchange_sizes(speck_comm* c, uint64_t rows, uint64_t nnz, uint64_t plan_id, std::vector<uint64_t>& all_rows,
                   std::vector<uint64_t>& all_nnz)
{
    all_rows.assign(c->nranks, 0);
    all_nnz.assign(c->nranks, 0);
    if (c->nranks == 1) {
        all_rows[0] = rows;
        all_nnz[0] = nnz;
        return SPECK_OK;
    }
    bool same_plan = true;  // every rank is creating the plan with THIS id (else: a rank skipped or repeated a create)
    if (c->transport == SPECK_TRANSPORT_RCCL) {
        c->h_sizes[0] = rows;
        c->h_sizes[1] = nnz;
        c->h_sizes[2] = plan_id;
        COMM_HIP(hipMemcpyAsync(c->d_sizes, c->h_sizes, 24, hipMemcpyHostToDevice, c->stream));
        COMM_NCCL(rccl()->AllGather(c->d_sizes, c->d_sizes + 3, 3, ncclUint64, c->nccl, c->stream));
        uint64_t* all = c->h_sizes + 3;
        COMM_HIP(hipMemcpyAsync(all, c->d_sizes + 3, size_t(3 * c->nranks) * 8, hipMemcpyDeviceToHost, c->stream));
        COMM_HIP(hipStreamSynchronize(c->stream));
        for (int p = 0; p < c->nranks; ++p) {
            all_rows[p] = all[3 * p];
            all_nnz[p] = all[3 * p + 1];
            same_plan = same_plan && all[3 * p + 2] == plan_id;
        }
        return same_plan ? SPECK_OK : SPECK_ERR_COMM;
    }
    ShmControl* k = c->control();
    const uint64_t g = ++c->sizes_gen;
    k->sizes[c->rank][0].store(rows);
    k->sizes[c->rank][1].store(nnz);
    k->sizes[c->rank][2].store(plan_id);
    k->sizes_gen[c->rank].store(g, std::memory_order_release);
    for (int p = 0; p < c->nranks; ++p) {
        if (!spin_until([&] { return k->sizes_gen[p].load(std::memory_order_acquire) >= g; })) return SPECK_ERR_COMM;
        all_rows[p] = k->sizes[p][0].load();
        all_nnz[p] = k->sizes[p][1].load();
        same_plan = same_plan && k->sizes[p][2].load() == plan_id;
    }
    // nobody starts the next exchange of sizes before everyone has read this one
    const int rc = shm_barrier(c);
    return rc != SPECK_OK ? rc : (same_plan ? SPECK_OK : SPECK_ERR_COMM);
}

}  // namespace

extern "C" {

int speck_comm_unique_id(int transport, void* id128)
{
    if (!id128) return SPECK_ERR_INVALID;
    std::memset(id128, 0, 128);
    if (transport == SPECK_TRANSPORT_RCCL) {
        if (!rccl()) return SPECK_ERR_COMM;
        static_assert(sizeof(ncclUniqueId) <= 128, "unique id fits the 128-byte token");
        ncclUniqueId id;
        COMM_NCCL(rccl()->GetUniqueId(&id));
        std::memcpy(id128, &id, sizeof(id));
        return SPECK_OK;
    }
    if (transport != SPECK_TRANSPORT_HOSTMEM) return SPECK_ERR_INVALID;
    std::random_device rd;
    const uint64_t token = (uint64_t(rd()) << 32) ^ rd() ^ (uint64_t(getpid()) << 17);
    std::memcpy(id128, kShmMagic, 8);
    std::memcpy(static_cast<char*>(id128) + 8, &token, 8);
    return SPECK_OK;
}

int speck_comm_init(int device, int nranks, int rank, int transport, const void* id128, speck_comm** out)
{
    if (!out || !id128 || nranks < 1 || rank < 0 || rank >= nranks || nranks > kMaxRanks) return SPECK_ERR_INVALID;
    if (transport != SPECK_TRANSPORT_RCCL && transport != SPECK_TRANSPORT_HOSTMEM) return SPECK_ERR_INVALID;
    COMM_HIP(hipSetDevice(device));
    auto* c = new speck_comm();
    c->nranks = nranks;
    c->rank = rank;
    c->transport = transport;
    c->device = device;
    auto fail = [&](int code) {
        (void)speck_comm_destroy(c);  // releases whatever exists so far
        return code;
    };
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return fail(SPECK_ERR_HIP);
    if (transport == SPECK_TRANSPORT_RCCL) {
        if (!rccl()) return fail(SPECK_ERR_COMM);
        ncclUniqueId id;
        std::memcpy(&id, id128, sizeof(id));
        if (rccl()->CommInitRank(&c->nccl, nranks, id, rank) != ncclSuccess) {
            c->nccl = nullptr;
            return fail(SPECK_ERR_COMM);
        }
        if (hipMalloc(reinterpret_cast<void**>(&c->d_sizes), (3 + 3 * size_t(nranks)) * 8) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void**>(&c->h_sizes), (3 + 3 * size_t(nranks)) * 8, hipHostMallocDefault) !=
                hipSuccess)
            return fail(SPECK_ERR_OOM);
    } else {
        if (std::memcmp(id128, kShmMagic, 8) != 0) return fail(SPECK_ERR_INVALID);
        uint64_t token;
        std::memcpy(&token, static_cast<const char*>(id128) + 8, 8);
        char buf[64];
        std::snprintf(buf, sizeof(buf), "/speck_%016llx", (unsigned long long)token);
        c->shm_token = buf;
        if (!map_segment(c->ctl, c->shm_token + "_ctl", sizeof(ShmControl), rank == 0)) return fail(SPECK_ERR_COMM);
        ShmControl* k = c->control();  // a fresh segment is zero-filled: every counter starts at 0
        k->arrived.fetch_add(1);
        if (!spin_until([&] { return k->arrived.load() >= (uint32_t)nranks; })) return fail(SPECK_ERR_COMM);
    }
    *out = c;
    return SPECK_OK;
}

int speck_comm_destroy(speck_comm* c)
{
    if (!c) return SPECK_ERR_INVALID;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->nccl) (void)rccl()->CommDestroy(c->nccl);
    if (c->d_sizes) (void)hipFree(c->d_sizes);
    if (c->h_sizes) (void)hipHostFree(c->h_sizes);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    unmap_segment(c->ctl);
    delete c;
    return SPECK_OK;
}

int speck_comm_info(const speck_comm* c, int* nranks, int* rank, int* transport)
{
    if (!c) return SPECK_ERR_INVALID;
    if (nranks) *nranks = c->nranks;
    if (rank) *rank = c->rank;
    if (transport) *transport = c->transport;
    return SPECK_OK;
}

static int fill_plan(speck_gather_plan* p, speck_comm* c, int root, uint64_t rows_local, uint64_t cols,
                     uint64_t nnz_local, size_t value_size, int slots);

int speck_gather_plan_create(speck_comm* c, int root, uint64_t rows_local, uint64_t cols, uint64_t nnz_local,
                             size_t value_size, int slots, speck_gather_plan** out)
{
    if (!c || !out || root < 0 || root >= c->nranks || slots < 1 || slots > kMaxSlots ||
        (value_size != 4 && value_size != 8))
        return SPECK_ERR_INVALID;
    auto* p = new speck_gather_plan();
    const int rc = fill_plan(p, c, root, rows_local, cols, nnz_local, value_size, slots);
    if (rc != SPECK_OK) {
        (void)speck_gather_plan_destroy(p);  // whatever was allocated so far
        return rc;
    }
    *out = p;
    return SPECK_OK;
}

static int fill_plan(speck_gather_plan* p, speck_comm* c, int root, uint64_t rows_local, uint64_t cols,
                     uint64_t nnz_local, size_t value_size, int slots)
{
    COMM_HIP(hipSetDevice(c->device));
    p->comm = c;
    p->root = root;
    p->slots = slots;
    p->vsize = value_size;
    p->cols = cols;
    p->id = c->next_plan_id++;  // (counted at ENTRY: a create that fails later on one rank still advances every rank)
    std::vector<uint64_t> all_rows, all_nnz;
    p->pending.assign(slots, 0);
    p->gen.assign(slots, 0);
    int rc = exchange_sizes(c, rows_local, nnz_local, p->id, all_rows, all_nnz);
    if (rc != SPECK_OK) return rc;
    // the concatenation must fit the u32 row offsets of the dCSR layout
    if (!speck::gather_layout(all_rows.data(), all_nnz.data(), c->nranks, &p->lay)) return SPECK_ERR_NNZ_OVERFLOW;
    for (int i = 0; i < slots; ++i) {
        hipEvent_t e = nullptr;
        COMM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        p->done.push_back(e);
    }
    const bool is_root = c->rank == root;
    if (is_root) {
        p->out.resize(slots);
        for (auto& o : p->out) {
            o = speck_dcsr{};
            rc = speck_dcsr_alloc(&o, p->lay.total_rows, cols, p->lay.total_nnz, 1, value_size);
            if (rc != SPECK_OK) return rc;
        }
        COMM_HIP(hipMalloc(reinterpret_cast<void**>(&p->d_off), 2 * (size_t(c->nranks) + 1) * 8));
        COMM_HIP(hipMemcpy(p->d_off, p->lay.r_off.data(), (size_t(c->nranks) + 1) * 8, hipMemcpyHostToDevice));
        COMM_HIP(hipMemcpy(p->d_off + c->nranks + 1, p->lay.n_off.data(), (size_t(c->nranks) + 1) * 8,
                           hipMemcpyHostToDevice));
    }
    if (c->transport == SPECK_TRANSPORT_HOSTMEM && c->nranks > 1) {
        // every plan of a communicator is created collectively and in the same order: p->id is the communicator's own
        // count and the sizes exchange above has checked that every rank holds the same one
        auto seg_name = [&](int rank, int slot) {
            return c->shm_token + "_p" + std::to_string(p->id) + "_r" + std::to_string(rank) + "_s" + std::to_string(slot);
        };
        if (!is_root) {
            p->mine.resize(slots);
            for (int s = 0; s < slots; ++s)
                if (!map_segment(p->mine[s], seg_name(c->rank, s), segment_bytes(rows_local, nnz_local, value_size), true))
                    return SPECK_ERR_COMM;
        } else {
            p->theirs.assign(slots, std::vector<Segment>(c->nranks));
            for (int s = 0; s < slots; ++s)
                for (int r = 0; r < c->nranks; ++r)
                    if (r != root && !map_segment(p->theirs[s][r], seg_name(r, s),
                                                  segment_bytes(all_rows[r], all_nnz[r], value_size), false))
                        return SPECK_ERR_COMM;
        }
        // every segment exists and is mapped by both sides before anyone moves on (its owner unlinks the name
        // when the plan goes)
        if (shm_barrier(c) != SPECK_OK) return SPECK_ERR_COMM;
    }
    return SPECK_OK;
}

int speck_gather_start(speck_gather_plan* p, int slot, const speck_dcsr* shard)
{
    if (!p || !shard || slot < 0 || slot >= p->slots) return SPECK_ERR_INVALID;
    speck_comm* c = p->comm;
    if (p->pending[slot]) return SPECK_ERR_INVALID;  // wait for the slot first
    // A shard whose rows of A hold no products: the multiply returned nnz = 0 and either no row_offsets (and, on the
    // nnz(A) == 0 early-out, not even the row count).  It stands for rows(plan) rows of zero entries.
    const bool no_products = shard->nnz == 0 && p->lay.nnz[c->rank] == 0 && (shard->row_offsets == nullptr || shard->rows == 0);
    if (!no_products && (shard->rows != p->lay.rows[c->rank] || shard->nnz != p->lay.nnz[c->rank])) return SPECK_ERR_INVALID;
    COMM_HIP(hipSetDevice(c->device));
    const bool is_root = c->rank == p->root;
    const uint64_t rows = p->lay.rows[c->rank], nnz = p->lay.nnz[c->rank];
    hipStream_t s = c->stream;
    const uint32_t* my_ro = shard->row_offsets;
    if (no_products && rows) {
        if (!p->d_zero_ro) {
            COMM_HIP(hipMalloc(reinterpret_cast<void**>(&p->d_zero_ro), rows * 4));
            COMM_HIP(hipMemset(p->d_zero_ro, 0, rows * 4));
        }
        my_ro = p->d_zero_ro;
    }
    if (is_root) {
        // my own shard: device-to-device, at my displacement
        speck_dcsr& o = p->out[slot];
        const uint64_t r0 = p->lay.r_off[c->rank], n0 = p->lay.n_off[c->rank];
        if (rows) COMM_HIP(hipMemcpyAsync(o.row_offsets + r0, my_ro, rows * 4, hipMemcpyDeviceToDevice, s));
        if (nnz) {
            COMM_HIP(hipMemcpyAsync(o.col_ids + n0, shard->col_ids, nnz * 4, hipMemcpyDeviceToDevice, s));
            COMM_HIP(hipMemcpyAsync(static_cast<char*>(o.data) + n0 * p->vsize, shard->data, nnz * p->vsize,
                                    hipMemcpyDeviceToDevice, s));
        }
    }
    if (c->nranks > 1 && c->transport == SPECK_TRANSPORT_RCCL) {
        Rccl* n = rccl();
        COMM_NCCL(n->GroupStart());
        // (no early return inside the group: a thread that leaves it open queues every later collective -- the sizes
        //  all-gather of the next plan included -- and never issues it.  The first error is kept, the group is closed.)
        ncclResult_t first = ncclSuccess;
        auto post = [&](ncclResult_t r) {
            if (r != ncclSuccess && first == ncclSuccess) first = r;
        };
        if (!is_root) {
            if (rows) post(n->Send(my_ro, rows, ncclUint32, p->root, c->nccl, s));
            if (nnz) {
                post(n->Send(shard->col_ids, nnz, ncclUint32, p->root, c->nccl, s));
                post(n->Send(shard->data, nnz * p->vsize, ncclUint8, p->root, c->nccl, s));
            }
        } else {
            speck_dcsr& o = p->out[slot];
            for (int r = 0; r < c->nranks; ++r) {
                if (r == p->root) continue;
                const uint64_t r0 = p->lay.r_off[r], n0 = p->lay.n_off[r];
                if (p->lay.rows[r]) post(n->Recv(o.row_offsets + r0, p->lay.rows[r], ncclUint32, r, c->nccl, s));
                if (p->lay.nnz[r]) {
                    post(n->Recv(o.col_ids + n0, p->lay.nnz[r], ncclUint32, r, c->nccl, s));
                    post(n->Recv(static_cast<char*>(o.data) + n0 * p->vsize, p->lay.nnz[r] * p->vsize, ncclUint8, r,
                                 c->nccl, s));
                }
            }
        }
        post(n->GroupEnd());
        COMM_NCCL(first);
    } else if (c->nranks > 1 && !is_root) {
        // host-staged: my shard into my segment of the slot (once the root has taken the previous content)
        ShmControl* k = c->control();
        const uint64_t g = p->gen[slot];
        const uint64_t tag = (p->id << 32) | g;  // generations are per plan
        if (g && !spin_until([&] { return k->taken[c->rank][slot].load(std::memory_order_acquire) == tag; }))
            return SPECK_ERR_COMM;
        char* base = static_cast<char*>(p->mine[slot].p);
        if (rows && no_products) std::memset(base, 0, rows * 4);
        else if (rows) COMM_HIP(hipMemcpy(base, my_ro, rows * 4, hipMemcpyDeviceToHost));
        if (nnz) {
            COMM_HIP(hipMemcpy(base + rows * 4, shard->col_ids, nnz * 4, hipMemcpyDeviceToHost));
            COMM_HIP(hipMemcpy(base + rows * 4 + nnz * 4, shard->data, nnz * p->vsize, hipMemcpyDeviceToHost));
        }
        k->ready[c->rank][slot].store((p->id << 32) | (g + 1), std::memory_order_release);
    }
    ++p->gen[slot];
    if (is_root && (c->transport == SPECK_TRANSPORT_RCCL || c->nranks == 1)) {
        hipLaunchKernelGGL(rebase_offsets_kernel, dim3(256), dim3(256), 0, s, p->out[slot].row_offsets, p->d_off,
                           p->d_off + c->nranks + 1, c->nranks, -1);
        COMM_HIP(hipGetLastError());
    }
    COMM_HIP(hipEventRecord(p->done[slot], s));
    p->pending[slot] = 1;
    return SPECK_OK;
}

int speck_gather_wait(speck_gather_plan* p, int slot, speck_dcsr* full_view)
{
    if (!p || slot < 0 || slot >= p->slots) return SPECK_ERR_INVALID;
    speck_comm* c = p->comm;
    if (full_view) *full_view = speck_dcsr{};
    if (!p->pending[slot]) return SPECK_OK;
    COMM_HIP(hipSetDevice(c->device));
    const bool is_root = c->rank == p->root;
    if (is_root && c->nranks > 1 && c->transport == SPECK_TRANSPORT_HOSTMEM) {
        ShmControl* k = c->control();
        const uint64_t tag = (p->id << 32) | p->gen[slot];
        speck_dcsr& o = p->out[slot];
        for (int r = 0; r < c->nranks; ++r) {
            if (r == p->root) continue;
            if (!spin_until([&] { return k->ready[r][slot].load(std::memory_order_acquire) == tag; })) return SPECK_ERR_COMM;
            const char* base = static_cast<const char*>(p->theirs[slot][r].p);
            const uint64_t rows = p->lay.rows[r], nnz = p->lay.nnz[r], r0 = p->lay.r_off[r], n0 = p->lay.n_off[r];
            // (blocking copies: the source is pageable shared memory that is unmapped later -- see h_sizes)
            if (rows) COMM_HIP(hipMemcpy(o.row_offsets + r0, base, rows * 4, hipMemcpyHostToDevice));
            if (nnz) {
                COMM_HIP(hipMemcpy(o.col_ids + n0, base + rows * 4, nnz * 4, hipMemcpyHostToDevice));
                COMM_HIP(hipMemcpy(static_cast<char*>(o.data) + n0 * p->vsize, base + rows * 4 + nnz * 4,
                                   nnz * p->vsize, hipMemcpyHostToDevice));
            }
            // the segment may be refilled once it is marked taken
            k->taken[r][slot].store(tag, std::memory_order_release);
        }
        hipLaunchKernelGGL(rebase_offsets_kernel, dim3(256), dim3(256), 0, c->stream, o.row_offsets, p->d_off,
                           p->d_off + c->nranks + 1, c->nranks, -1);
        COMM_HIP(hipGetLastError());
        COMM_HIP(hipEventRecord(p->done[slot], c->stream));
    }
    COMM_HIP(hipEventSynchronize(p->done[slot]));  // an event, not a whole-stream synchronisation
    p->pending[slot] = 0;
    if (is_root && full_view) *full_view = p->out[slot];  // a VIEW: the plan owns the buffers
    return SPECK_OK;
}

int speck_gather_plan_layout(const speck_gather_plan* p, uint64_t* row_displs, uint64_t* nnz_displs)
{
    if (!p) return SPECK_ERR_INVALID;
    const int n = p->comm->nranks;
    if (row_displs) std::memcpy(row_displs, p->lay.r_off.data(), (size_t(n) + 1) * 8);
    if (nnz_displs) std::memcpy(nnz_displs, p->lay.n_off.data(), (size_t(n) + 1) * 8);
    return SPECK_OK;
}

int speck_gather_plan_destroy(speck_gather_plan* p)
{
    if (!p) return SPECK_ERR_INVALID;
    if (!p->comm) {  // never got as far as a communicator
        delete p;
        return SPECK_OK;
    }
    (void)hipSetDevice(p->comm->device);
    for (int s = 0; s < (int)p->done.size() && s < (int)p->pending.size(); ++s) (void)speck_gather_wait(p, s, nullptr);
    // host-staged transport: my segments stay until the root has taken what I put there
    if (p->comm->transport == SPECK_TRANSPORT_HOSTMEM && p->comm->nranks > 1 && p->comm->rank != p->root &&
        !p->mine.empty()) {
        ShmControl* k = p->comm->control();
        for (int s = 0; s < (int)p->mine.size(); ++s)
            if (p->gen[s])
                (void)spin_until([&] { return k->taken[p->comm->rank][s].load(std::memory_order_acquire) ==
                                              ((p->id << 32) | p->gen[s]); }, 30);
    }
    for (auto& o : p->out) (void)speck_dcsr_free(&o);
    for (auto& e : p->done) (void)hipEventDestroy(e);
    for (auto& m : p->mine) unmap_segment(m);
    for (auto& v : p->theirs)
        for (auto& m : v) unmap_segment(m);
    if (p->d_off) (void)hipFree(p->d_off);
    if (p->d_zero_ro) (void)hipFree(p->d_zero_ro);
    delete p;
    return SPECK_OK;
}

int speck_gatherv_csr(speck_comm* c, int root, const speck_dcsr* shard, uint64_t cols, size_t value_size, speck_dcsr* full)
{
    if (!c || !shard) return SPECK_ERR_INVALID;
    speck_gather_plan* p = nullptr;
    int rc = speck_gather_plan_create(c, root, shard->rows, cols, shard->nnz, value_size, 1, &p);
    if (rc != SPECK_OK) return rc;
    rc = speck_gather_start(p, 0, shard);
    speck_dcsr view{};
    if (rc == SPECK_OK) rc = speck_gather_wait(p, 0, &view);
    if (rc == SPECK_OK && c->rank == root && full) {
        (void)speck_dcsr_free(full);
        *full = view;        // ownership moves to the caller
        p->out.clear();
    }
    (void)speck_gather_plan_destroy(p);
    return rc;
}

}  // extern "C"
