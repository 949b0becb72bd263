// runspECK -- the reference's benchmark driver (source/runspECK.cpp:13-32, source/Executor.cpp:13-81,
// source/RunConfig.cpp:8-23, source/DataLoader.cpp:24-75) on top of the MI355X backend.
//
//   runspECK <matrix.mtx | gen:<kind>[:scale[:seed]]> [config.ini] [--gpus N] [--shared-gpu] [--time-library]
//
// --gpus N (new: the reference is single-GPU, source/Executor.cpp:25): the driver re-launches itself as N rank
// processes, one per GPU.  Every rank loads the matrix, multiplies its row range of A (speck_partition_rows: equal
// intermediate products) with the replicated B, and ONE exchange per multiply concatenates the shards on rank 0 --
// speck_gather_plan of the C ABI: RCCL all-gather of the sizes + grouped send / recv over xGMI, two slots so that
// the exchange of iteration k runs under the multiply of iteration k + 1.  Rank 0 prints the usual two lines for
// the CONCATENATED product (and compares it with rocSPARSE when CompareResult is on).  --shared-gpu puts every
// rank on GPU 0 with the library's host-staged transport (plumbing check on a one-GPU box).
//
// Same behaviour: loads "<path>d_.hicsr" if present, else the .mtx (and writes the cache);
// B = A when square, else A^T; IterationsWarmUp + IterationsExecution calls of
// spECK::MultiplyspECK<double,4,1024,DYN,STATIC> with the SAME matOut and config; prints
//   var-SpGEMM -> NNZ: <nnz(C)>
//   var-SpGEMM SpGEMM: <mean complete ms> ms
// INI keys honoured (the six the reference reads, Executor.cpp:15-29, RunConfig.cpp:22):
// IterationsWarmUp, IterationsExecution, TrackIndividualTimes, TrackCompleteTimes,
// CompareResult, InputFile.  CompareResult checks C against rocSPARSE SpGEMM (the reference
// checks against cuSPARSE, Executor.cpp:29-40): row lengths and column ids bit-exact, values 1e-10.
// "gen:" inputs are the synthetic SuiteSparse stand-ins (no network in the build image).
#include <hip/hip_runtime.h>
#include <rocsparse/rocsparse.h>
#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <chrono>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "CSR.h"
#include "Compare.h"
#include "Multiply.h"
#include "Transpose.h"

extern char** environ;

namespace {

std::map<std::string, std::string> read_ini(const char* path)
{
    std::map<std::string, std::string> kv;
    std::ifstream f(path);
    std::string line;
    while (std::getline(f, line)) {
        const size_t c = line.find_first_of(";#");
        if (c != std::string::npos) line = line.substr(0, c);
        const size_t eq = line.find('=');
        if (eq == std::string::npos) continue;
        auto trim = [](std::string s) {
            const char* ws = " \t\r\n";
            const size_t b = s.find_first_not_of(ws);
            if (b == std::string::npos) return std::string();
            return s.substr(b, s.find_last_not_of(ws) - b + 1);
        };
        kv[trim(line.substr(0, eq))] = trim(line.substr(eq + 1));
    }
    return kv;
}
int get_int(const std::map<std::string, std::string>& kv, const char* k, int def)
{
    auto it = kv.find(k);
    return it == kv.end() ? def : std::atoi(it->second.c_str());
}
bool get_bool(const std::map<std::string, std::string>& kv, const char* k, bool def)
{
    auto it = kv.find(k);
    if (it == kv.end()) return def;
    std::string v = it->second;
    std::transform(v.begin(), v.end(), v.begin(), ::tolower);
    return v == "true" || v == "1" || v == "yes" || v == "on";
}

#define RS(expr)                                                                   \
    do {                                                                           \
        rocsparse_status _s = (expr);                                              \
        if (_s != rocsparse_status_success) {                                      \
            std::printf("rocSPARSE error %d at %s:%d\n", (int)_s, __FILE__, __LINE__); \
            return false;                                                          \
        }                                                                          \
    } while (0)

// C_ref = A*B with rocSPARSE (generic SpGEMM + csrsort), into a dCSR<double>.
bool rocsparse_reference(const dCSR<double>& A, const dCSR<double>& B, dCSR<double>& C)
{
    rocsparse_handle h;
    RS(rocsparse_create_handle(&h));
    const double alpha = 1.0, beta = 0.0;
    C.alloc(A.rows, B.cols, 0, true);
    rocsparse_spmat_descr dA, dB, dC, dD;
    RS(rocsparse_create_csr_descr(&dA, A.rows, A.cols, A.nnz, A.row_offsets, A.col_ids, A.data,
                                  rocsparse_indextype_i32, rocsparse_indextype_i32, rocsparse_index_base_zero,
                                  rocsparse_datatype_f64_r));
    RS(rocsparse_create_csr_descr(&dB, B.rows, B.cols, B.nnz, B.row_offsets, B.col_ids, B.data,
                                  rocsparse_indextype_i32, rocsparse_indextype_i32, rocsparse_index_base_zero,
                                  rocsparse_datatype_f64_r));
    // nnz = 0 placeholders need valid pointers: C.alloc(.., 0, ..) keeps one element of each
    (void)hipMemset(C.row_offsets, 0, (A.rows + 1) * sizeof(unsigned int));
    RS(rocsparse_create_csr_descr(&dC, A.rows, B.cols, 0, C.row_offsets, C.col_ids, C.data, rocsparse_indextype_i32,
                                  rocsparse_indextype_i32, rocsparse_index_base_zero, rocsparse_datatype_f64_r));
    unsigned int* d_ro = nullptr;  // D is empty (beta = 0) but needs its own zeroed row offsets
    if (hipMalloc((void**)&d_ro, (A.rows + 1) * sizeof(unsigned int)) != hipSuccess) return false;
    (void)hipMemset(d_ro, 0, (A.rows + 1) * sizeof(unsigned int));
    RS(rocsparse_create_csr_descr(&dD, A.rows, B.cols, 0, d_ro, C.col_ids, C.data, rocsparse_indextype_i32,
                                  rocsparse_indextype_i32, rocsparse_index_base_zero, rocsparse_datatype_f64_r));
    size_t bytes = 0;
    RS(rocsparse_spgemm(h, rocsparse_operation_none, rocsparse_operation_none, &alpha, dA, dB, &beta, dD, dC,
                        rocsparse_datatype_f64_r, rocsparse_spgemm_alg_default, rocsparse_spgemm_stage_buffer_size,
                        &bytes, nullptr));
    void* buf = nullptr;
    if (hipMalloc(&buf, bytes ? bytes : 16) != hipSuccess) return false;
    RS(rocsparse_spgemm(h, rocsparse_operation_none, rocsparse_operation_none, &alpha, dA, dB, &beta, dD, dC,
                        rocsparse_datatype_f64_r, rocsparse_spgemm_alg_default, rocsparse_spgemm_stage_nnz, &bytes,
                        buf));
    int64_t r, c, nnz;
    RS(rocsparse_spmat_get_size(dC, &r, &c, &nnz));
    unsigned int* ro = C.row_offsets;
    C.row_offsets = nullptr;
    C.alloc(A.rows, B.cols, (size_t)nnz, false);
    C.row_offsets = ro;
    RS(rocsparse_csr_set_pointers(dC, C.row_offsets, C.col_ids, C.data));
    RS(rocsparse_spgemm(h, rocsparse_operation_none, rocsparse_operation_none, &alpha, dA, dB, &beta, dD, dC,
                        rocsparse_datatype_f64_r, rocsparse_spgemm_alg_default, rocsparse_spgemm_stage_compute, &bytes,
                        buf));
    // ascending column ids per row (our contract; rocSPARSE does not promise it)
    rocsparse_mat_descr md;
    RS(rocsparse_create_mat_descr(&md));
    size_t sbytes = 0;
    RS(rocsparse_csrsort_buffer_size(h, (rocsparse_int)A.rows, (rocsparse_int)B.cols, (rocsparse_int)nnz,
                                     (const rocsparse_int*)C.row_offsets, (const rocsparse_int*)C.col_ids, &sbytes));
    void* sbuf = nullptr;
    rocsparse_int* perm = nullptr;
    double* sorted = nullptr;
    if (hipMalloc(&sbuf, sbytes ? sbytes : 16) != hipSuccess) return false;
    if (hipMalloc((void**)&perm, (nnz ? nnz : 1) * sizeof(rocsparse_int)) != hipSuccess) return false;
    if (hipMalloc((void**)&sorted, (nnz ? nnz : 1) * sizeof(double)) != hipSuccess) return false;
    RS(rocsparse_create_identity_permutation(h, (rocsparse_int)nnz, perm));
    RS(rocsparse_csrsort(h, (rocsparse_int)A.rows, (rocsparse_int)B.cols, (rocsparse_int)nnz, md,
                         (const rocsparse_int*)C.row_offsets, (rocsparse_int*)C.col_ids, perm, sbuf));
    RS(rocsparse_dgthr(h, (rocsparse_int)nnz, C.data, sorted, perm, rocsparse_index_base_zero));
    (void)hipMemcpy(C.data, sorted, nnz * sizeof(double), hipMemcpyDeviceToDevice);
    (void)hipDeviceSynchronize();
    (void)hipFree(buf);
    (void)hipFree(d_ro);
    (void)hipFree(sbuf);
    (void)hipFree(perm);
    (void)hipFree(sorted);
    rocsparse_destroy_mat_descr(md);
    rocsparse_destroy_spmat_descr(dA);
    rocsparse_destroy_spmat_descr(dB);
    rocsparse_destroy_spmat_descr(dC);
    rocsparse_destroy_spmat_descr(dD);
    rocsparse_destroy_handle(h);
    return true;
}

// The same-box LIBRARY baseline (role of the reference's cuSPARSE product beside its own, source/Executor.cpp:29-40, here
// TIMED): rocSPARSE generic SpGEMM on the same device buffers, the reference's protocol -- `warm` untimed and `iters`
// timed products, work buffer and C allocated once and reused (the way the loop above reuses matOut) -- each product =
// the nnz stage + the compute stage (rocSPARSE's symbolic + numeric), HIP events around the pair.  Its rows come out
// UNSORTED (a csrsort would be extra); reported as it is.  Returns mean ms, < 0 on failure.
double rocsparse_timed(const dCSR<double>& A, const dCSR<double>& B, int warm, int iters, size_t* nnz_out)
{
#define RT(expr)                                                                   \
    do {                                                                           \
        rocsparse_status _s = (expr);                                              \
        if (_s != rocsparse_status_success) {                                      \
            std::printf("rocSPARSE error %d at %s:%d\n", (int)_s, __FILE__, __LINE__); \
            return -1.0;                                                           \
        }                                                                          \
    } while (0)
    rocsparse_handle h;
    RT(rocsparse_create_handle(&h));
    const double alpha = 1.0, beta = 0.0;
    rocsparse_spmat_descr dA, dB, dC, dD;
    RT(rocsparse_create_csr_descr(&dA, A.rows, A.cols, A.nnz, A.row_offsets, A.col_ids, A.data, rocsparse_indextype_i32,
                                  rocsparse_indextype_i32, rocsparse_index_base_zero, rocsparse_datatype_f64_r));
    RT(rocsparse_create_csr_descr(&dB, B.rows, B.cols, B.nnz, B.row_offsets, B.col_ids, B.data, rocsparse_indextype_i32,
                                  rocsparse_indextype_i32, rocsparse_index_base_zero, rocsparse_datatype_f64_r));
    unsigned int *c_ro = nullptr, *d_ro = nullptr, *c_col = nullptr;
    double* c_val = nullptr;
    if (hipMalloc((void**)&c_ro, (A.rows + 1) * 4) != hipSuccess || hipMalloc((void**)&d_ro, (A.rows + 1) * 4) != hipSuccess ||
        hipMalloc((void**)&c_col, 16) != hipSuccess || hipMalloc((void**)&c_val, 16) != hipSuccess)
        return -1.0;
    (void)hipMemset(c_ro, 0, (A.rows + 1) * 4);
    (void)hipMemset(d_ro, 0, (A.rows + 1) * 4);
    RT(rocsparse_create_csr_descr(&dC, A.rows, B.cols, 0, c_ro, c_col, c_val, rocsparse_indextype_i32, rocsparse_indextype_i32,
                                  rocsparse_index_base_zero, rocsparse_datatype_f64_r));
    RT(rocsparse_create_csr_descr(&dD, A.rows, B.cols, 0, d_ro, c_col, c_val, rocsparse_indextype_i32, rocsparse_indextype_i32,
                                  rocsparse_index_base_zero, rocsparse_datatype_f64_r));
    size_t bytes = 0;
    RT(rocsparse_spgemm(h, rocsparse_operation_none, rocsparse_operation_none, &alpha, dA, dB, &beta, dD, dC,
                        rocsparse_datatype_f64_r, rocsparse_spgemm_alg_default, rocsparse_spgemm_stage_buffer_size, &bytes, nullptr));
    void* buf = nullptr;
    if (hipMalloc(&buf, bytes ? bytes : 16) != hipSuccess) return -1.0;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    int64_t r = 0, c = 0, nnz = 0, cap = 0;
    double total_ms = 0.0;
    for (int it = 0; it < warm + iters; ++it) {
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, nullptr);
        RT(rocsparse_spgemm(h, rocsparse_operation_none, rocsparse_operation_none, &alpha, dA, dB, &beta, dD, dC,
                            rocsparse_datatype_f64_r, rocsparse_spgemm_alg_default, rocsparse_spgemm_stage_nnz, &bytes, buf));
        RT(rocsparse_spmat_get_size(dC, &r, &c, &nnz));
        if (nnz > cap) {  // (first product only: C is reused from then on, like matOut)
            (void)hipFree(c_col);
            (void)hipFree(c_val);
            if (hipMalloc((void**)&c_col, (size_t)nnz * 4) != hipSuccess || hipMalloc((void**)&c_val, (size_t)nnz * 8) != hipSuccess)
                return -1.0;
            cap = nnz;
        }
        RT(rocsparse_csr_set_pointers(dC, c_ro, c_col, c_val));
        RT(rocsparse_spgemm(h, rocsparse_operation_none, rocsparse_operation_none, &alpha, dA, dB, &beta, dD, dC,
                            rocsparse_datatype_f64_r, rocsparse_spgemm_alg_default, rocsparse_spgemm_stage_compute, &bytes, buf));
        (void)hipEventRecord(e1, nullptr);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (it >= warm) total_ms += ms;
    }
    if (nnz_out) *nnz_out = (size_t)nnz;
    (void)hipFree(buf);
    (void)hipFree(c_ro);
    (void)hipFree(d_ro);
    (void)hipFree(c_col);
    (void)hipFree(c_val);
    rocsparse_destroy_spmat_descr(dA);
    rocsparse_destroy_spmat_descr(dB);
    rocsparse_destroy_spmat_descr(dC);
    rocsparse_destroy_spmat_descr(dD);
    rocsparse_destroy_handle(h);
    return iters > 0 ? total_ms / iters : -1.0;
#undef RT
}

CSR<double> load_input(const std::string& path)
{
    if (path.rfind("gen:", 0) == 0) {
        std::stringstream ss(path.substr(4));
        std::string kind, tok;
        std::getline(ss, kind, ':');
        double scale = 1.0;
        uint64_t seed = 1;
        if (std::getline(ss, tok, ':')) scale = std::atof(tok.c_str());
        if (std::getline(ss, tok, ':')) seed = std::strtoull(tok.c_str(), nullptr, 10);
        speck_host_csr* h = nullptr;
        if (speck_gen_matrix(kind.c_str(), scale, seed, 1, &h) != SPECK_OK) throw std::runtime_error("unknown generator");
        return speck_detail::from_handle<double>(h);
    }
    // DataLoader.cpp:24-58
    const std::string csrPath = path + "d_" + ".hicsr";
    try {
        std::cout << "trying to load csr file \"" << csrPath << "\"\n";
        CSR<double> m = loadCSR<double>(csrPath.c_str());
        std::cout << "successfully loaded: \"" << csrPath << "\"\n";
        return m;
    } catch (std::exception& ex) {
        std::cout << "could not load csr file:\n\t" << ex.what() << "\n";
    }
    std::cout << "trying to load mtx file \"" << path << "\"\n";
    CSR<double> m = loadMTXasCSR<double>(path.c_str());
    std::cout << "successfully loaded and converted: \"" << csrPath << "\"\n";
    try {
        std::cout << "write csr file for future use\n";
        storeCSR(m, csrPath.c_str());
    } catch (std::exception& ex) {
        std::cout << ex.what() << std::endl;
    }
    return m;
}

// launcher: N copies of this executable, SPECK_RANK / SPECK_WORLD / SPECK_RENDEZVOUS in their environment
int launch_ranks(int argc, char* argv[], int gpus)
{
    char exe[4096];
    const ssize_t n = readlink("/proc/self/exe", exe, sizeof(exe) - 1);
    if (n <= 0) return 1;
    exe[n] = 0;
    const std::string rdv = "/tmp/speck_rdv_" + std::to_string(getpid());
    std::remove(rdv.c_str());
    std::vector<pid_t> pids;
    for (int r = 0; r < gpus; ++r) {
        std::vector<std::string> env_s;
        for (char** e = environ; *e; ++e) env_s.push_back(*e);
        env_s.push_back("SPECK_RANK=" + std::to_string(r));
        env_s.push_back("SPECK_WORLD=" + std::to_string(gpus));
        env_s.push_back("SPECK_RENDEZVOUS=" + rdv);
        std::vector<char*> envp;
        for (auto& x : env_s) envp.push_back(const_cast<char*>(x.c_str()));
        envp.push_back(nullptr);
        std::vector<char*> av(argv, argv + argc);
        av.push_back(nullptr);
        pid_t pid;
        if (posix_spawn(&pid, exe, nullptr, nullptr, av.data(), envp.data()) != 0) return 1;
        pids.push_back(pid);
    }
    int worst = 0;
    for (pid_t pid : pids) {
        int st = 0;
        waitpid(pid, &st, 0);
        const int rc = WIFEXITED(st) ? WEXITSTATUS(st) : 128;
        worst = std::max(worst, rc);
    }
    std::remove(rdv.c_str());
    return worst;
}

// rank 0 writes the 128-byte id (temp file + rename: readers never see a partial file), the others poll for it
bool rendezvous_id(const std::string& path, int rank, int transport, unsigned char (&id)[128])
{
    if (rank == 0) {
        if (speck_comm_unique_id(transport, id) != SPECK_OK) return false;
        const std::string tmp = path + ".tmp";
        std::ofstream f(tmp, std::ios::binary);
        f.write(reinterpret_cast<const char*>(id), 128);
        f.close();
        return std::rename(tmp.c_str(), path.c_str()) == 0;
    }
    for (int i = 0; i < 60000; ++i) {
        std::ifstream f(path, std::ios::binary);
        if (f && f.read(reinterpret_cast<char*>(id), 128)) return true;
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    return false;
}

}  // namespace

int main(int argc, char* argv[])
{
    // ---- options behind the two positional arguments of the reference
    int gpus = 1;
    bool shared_gpu = false, time_library = false;
    std::vector<char*> pos;
    for (int i = 0; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--gpus" && i + 1 < argc) gpus = std::atoi(argv[++i]);
        else if (a == "--shared-gpu") shared_gpu = true;
        else if (a == "--time-library") time_library = true;
        else pos.push_back(argv[i]);
    }
    const char* env_rank = std::getenv("SPECK_RANK");
    if (gpus > 1 && !env_rank) return launch_ranks(argc, argv, gpus);
    const int rank = env_rank ? std::atoi(env_rank) : 0;
    const int world = env_rank ? std::atoi(std::getenv("SPECK_WORLD")) : 1;
    const int device = shared_gpu ? 0 : rank;
    argc = (int)pos.size();
    argv = pos.data();
    if (argc < 2) {
        std::printf("no .mtx file path set. please call using 'runspECK /path/to/matrix.mtx [config.ini]'");
        return -1;
    }
    std::map<std::string, std::string> ini;
    if (argc > 2) ini = read_ini(argv[2]);
    std::string filePath = argv[1];
    if (ini.count("InputFile")) filePath = ini["InputFile"];  // RunConfig.cpp:22

    const int iterationsWarmup = get_int(ini, "IterationsWarmUp", 5);
    const int iterationsExecution = get_int(ini, "IterationsExecution", 10);
    const bool measureAll = get_bool(ini, "TrackIndividualTimes", false);
    const bool measureCompleteTimes = get_bool(ini, "TrackCompleteTimes", true);
    const bool compareResult = get_bool(ini, "CompareResult", false);

    try {
        CSR<double> cpuA = load_input(filePath);
        std::cout << "Matrix: " << cpuA.rows << "x" << cpuA.cols << ": " << cpuA.nnz << " nonzeros\n";
        dCSR<double> gpuA, gpuB, dCsrHiRes, dCsrReference, dCsrAbs;
        if (hipSetDevice(device) != hipSuccess) throw std::runtime_error("no such HIP device for this rank");
        convert(gpuA, cpuA, 0);
        if (gpuA.rows != gpuA.cols)
            spECK::Transpose(gpuA, gpuB);  // DataLoader.cpp:65-69
        else
            convert(gpuB, cpuA, 0);

        auto config = spECK::spECKConfig::initialize(device);
        if (world > 1) {
            // ------------------------------------------------------------ row-sharded run (one process per GPU)
            const int transport = shared_gpu ? SPECK_TRANSPORT_HOSTMEM : SPECK_TRANSPORT_RCCL;
            unsigned char id[128];
            if (!rendezvous_id(std::getenv("SPECK_RENDEZVOUS"), rank, transport, id)) throw std::runtime_error("rendezvous failed");
            speck_comm* comm = nullptr;
            if (speck_comm_init(device, world, rank, transport, id, &comm) != SPECK_OK) throw std::runtime_error("speck_comm_init failed");
            speck_dcsr a = gpuA.raw(), b = gpuB.raw();
            std::vector<uint64_t> bounds(world + 1);
            if (speck_partition_rows(config.handle, &a, &b, world, bounds.data()) != SPECK_OK) throw std::runtime_error("partition failed");
            uint64_t products = 0;
            speck_analysis(config.handle, &a, &b, nullptr, nullptr, nullptr, nullptr, &products, nullptr);
            speck_dcsr mine = a;  // a VIEW of my rows: offsets stay absolute
            mine.rows = bounds[rank + 1] - bounds[rank];
            mine.row_offsets = a.row_offsets + bounds[rank];
            mine.nnz = cpuA.row_offsets[bounds[rank + 1]] - cpuA.row_offsets[bounds[rank]];
            // two output matrices (each with its own config: a replayed launch sequence is tied to its buffers)
            spECK::spECKConfig cfgs[2] = {config, spECK::spECKConfig::initialize(device)};
            speck_dcsr out[2] = {speck_dcsr{}, speck_dcsr{}};
            speck_gather_plan* plan = nullptr;
            speck_dcsr full{};
            int errors = 0;
            const int total = iterationsWarmup + iterationsExecution;
            std::chrono::steady_clock::time_point t0;
            for (int it = 0; it < total; ++it) {
                const int slot = it & 1;
                if (it == iterationsWarmup) {
                    if (plan) {  // drain, so that the timed region starts from an idle exchange
                        speck_gather_wait(plan, 0, nullptr);
                        speck_gather_wait(plan, 1, nullptr);
                    }
                    t0 = std::chrono::steady_clock::now();
                }
                if (plan && speck_gather_wait(plan, slot, nullptr) != SPECK_OK) throw std::runtime_error("gather wait failed");
                speck_timings tm{};
                int rc = speck_multiply_f64(cfgs[slot].handle, &mine, &b, &out[slot], &tm);
                if (rc != SPECK_OK) throw std::runtime_error(speck_status_string(rc));
                // (mine.rows, not out[slot].rows: a shard without products comes back with nnz = 0 and no row count)
                if (!plan && speck_gather_plan_create(comm, 0, mine.rows, b.cols, out[slot].nnz, 8, 2, &plan) != SPECK_OK)
                    throw std::runtime_error("gather plan failed");
                if (speck_gather_start(plan, slot, &out[slot]) != SPECK_OK) throw std::runtime_error("gather start failed");
            }
            speck_gather_wait(plan, total & 1, nullptr);
            if (speck_gather_wait(plan, (total - 1) & 1, &full) != SPECK_OK) throw std::runtime_error("gather wait failed");
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() /
                              std::max(1, iterationsExecution);
            if (rank == 0) {
                if (compareResult) {
                    dCSR<double> absA, absB, got;
                    CSR<double> cpuAbs;
                    convert(cpuAbs, cpuA, 0);
                    for (size_t i = 0; i < cpuAbs.nnz; ++i) cpuAbs.data[i] = std::fabs(cpuAbs.data[i]);
                    convert(absA, cpuAbs, 0);
                    if (gpuA.rows != gpuA.cols) spECK::Transpose(absA, absB); else convert(absB, cpuAbs, 0);
                    if (!rocsparse_reference(gpuA, gpuB, dCsrReference) || !rocsparse_reference(absA, absB, dCsrAbs)) {
                        std::printf("Error: rocSPARSE reference failed\n");
                        return 2;
                    }
                    speck_dcsr ref = dCsrReference.raw(), sc = dCsrAbs.raw();
                    uint64_t bad = 1, badv = 1;
                    speck_compare_bounded_f64(nullptr, &ref, &full, &sc, 1e-12, &bad, &badv);
                    if (bad || badv) {
                        std::printf("Error: concatenated matrix incorrect (%llu structure rows, %llu value rows)\n",
                                    (unsigned long long)bad, (unsigned long long)badv);
                        ++errors;
                    }
                }
                std::cout << std::setw(20) << "var-SpGEMM -> NNZ: " << full.nnz << std::endl;
                std::cout << std::setw(20) << "var-SpGEMM SpGEMM: " << ms << " ms" << std::endl;
                std::cout << std::setw(20) << "var-SpGEMM GFLOPS: " << 2.0 * (double)products / (ms * 1e6) << std::endl;
                std::cout << "row shards: " << world << " ranks, gatherv to rank 0 ("
                          << (shared_gpu ? "host-staged, ranks share GPU 0" : "RCCL") << ")" << std::endl;
                if (compareResult) std::cout << "compare vs rocSPARSE: " << (errors ? "FAILED" : "ok") << std::endl;
            }
            speck_gather_plan_destroy(plan);
            speck_comm_destroy(comm);
            speck_dcsr_free(&out[0]);
            speck_dcsr_free(&out[1]);
            cfgs[1].cleanup();
            config.cleanup();
            return errors ? 3 : 0;
        }
        if (compareResult) {
            // reference product and, for the value bound, |A| * |B| = sum |a*b| per entry (rocSPARSE both)
            dCSR<double> absA, absB;
            CSR<double> cpuAbs;
            convert(cpuAbs, cpuA, 0);
            for (size_t i = 0; i < cpuAbs.nnz; ++i) cpuAbs.data[i] = std::fabs(cpuAbs.data[i]);
            convert(absA, cpuAbs, 0);
            if (gpuA.rows != gpuA.cols)
                spECK::Transpose(absA, absB);
            else
                convert(absB, cpuAbs, 0);
            if (!rocsparse_reference(gpuA, gpuB, dCsrReference) || !rocsparse_reference(absA, absB, dCsrAbs)) {
                std::printf("Error: rocSPARSE reference failed\n");
                return 2;
            }
            // golden vectors for the CPU-side oracle tests (tests/golden/make_rocsparse_golden.py): the rocSPARSE
            // product as a .hicsr file
            if (const char* dump = std::getenv("SPECK_DUMP_ROCSPARSE")) {
                CSR<double> cpuRef;
                convert(cpuRef, dCsrReference, 0);
                storeCSR(cpuRef, dump);
            }
        }
        if (time_library) {
            // --time-library: the rocSPARSE product of the same inputs, same protocol, INSTEAD of the library's own
            size_t nnz_lib = 0;
            const double ms = rocsparse_timed(gpuA, gpuB, iterationsWarmup, iterationsExecution, &nnz_lib);
            if (ms < 0) {
                std::printf("Error: rocSPARSE SpGEMM failed\n");
                return 2;
            }
            std::cout << std::setw(20) << "rocSPARSE -> NNZ: " << nnz_lib << std::endl;
            std::cout << std::setw(20) << "rocSPARSE SpGEMM: " << ms << " ms" << std::endl;
            config.cleanup();
            return 0;
        }
        Timings timings, warmupTimings, benchTimings;
        int errors = 0;
        auto one = [&](Timings& acc) {
            timings = Timings();
            timings.measureAll = measureAll;
            timings.measureCompleteTime = measureCompleteTimes;
            spECK::MultiplyspECK<double, 4, 1024, spECK_DYNAMIC_MEM_PER_BLOCK, spECK_STATIC_MEM_PER_BLOCK>(
                gpuA, gpuB, dCsrHiRes, config, timings);
            acc += timings;
            if (compareResult && dCsrHiRes.data != nullptr && dCsrHiRes.col_ids != nullptr) {
                // structure bit-exact AND values within 1e-12 * sum|a*b| per entry: either one fails the run
                speck_dcsr a = dCsrReference.raw(), b = dCsrHiRes.raw(), sc = dCsrAbs.raw();
                uint64_t bad = 1, badv = 1;
                speck_compare_bounded_f64(nullptr, &a, &b, &sc, 1e-12, &bad, &badv);
                if (bad != 0) {
                    std::printf("Error: Matrix incorrect\n");
                    ++errors;
                } else if (badv != 0) {
                    std::printf("Error: %llu rows differ from rocSPARSE by more than 1e-12 * sum|a*b|\n",
                                (unsigned long long)badv);
                    ++errors;
                }
            }
        };
        for (int i = 0; i < iterationsWarmup; ++i) one(warmupTimings);
        for (int i = 0; i < iterationsExecution; ++i) one(benchTimings);
        benchTimings /= (float)iterationsExecution;

        std::cout << std::setw(20) << "var-SpGEMM -> NNZ: " << dCsrHiRes.nnz << std::endl;
        std::cout << std::setw(20) << "var-SpGEMM SpGEMM: " << benchTimings.complete << " ms" << std::endl;
        speck_stats st;
        speck_last_stats(config.handle, &st);
        if (benchTimings.complete > 0)
            std::cout << std::setw(20) << "var-SpGEMM GFLOPS: " << 2.0 * (double)st.sum_products / (benchTimings.complete * 1e6)
                      << std::endl;
        if (compareResult) std::cout << "compare vs rocSPARSE: " << (errors ? "FAILED" : "ok") << std::endl;
        config.cleanup();
        return errors ? 3 : 0;
    } catch (std::exception& ex) {
        std::cout << ex.what() << std::endl;
        return 1;
    } catch (const char* msg) {
        std::cout << msg << std::endl;
        return 1;
    }
}
