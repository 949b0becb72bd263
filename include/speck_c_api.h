/*
 * speck_c_api.h -- C-ABI boundary of the MI355X-native SpGEMM backend
 * (libspeck_amd.so).  Plain pointers and sizes only; every pointer inside a
 * speck_dcsr is a DEVICE pointer (HIP), exactly as in the reference's dCSR<T>.
 *
 * Each entry point cites the reference interface it replaces
 * (paths relative to the upstream GPUPeople/spECK tree).
 */
#ifndef SPECK_C_API_H
#define SPECK_C_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (the reference printf()s and returns; include/common.h:19-33,
 *      source/GPU/Multiply.cu:57-97 -- a C ABI reports instead) ---- */
enum {
    SPECK_OK = 0,
    SPECK_ERR_INVALID = 1,        /* null pointers / inconsistent sizes */
    SPECK_ERR_DIM_LIMIT = 2,      /* rows(A) or cols(B) > 2^27, source/GPU/Multiply.cu:57-66 */
    SPECK_ERR_HIP = 3,            /* a HIP runtime call failed */
    SPECK_ERR_OOM = 4,            /* device allocation failed, source/GPU/Multiply.cu:594-599 */
    SPECK_ERR_NNZ_OVERFLOW = 5,   /* nnz(C) does not fit the u32 row_offsets of dCSR */
    SPECK_ERR_NO_DEVICE = 6,
    SPECK_ERR_IO = 7,
    SPECK_ERR_UNSORTED = 8,       /* a row of B is not strictly ascending / column >= cols: the reference's
                                   * undocumented precondition (its loader sorts, source/CSR.cpp:173-212) */
    SPECK_ERR_COMM = 9            /* RCCL / shared-memory transport failure (row-sharded multi-GPU exchange) */
};

/* ---- device CSR: field-for-field the reference's dCSR<T> / dCSRNoDealloc<T>
 *      (include/dCSR.h:9-35): rows, cols, nnz, data, row_offsets, col_ids.
 *      row_offsets has rows+1 entries; they may be ABSOLUTE offsets into
 *      col_ids/data of a larger matrix (a row-range view used for sharding). ---- */
typedef struct speck_dcsr {
    uint64_t rows, cols, nnz;
    void *data;             /* double* or float*  (device) */
    uint32_t *row_offsets;  /* device */
    uint32_t *col_ids;      /* device */
} speck_dcsr;

/* ---- per-stage timings: the reference's Timings (include/Timings.h:4-18),
 *      milliseconds, same field names/order ---- */
typedef struct speck_timings {
    int32_t measureAll;
    int32_t measureCompleteTime;
    float init, countProducts, loadBalanceCounting, globalMapsCounting, spGEMMCounting, allocC,
        loadBalanceNumeric, globalMapsNumeric, spGEMMNumeric, sorting, cleanup, complete;
} speck_timings;

/* ---- what the last multiply did (drives bench.py's roofline object) ---- */
#define SPECK_NUM_SYM_BINS 16
#define SPECK_NUM_NUM_BINS 16
typedef struct speck_stats {
    uint64_t sum_products;                       /* P, u64 (reference: u32, Multiply.cu:237) */
    uint64_t nnz_c;
    uint32_t max_row_ops;                        /* maxComputationsPerRow, Multiply.cu:252 */
    uint32_t max_row_nnz_c;                      /* maxElementsPerRow, Multiply.cu:615 */
    uint32_t sym_bin_rows[SPECK_NUM_SYM_BINS];   /* rows per symbolic kernel class */
    uint32_t num_bin_rows[SPECK_NUM_NUM_BINS];   /* rows per numeric kernel class */
    uint64_t num_bin_bytes[SPECK_NUM_NUM_BINS];  /* algorithmic bytes per numeric class */
    uint64_t sym_bin_bytes[SPECK_NUM_SYM_BINS];
    float num_bin_ms[SPECK_NUM_NUM_BINS];        /* HIP-event ms of each numeric kernel launch */
    float sym_bin_ms[SPECK_NUM_SYM_BINS];
    float analysis_ms, scan_ms;
    float sym_light_ms, num_light_ms;            /* merged launch of the 256-thread classes (big-LDS part) */
    float sym_tiny_ms, num_tiny_ms;              /* ... and of the small classes, launched right behind it */
    int32_t kernel_events_valid;                 /* 1 if *_ms were recorded for the last call */
    int32_t numeric_reruns;                      /* replayed sequences rejected by the device-side checks */
    int32_t graph_replays;                       /* multiplies served by a reuse sequence (cumulative; the field names are
                                                  *    those of rounds 2-4 -- no executable graph is involved any more) */
    int32_t graph_captures;                      /* reuse sequences planned (cumulative) */
    float sym_phase_ms, num_phase_ms;            /* fork-to-join span of the symbolic / numeric launches (pipeline stream) */
    int32_t replayed;                            /* 1: the last multiply was served by a reuse sequence */
    int32_t nf_direct;                           /* 1: that sequence wrote the numeric-first rows straight to C at the row
                                                  *    offsets of the previous identical call (verified) */
    int32_t pool_fallbacks;                      /* scratch-pool classes switched off because the pool did not fit */
    int32_t esc_fused;                           /* 1: that sequence finished the rows of the register classes (<= 64
                                                  *    products) in its symbolic phase, at those offsets */
    uint64_t scratch_pool_bytes;                 /* numeric-first / global-key-set pool currently allocated */
    int32_t pred_stages;                         /* that sequence's integer stages verified the previous identical call's
                                                  *    decisions instead of folding them again: bit 0 the row-offset scan +
                                                  *    numeric binning (one kernel), bit 1 the symbolic binning (inside the
                                                  *    analysis kernel), bit 2 the analysis itself (a verifier on its own
                                                  *    stream beside the sequence), bit 3 no scan kernel at all (every row's
                                                  *    nnz compared where it is produced), bit 4 no symbolic pass for the hash
                                                  *    / dense rows either (their numeric bodies verify the nnz themselves:
                                                  *    option num_verify) -- DESIGN.md 4.3 */
    int32_t eager_speculated;                    /* an EAGER call that ran analysis .. scan as one batch sized from the previous
                                                  * eager call on the config (one read-back instead of two; option
                                                  * eager_speculate): 1 = its device-side checks held, -1 = they did not and
                                                  * the two-read-back sequence re-ran, 0 = not attempted */
    int32_t one_walk;                            /* 1: the last multiply was a ONE-WALK complete call (walk.hip): the
                                                  * rows of the register classes were finished inside the kernel that places the
                                                  * rows -- no symbolic pass for them, no scan kernel; scan_ms is that kernel */
    int32_t walk_misses;                         /* one-walk calls the device-side checks declared void (cumulative; the
                                                  * two-phase call re-ran) */
    int32_t eager_through;                       /* 1: the last multiply was a complete two-phase call enqueued as ONE batch --
                                                  * the numeric launches queued behind the scan, into the buffers matOut already
                                                  * had, everything the host checks between the phases checked by the scan
                                                  * (option eager_through; eager_speculated is 1 as well); -1: attempted, the
                                                  * device-side checks did not hold, nothing of C was written, the call re-ran */
} speck_stats;

typedef struct speck_config speck_config; /* opaque; reference: spECKConfig, include/spECKConfig.h:8-53 */

/* spECKConfig::initialize(device) -- include/spECKConfig.h:15-32: queries the
 * device (CU count, LDS limits), creates 6 streams and 4 events, and (new) a
 * grow-only scratch arena reused across calls. */
int speck_config_create(int device, speck_config **out);
/* spECKConfig::cleanup() -- include/spECKConfig.h:34-43 */
int speck_config_destroy(speck_config *cfg);
/* spECKConfig::{sm,maxStaticSharedMemoryPerBlock,maxDynamicSharedMemoryPerBlock} */
int speck_config_info(const speck_config *cfg, int *sm, int *max_static_lds, int *max_dynamic_lds);
/* spECKConfig::{streams, completeStart, completeEnd, individualStart, individualEnd} (include/spECKConfig.h:12-13):
 * the 6 hipStream_t and 4 hipEvent_t the config created, as void*; they stay owned by the config. */
int speck_config_handles(const speck_config *cfg, void *streams6[6], void *events4[4]);
/* Run the pipeline on a caller-owned stream (e.g. torch's current stream); NULL restores streams[0]. */
int speck_config_set_stream(speck_config *cfg, void *hip_stream);
/* Tunables (thresholds the reference hard-codes in Multiply.cu:128-131,321-324); name -> value. */
int speck_config_set_option(speck_config *cfg, const char *name, int64_t value);
/* Record HIP events around every kernel of the next calls (fills speck_stats.*_ms); enable = 2: around the
 * phases only (analysis_ms, scan_ms, sym_phase_ms, num_phase_ms -- no event between the launches of a phase). */
int speck_config_profile_kernels(speck_config *cfg, int enable);
int speck_last_stats(const speck_config *cfg, speck_stats *out);

/* spECK::MultiplyspECK<double,...>(A, B, matOut, config, timings) --
 * include/Multiply.h:15-16, source/GPU/Multiply.cu:51-1128.
 * Ownership as in the reference: A,B caller-owned, read-only; C is
 * allocated by the callee and freed by the caller (speck_dcsr_free); if
 * C->rows == A->rows and C->row_offsets != NULL that buffer is reused; data/col_ids are
 * re-allocated only when C->nnz != nnz(C).  On error the C STRUCT and its allocations are left untouched
 * (no field rewritten, nothing freed or allocated).  The CONTENTS of the buffers are untouched too, with one
 * exception: a repeated call on the same buffers that runs the replayed launch sequence places finished rows
 * straight into C->col_ids / C->data (options nf_direct / esc_fused, both on by default) before the device-side
 * checks of that sequence can reject it; if the eager re-run that follows then fails as well (inputs changed in
 * place into something invalid, out of memory for a grown C), the call returns the error with the contents of
 * col_ids / data unspecified.  The same holds for a ONE-WALK complete call (option one_walk, OFF by default: a complete
 * call on a matOut that is already allocated finishes short rows straight into C->col_ids / C->data before the input
 * check of B and its own device-side checks have spoken; a miss re-runs the two-phase call, an invalid input returns its
 * error with the contents unspecified).  A complete call enqueued as one batch (option eager_through, on by default) keeps
 * the rule: its scan looks at the verdict of the input check and at its own checks before any numeric kernel starts, and a
 * voided batch writes nothing.  row_offsets is rewritten only by a call that completes. */
int speck_multiply_f64(speck_config *cfg, const speck_dcsr *A, const speck_dcsr *B, speck_dcsr *C,
                       speck_timings *timings);
/* the <float,...> instantiation, source/GPU/Multiply.cu:1130 */
int speck_multiply_f32(speck_config *cfg, const speck_dcsr *A, const speck_dcsr *B, speck_dcsr *C,
                       speck_timings *timings);

/* ---- staged entry points (each a prefix of the pipeline; used by the parity
 *      tests and by the row-shard partitioner) ---- */
/* readOperations -- include/common.cuh:321-459, launch source/GPU/Multiply.cu:239-252.
 * d_* are device arrays of A->rows u32 (any may be NULL); h_* are host scalars. */
int speck_analysis(speck_config *cfg, const speck_dcsr *A, const speck_dcsr *B, uint32_t *d_row_ops,
                   uint32_t *d_row_max_ops, uint32_t *d_row_col_min, uint32_t *d_row_col_max,
                   uint64_t *h_sum_products, uint32_t *h_max_row_ops);
/* analysis + binning + symbolic + scan (source/GPU/Multiply.cu:239-575):
 * d_row_offsets (A->rows+1 u32, device) receives C's row offsets; *h_nnz_c = nnz(C). */
int speck_symbolic(speck_config *cfg, const speck_dcsr *A, const speck_dcsr *B,
                   uint32_t *d_row_offsets, uint64_t *h_nnz_c);
/* Row-shard boundaries with equal sum of per-row products:
 * h_bounds[parts+1], h_bounds[0]=0, h_bounds[parts]=A->rows. */
int speck_partition_rows(speck_config *cfg, const speck_dcsr *A, const speck_dcsr *B, int parts,
                         uint64_t *h_bounds);

/* ---- dCSR memory helpers: dCSR<T>::alloc / reset / convert(),
 *      include/dCSR.h:19-47, source/dCSR.cpp:25-115 ---- */
int speck_dcsr_alloc(speck_dcsr *m, uint64_t rows, uint64_t cols, uint64_t nnz, int alloc_offsets,
                     size_t value_size);
int speck_dcsr_free(speck_dcsr *m);
int speck_dcsr_upload(speck_dcsr *dst, uint64_t rows, uint64_t cols, uint64_t nnz,
                      const uint32_t *h_row_offsets, const uint32_t *h_col_ids, const void *h_data,
                      size_t value_size);
int speck_dcsr_download(const speck_dcsr *src, uint32_t *h_row_offsets, uint32_t *h_col_ids,
                        void *h_data, size_t value_size);
/* convert(dCSR&, const CSR&, padding) -- source/dCSR.cpp:51-66: buffers for rows + padding rows and nnz + 8 * padding
 * entries, rows / nnz of the source, the padding zero-filled.  speck_dcsr_upload = padding 0. */
int speck_dcsr_upload_padded(speck_dcsr *dst, uint64_t rows, uint64_t cols, uint64_t nnz,
                             const uint32_t *h_row_offsets, const uint32_t *h_col_ids, const void *h_data,
                             size_t value_size, uint32_t padding);
/* convert(dCSR&, const dCSR&, padding) -- source/dCSR.cpp:81-89: device-to-device, no host round trip.  `src` may be
 * a row-range view with absolute offsets: the copy is rebased to start at 0.  dst must not share any buffer with src
 * (SPECK_ERR_INVALID: the allocation of dst frees what it held).  Like every speck_dcsr_* call it works on the NULL
 * stream and returns when the copy is complete; a caller that produced src on its own non-blocking stream synchronises
 * that stream first. */
int speck_dcsr_copy(speck_dcsr *dst, const speck_dcsr *src, size_t value_size, uint32_t padding);
/* overwrite the contents of an existing device matrix in place (same rows / nnz, same device
 * pointers); any of the host arrays may be NULL */
int speck_dcsr_update(speck_dcsr *dst, const uint32_t *h_row_offsets, const uint32_t *h_col_ids,
                      const void *h_data, size_t value_size);
/* spECK::Compare(ref, cmp, compare_data) -- include/Compare.h:5-6, source/GPU/Compare.cu:11-82;
 * stricter: offsets + col ids bit-exact; values |x-y| <= rel_tol*max(|x|,|y|) when compare_data.
 * *h_mismatches = number of differing rows (0 = equal). */
int speck_compare_f64(speck_config *cfg, const speck_dcsr *ref, const speck_dcsr *cmp,
                      int compare_data, double rel_tol, uint64_t *h_mismatches);
/* ... and its float instantiation (source/GPU/Compare.cu:84) */
int speck_compare_f32(speck_config *cfg, const speck_dcsr *ref, const speck_dcsr *cmp,
                      int compare_data, double rel_tol, uint64_t *h_mismatches);
/* The value check a SpGEMM result admits whatever its summation order: |ref - cmp| <= tol * S per
 * entry, S = sum |a*b| of the entry, handed over as `abs_products` = |A|*|B| (same pattern as ref).
 * Role of the reference's compare against cuSPARSE (source/Executor.cpp:29-40), made to FAIL on
 * values: *h_structure_rows / *h_value_rows = rows that differ in pattern / beyond the bound. */
int speck_compare_bounded_f64(speck_config *cfg, const speck_dcsr *ref, const speck_dcsr *cmp,
                              const speck_dcsr *abs_products, double tol, uint64_t *h_structure_rows,
                              uint64_t *h_value_rows);
/* Order-preserving transpose (source/GPU/Transpose.cu:10-117; DataLoader.cpp:65-69 for rows!=cols). */
int speck_transpose_f64(speck_config *cfg, const speck_dcsr *A, speck_dcsr *At);
/* ... and its float instantiation (source/GPU/Transpose.cu:116) */
int speck_transpose_f32(speck_config *cfg, const speck_dcsr *A, speck_dcsr *At);

/* ---- row-sharded multi-GPU (new: the reference is single-GPU, source/Executor.cpp:25).  One process per GPU;
 *      rank p multiplies the row range [b_p, b_{p+1}) of A (a view with absolute offsets, boundaries from
 *      speck_partition_rows) with a replicated B, then ONE exchange concatenates the shards on a root rank:
 *      ncclAllGather of the shard sizes + a gatherv built from grouped ncclSend / ncclRecv over xGMI (RCCL has no
 *      gatherv) + a rebase of the received row offsets.  librccl is resolved with dlopen at the first call. ---- */
enum {
    SPECK_TRANSPORT_RCCL = 0,     /* device-to-device over xGMI */
    SPECK_TRANSPORT_HOSTMEM = 1   /* staged through POSIX shared memory: ranks that cannot form an RCCL
                                   * communicator (several ranks on one GPU -- plumbing checks) */
};
typedef struct speck_comm speck_comm;
typedef struct speck_gather_plan speck_gather_plan;
/* ncclGetUniqueId: 128 bytes made by ONE rank and handed to the others by the launcher (file, pipe, MPI, ...) */
int speck_comm_unique_id(int transport, void *id128);
/* ncclCommInitRank on `device`; collective over all ranks of the job */
int speck_comm_init(int device, int nranks, int rank, int transport, const void *id128, speck_comm **out);
int speck_comm_destroy(speck_comm *comm);
int speck_comm_info(const speck_comm *comm, int *nranks, int *rank, int *transport);
/* One-shot gatherv (collective): every rank passes its shard of C (local row offsets, as speck_multiply returns
 * it for a row-range view of A); on `root`, *full receives the concatenated matrix (callee allocates, caller
 * frees with speck_dcsr_free); other ranks may pass NULL. */
int speck_gatherv_csr(speck_comm *comm, int root, const speck_dcsr *shard, uint64_t cols, size_t value_size,
                      speck_dcsr *full);
/* Repeated exchange of shards whose sizes do not change (the benchmark loop): sizes are exchanged and the
 * root's buffers allocated ONCE (collective; create plans in the same order on every rank), start() posts the
 * transfers of a slot on the communicator's own stream and returns -- they overlap the next multiply -- and
 * wait() blocks on the slot's event.  A slot's shard must stay untouched between start() and wait() (alternate
 * two output matrices).  On the root, wait() fills *full_view with a VIEW of the slot's buffers (owned by the
 * plan, valid until the slot is started again). */
int speck_gather_plan_create(speck_comm *comm, int root, uint64_t rows_local, uint64_t cols, uint64_t nnz_local,
                             size_t value_size, int slots, speck_gather_plan **out);
int speck_gather_start(speck_gather_plan *plan, int slot, const speck_dcsr *shard);
int speck_gather_wait(speck_gather_plan *plan, int slot, speck_dcsr *full_view);
/* displacements of every rank's rows / entries in the concatenation (nranks + 1 entries each; either may be NULL) */
int speck_gather_plan_layout(const speck_gather_plan *plan, uint64_t *row_displs, uint64_t *nnz_displs);
int speck_gather_plan_destroy(speck_gather_plan *plan);

/* ---- host-side synthetic inputs and on-disk formats ---- */
typedef struct speck_host_csr speck_host_csr; /* opaque host CSR<double>, include/CSR.h */
/* kind: "uniform" (config #1), "scircuit", "webbase", "mac_econ", "cant", "nlpkkt";
 * scale multiplies the row count (1.0 = the SuiteSparse dimensions). */
int speck_gen_matrix(const char *kind, double scale, uint64_t seed, int signed_values,
                     speck_host_csr **out);
/* loadMTX + convert(COO->CSR) -- source/COO.cpp:53-164, source/CSR.cpp:173-212 */
int speck_load_mtx(const char *path, speck_host_csr **out);
/* MatrixMarket writer (no reference counterpart): coordinate real, general or -- symmetric_lower != 0 -- the
 * lower triangle as `symmetric` (the caller vouches for the symmetry) */
int speck_store_mtx(const speck_host_csr *m, const char *path, int symmetric_lower);
/* loadCSR / storeCSR (.hicsr) -- source/CSR.cpp:88-137 */
int speck_load_hicsr(const char *path, speck_host_csr **out);
int speck_store_hicsr(const speck_host_csr *m, const char *path);
/* DataLoader: "<path>d_.hicsr" cache, else .mtx then write cache -- source/DataLoader.cpp:24-58 */
int speck_load_matrix(const char *path, int write_cache, speck_host_csr **out);
int speck_host_csr_dims(const speck_host_csr *m, uint64_t *rows, uint64_t *cols, uint64_t *nnz);
int speck_host_csr_copy(const speck_host_csr *m, uint32_t *row_offsets, uint32_t *col_ids, double *data);
int speck_host_csr_from_arrays(uint64_t rows, uint64_t cols, uint64_t nnz, const uint32_t *row_offsets,
                               const uint32_t *col_ids, const double *data, speck_host_csr **out);
int speck_host_csr_free(speck_host_csr *m);

const char *speck_status_string(int status);
const char *speck_version(void);

#ifdef __cplusplus
}
#endif
#endif
