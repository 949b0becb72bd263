/*
 * oracle/speck_oracle.h -- CPU restatement of the spECK SpGEMM contract.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 * The product (speck_amd/libspeck_amd.so) never links or calls it.
 *
 * PARITY UNPINNED BY THE REFERENCE: /root/reference ships no tests, no golden
 * vectors and no CPU SpGEMM.  The reference's
 * hot path is CUDA-only and cannot be built in this image.  This oracle
 * therefore restates the *contract* of the path, taken from the reference
 * call sites cited per function below, and is pinned by
 *   (i)  the known-answer figures for the config-#1 generator in
 *        SURVEY.md section 8d (tests/golden/synth10k.json),
 *   (ii) scipy.sparse on cancellation-free inputs (tests/test_oracle.py),
 *   (iii) the reference's own host-side C++ (CSR.cpp / COO.cpp), compiled
 *        into oracle/_ref/ for the on-disk formats,
 *   (iv) golden vectors of a third-party SpGEMM: rocSPARSE's products of six
 *        small stand-in inputs (tests/golden/rocsparse/, written on an MI355X
 *        by tests/golden/make_rocsparse_golden.py; the stand-in for the
 *        reference's cuSPARSE compare path, source/Executor.cpp:29-40).
 */
#ifndef SPECK_ORACLE_H
#define SPECK_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* splitmix64, SURVEY.md 8d. */
uint64_t orc_splitmix64_next(uint64_t *state);

/*
 * Config-#1 generator: n x n, per row k = kmin + next()%kspan
 * distinct columns (rejection), ascending, value 0.5 + (next()>>11)*2^-53,
 * optional sign flip.  row_offsets has n+1 entries; col_ids/data must hold
 * n*(kmin+kspan-1) entries.  Returns nnz.
 */
uint64_t orc_gen_uniform(uint32_t n, uint64_t seed, uint32_t kmin, uint32_t kspan,
                         int signed_values, uint32_t *row_offsets, uint32_t *col_ids,
                         double *data);

/*
 * Lightweight analysis quantities, defined as in the reference's
 * readOperations kernel (include/common.cuh:321-459):
 *   row_ops[i]     = sum_k nnz(B_k)       over the non-zeros a_ik   (:390-402)
 *   row_max_ops[i] = max_k nnz(B_k)                                (:403-404)
 *   row_col_min[i] / row_col_max[i] = min first col / max last col of the
 *                    referenced non-empty B rows (:395-400); for a row with no
 *                    products: min = 0xFFFFFFFF, max = 0 (:344-345)
 *   *sum_products  = sum_i row_ops[i]  (u64 here; the reference's u32
 *                    wraps on nlpkkt160)
 *   *max_row_ops   = max_i row_ops[i]
 * A may be a row-slice view: a_row_offsets holds absolute offsets into a_col_ids.
 */
void orc_analysis(uint64_t a_rows, const uint32_t *a_row_offsets, const uint32_t *a_col_ids,
                  const uint32_t *b_row_offsets, const uint32_t *b_col_ids,
                  uint32_t *row_ops, uint32_t *row_max_ops, uint32_t *row_col_min,
                  uint32_t *row_col_max, uint64_t *sum_products, uint32_t *max_row_ops);

/*
 * Symbolic pass: row_nnz[i] = number of DISTINCT columns reached by row i.
 * Structural count -- an entry whose products cancel to 0.0 is still counted
 * (reference: include/HashMap.cuh:167-195 counts distinct keys; numeric never
 * prunes).  Returns nnz(C) as u64.
 */
uint64_t orc_symbolic(uint64_t a_rows, uint64_t b_cols, const uint32_t *a_row_offsets,
                      const uint32_t *a_col_ids, const uint32_t *b_row_offsets,
                      const uint32_t *b_col_ids, uint32_t *row_nnz, int threads);

/*
 * Numeric pass (Gustavson, dense accumulator).  c_row_offsets (a_rows+1) must
 * be the exclusive scan of orc_symbolic's row_nnz.  Writes ascending column ids
 * per row (reference: include/GPU/spECK_HashSpGEMM.cuh:716-728, 1278-1289,
 * 1909-1923), values = sum of individually rounded products a*b
 * (:157-165 -- multiply, then accumulate; no FMA), in A-row order then B-row
 * order.  If c_abs != NULL it receives sum |a*b| per entry, the scale against
 * which any summation order differs by at most ~n*eps.
 */
void orc_numeric(uint64_t a_rows, uint64_t b_cols, const uint32_t *a_row_offsets,
                 const uint32_t *a_col_ids, const double *a_data,
                 const uint32_t *b_row_offsets, const uint32_t *b_col_ids, const double *b_data,
                 const uint32_t *c_row_offsets, uint32_t *c_col_ids, double *c_data,
                 double *c_abs, int threads);

/* float32 instantiation of the same contract (reference exports <float,...>,
 * source/GPU/Multiply.cu:1130). Products and sums are rounded to float. */
void orc_numeric_f32(uint64_t a_rows, uint64_t b_cols, const uint32_t *a_row_offsets,
                     const uint32_t *a_col_ids, const float *a_data,
                     const uint32_t *b_row_offsets, const uint32_t *b_col_ids,
                     const float *b_data, const uint32_t *c_row_offsets, uint32_t *c_col_ids,
                     float *c_data, float *c_abs, int threads);

/* In-place exclusive scan of n+1 u32 (last input ignored), returns total as u64
 * (reference: cub::DeviceScan::ExclusiveSum, source/GPU/Multiply.cu:570). */
uint64_t orc_exclusive_scan(uint32_t *counts, uint64_t n);

/* Order-preserving counting transpose (reference: source/GPU/Transpose.cu:10-117
 * / cuSPARSE csr2csc used by source/DataLoader.cpp:65-69). */
void orc_transpose(uint64_t rows, uint64_t cols, const uint32_t *row_offsets,
                   const uint32_t *col_ids, const double *data, uint32_t *t_row_offsets,
                   uint32_t *t_col_ids, double *t_data);

int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
