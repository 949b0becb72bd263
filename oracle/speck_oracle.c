/*
 * oracle/speck_oracle.c -- CPU restatement of the spECK SpGEMM contract.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED BY THE REFERENCE (see speck_oracle.h).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off keeps "multiply, round, then add" as the reference's
 * numeric kernels do (include/GPU/spECK_HashSpGEMM.cuh:157-165: the product is
 * formed first, then handed to atomicAdd).
 */
#include "speck_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

uint64_t orc_splitmix64_next(uint64_t *state)
{
    uint64_t z = (*state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static int cmp_u32(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}

/* ascending sort of distinct-or-not u32 keys: insertion sort for short rows,
 * LSD byte radix (only over the bytes that vary) otherwise. tmp holds n keys. */
static void sort_u32(uint32_t *a, uint32_t n, uint32_t *tmp)
{
    if (n <= 24) {
        for (uint32_t i = 1; i < n; ++i) {
            uint32_t x = a[i];
            uint32_t j = i;
            while (j > 0 && a[j - 1] > x) {
                a[j] = a[j - 1];
                --j;
            }
            a[j] = x;
        }
        return;
    }
    uint32_t all_or = 0, all_and = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < n; ++i) {
        all_or |= a[i];
        all_and &= a[i];
    }
    uint32_t vary = all_or ^ all_and;
    uint32_t *src = a, *dst = tmp;
    for (int shift = 0; shift < 32; shift += 8) {
        if (!((vary >> shift) & 0xFFu))
            continue;
        uint32_t cnt[257];
        memset(cnt, 0, sizeof(cnt));
        for (uint32_t i = 0; i < n; ++i)
            ++cnt[((src[i] >> shift) & 0xFFu) + 1];
        for (int b = 0; b < 256; ++b)
            cnt[b + 1] += cnt[b];
        for (uint32_t i = 0; i < n; ++i)
            dst[cnt[(src[i] >> shift) & 0xFFu]++] = src[i];
        uint32_t *t = src;
        src = dst;
        dst = t;
    }
    if (src != a)
        memcpy(a, src, (size_t)n * sizeof(uint32_t));
}

uint64_t orc_gen_uniform(uint32_t n, uint64_t seed, uint32_t kmin, uint32_t kspan,
                         int signed_values, uint32_t *row_offsets, uint32_t *col_ids,
                         double *data)
{
    uint64_t st = seed;
    uint64_t nnz = 0;
    uint8_t *used = (uint8_t *)calloc(n, 1);
    for (uint32_t r = 0; r < n; ++r) {
        row_offsets[r] = (uint32_t)nnz;
        uint32_t k = kmin + (uint32_t)(orc_splitmix64_next(&st) % kspan);
        if (k > n)
            k = n;
        uint32_t got = 0;
        while (got < k) {
            uint32_t c = (uint32_t)(orc_splitmix64_next(&st) % n);
            if (!used[c]) {
                used[c] = 1;
                col_ids[nnz + got++] = c;
            }
        }
        qsort(col_ids + nnz, k, sizeof(uint32_t), cmp_u32);
        for (uint32_t j = 0; j < k; ++j) {
            used[col_ids[nnz + j]] = 0;
            double v = 0.5 + (double)(orc_splitmix64_next(&st) >> 11) * 0x1.0p-53;
            if (signed_values && (orc_splitmix64_next(&st) & 1))
                v = -v;
            data[nnz + j] = v;
        }
        nnz += k;
    }
    row_offsets[n] = (uint32_t)nnz;
    free(used);
    return nnz;
}

void orc_analysis(uint64_t a_rows, const uint32_t *a_row_offsets, const uint32_t *a_col_ids,
                  const uint32_t *b_row_offsets, const uint32_t *b_col_ids,
                  uint32_t *row_ops, uint32_t *row_max_ops, uint32_t *row_col_min,
                  uint32_t *row_col_max, uint64_t *sum_products, uint32_t *max_row_ops)
{
    uint64_t total = 0;
    uint32_t gmax = 0;
    for (uint64_t i = 0; i < a_rows; ++i) {
        uint64_t ops = 0;
        uint32_t mx = 0, cmin = 0xFFFFFFFFu, cmax = 0;
        for (uint32_t ia = a_row_offsets[i]; ia < a_row_offsets[i + 1]; ++ia) {
            uint32_t k = a_col_ids[ia];
            uint32_t bs = b_row_offsets[k], be = b_row_offsets[k + 1];
            uint32_t len = be - bs;
            ops += len;
            if (len > mx)
                mx = len;
            if (len) {
                if (b_col_ids[bs] < cmin)
                    cmin = b_col_ids[bs];
                if (b_col_ids[be - 1] > cmax)
                    cmax = b_col_ids[be - 1];
            }
        }
        uint32_t ops32 = ops > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)ops;
        if (row_ops)
            row_ops[i] = ops32;
        if (row_max_ops)
            row_max_ops[i] = mx;
        if (row_col_min)
            row_col_min[i] = cmin;
        if (row_col_max)
            row_col_max[i] = cmax;
        total += ops;
        if (ops32 > gmax)
            gmax = ops32;
    }
    if (sum_products)
        *sum_products = total;
    if (max_row_ops)
        *max_row_ops = gmax;
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static int pick_threads(int threads)
{
    int mx = orc_max_threads();
    if (threads <= 0 || threads > mx)
        return mx;
    return threads;
}

/* Per-thread workspace kept across calls (OpenMP reuses its threads): a call that timed the
 * allocation and zeroing of cols-sized arrays per thread would measure the allocator, not the
 * multiply, once there are dozens of threads.  stamp[] never needs clearing: every call uses a fresh
 * range of stamp values (epoch). */
typedef struct {
    uint64_t *stamp;
    double *acc, *aacc; /* also used as float arrays */
    uint32_t *stmp;
    uint64_t cols, epoch;
} orc_workspace;
static _Thread_local orc_workspace g_ws;

static orc_workspace *workspace(uint64_t cols, uint64_t rows, uint64_t *epoch)
{
    orc_workspace *w = &g_ws;
    if (cols == 0)
        cols = 1;
    if (w->cols < cols) {
        free(w->stamp);
        free(w->acc);
        free(w->aacc);
        free(w->stmp);
        w->stamp = (uint64_t *)calloc(cols, sizeof(uint64_t));
        w->acc = (double *)malloc(cols * sizeof(double));
        w->aacc = (double *)malloc(cols * sizeof(double));
        w->stmp = (uint32_t *)malloc(cols * sizeof(uint32_t));
        w->cols = cols;
        w->epoch = 0;
    }
    *epoch = w->epoch;
    w->epoch += rows + 1;
    return w;
}

uint64_t orc_symbolic(uint64_t a_rows, uint64_t b_cols, const uint32_t *a_row_offsets,
                      const uint32_t *a_col_ids, const uint32_t *b_row_offsets,
                      const uint32_t *b_col_ids, uint32_t *row_nnz, int threads)
{
    uint64_t total = 0;
    int nt = pick_threads(threads);
#pragma omp parallel num_threads(nt) reduction(+ : total)
    {
        /* stamp[c] == epoch+i+1  <=>  column c already seen in row i */
        uint64_t epoch;
        uint64_t *stamp = workspace(b_cols, a_rows, &epoch)->stamp;
#pragma omp for schedule(dynamic, 256)
        for (uint64_t i = 0; i < a_rows; ++i) {
            uint32_t cnt = 0;
            const uint64_t mark = epoch + i + 1;
            for (uint32_t ia = a_row_offsets[i]; ia < a_row_offsets[i + 1]; ++ia) {
                uint32_t k = a_col_ids[ia];
                for (uint32_t ib = b_row_offsets[k]; ib < b_row_offsets[k + 1]; ++ib) {
                    uint32_t c = b_col_ids[ib];
                    if (stamp[c] != mark) {
                        stamp[c] = mark;
                        ++cnt;
                    }
                }
            }
            row_nnz[i] = cnt;
            total += cnt;
        }
    }
    return total;
}

uint64_t orc_exclusive_scan(uint32_t *counts, uint64_t n)
{
    uint64_t run = 0;
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t c = counts[i];
        counts[i] = (uint32_t)run;
        run += c;
    }
    counts[n] = (uint32_t)run;
    return run;
}

#define ORC_NUMERIC_BODY(T, ABSF)                                                              \
    int nt = pick_threads(threads);                                                            \
    _Pragma("omp parallel num_threads(nt)")                                                    \
    {                                                                                          \
        uint64_t epoch;                                                                        \
        orc_workspace *ws = workspace(b_cols, a_rows, &epoch);                                 \
        uint64_t *stamp = ws->stamp;                                                           \
        T *acc = (T *)ws->acc;                                                                 \
        T *aacc = (T *)ws->aacc;                                                               \
        uint32_t *stmp = ws->stmp;                                                             \
        _Pragma("omp for schedule(dynamic, 256)")                                              \
        for (uint64_t i = 0; i < a_rows; ++i) {                                                \
            uint32_t base = c_row_offsets[i];                                                  \
            uint32_t cnt = 0;                                                                  \
            const uint64_t mark = epoch + i + 1;                                               \
            for (uint32_t ia = a_row_offsets[i]; ia < a_row_offsets[i + 1]; ++ia) {            \
                uint32_t k = a_col_ids[ia];                                                    \
                T av = a_data[ia];                                                             \
                for (uint32_t ib = b_row_offsets[k]; ib < b_row_offsets[k + 1]; ++ib) {        \
                    uint32_t c = b_col_ids[ib];                                                \
                    T p = av * b_data[ib];                                                     \
                    if (stamp[c] != mark) {                                                    \
                        stamp[c] = mark;                                                       \
                        c_col_ids[base + cnt++] = c;                                           \
                        acc[c] = p;                                                            \
                        aacc[c] = ABSF(p);                                                     \
                    } else {                                                                   \
                        acc[c] = acc[c] + p;                                                   \
                        aacc[c] = aacc[c] + ABSF(p);                                           \
                    }                                                                          \
                }                                                                              \
            }                                                                                  \
            sort_u32(c_col_ids + base, cnt, stmp);                                             \
            for (uint32_t j = 0; j < cnt; ++j) {                                               \
                uint32_t c = c_col_ids[base + j];                                              \
                c_data[base + j] = acc[c];                                                     \
                if (c_abs)                                                                     \
                    c_abs[base + j] = aacc[c];                                                 \
            }                                                                                  \
        }                                                                                      \
    }

void orc_numeric(uint64_t a_rows, uint64_t b_cols, const uint32_t *a_row_offsets,
                 const uint32_t *a_col_ids, const double *a_data,
                 const uint32_t *b_row_offsets, const uint32_t *b_col_ids, const double *b_data,
                 const uint32_t *c_row_offsets, uint32_t *c_col_ids, double *c_data,
                 double *c_abs, int threads)
{
    ORC_NUMERIC_BODY(double, fabs)
}

void orc_numeric_f32(uint64_t a_rows, uint64_t b_cols, const uint32_t *a_row_offsets,
                     const uint32_t *a_col_ids, const float *a_data,
                     const uint32_t *b_row_offsets, const uint32_t *b_col_ids,
                     const float *b_data, const uint32_t *c_row_offsets, uint32_t *c_col_ids,
                     float *c_data, float *c_abs, int threads)
{
    ORC_NUMERIC_BODY(float, fabsf)
}

void orc_transpose(uint64_t rows, uint64_t cols, const uint32_t *row_offsets,
                   const uint32_t *col_ids, const double *data, uint32_t *t_row_offsets,
                   uint32_t *t_col_ids, double *t_data)
{
    memset(t_row_offsets, 0, (cols + 1) * sizeof(uint32_t));
    uint32_t nnz = row_offsets[rows];
    for (uint32_t j = 0; j < nnz; ++j)
        ++t_row_offsets[col_ids[j] + 1];
    for (uint64_t c = 0; c < cols; ++c)
        t_row_offsets[c + 1] += t_row_offsets[c];
    uint32_t *cursor = (uint32_t *)malloc((cols ? cols : 1) * sizeof(uint32_t));
    memcpy(cursor, t_row_offsets, cols * sizeof(uint32_t));
    for (uint64_t r = 0; r < rows; ++r)
        for (uint32_t j = row_offsets[r]; j < row_offsets[r + 1]; ++j) {
            uint32_t dst = cursor[col_ids[j]]++;
            t_col_ids[dst] = (uint32_t)r;
            t_data[dst] = data[j];
        }
    free(cursor);
}
