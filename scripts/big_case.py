import sys, time, numpy as np, torch
import speck_amd as sa
s = float(sys.argv[1])
t=time.time(); A = sa.gen_matrix("nlpkkt", s, 1); print("gen", A.rows, A.nnz, round(time.time()-t,1), flush=True)
dA = sa.dCSR.from_host(A); print("uploaded", flush=True)
cfg = sa.spECKConfig.initialize(0)
t=time.time(); an = sa.analysis(dA, dA, cfg); print("analysis", an["sum_products"], round(time.time()-t,2), flush=True)
t=time.time(); ro, nnz = sa.symbolic(dA, dA, cfg); print("symbolic nnzC", nnz, round(time.time()-t,2), flush=True)
dC = sa.dCSR(np.float64)
for i in range(3):
    t=time.time(); sa.MultiplyspECK(dA, dA, dC, cfg); print("multiply", i, round((time.time()-t)*1e3,2), "ms", dC.nnz, flush=True)
print(cfg.last_stats()["num_bin_rows"])
# size-independent checks + a sampled comparison with scipy (no oracle outside tests/)
import scipy.sparse as sp
got = dC.to_host()
ro = got.row_offsets.astype(np.int64)
assert ro[-1] == got.nnz and (np.diff(ro) >= 0).all()
d = np.diff(got.col_ids.astype(np.int64))
starts = ro[1:-1][ro[1:-1] < got.nnz]
inc = np.ones(got.nnz - 1, dtype=bool); inc[starts - 1] = False   # row boundaries may decrease
assert (d[inc] > 0).all(), "columns not strictly ascending inside a row"
S = sp.csr_matrix((A.data, A.col_ids.astype(np.int64), A.row_offsets.astype(np.int64)), shape=(A.rows, A.cols))
for lo in (0, A.rows // 2, A.rows - 2000):
    R = (S[lo:lo + 2000] @ S).tocsr(); R.sort_indices()
    a, b = ro[lo], ro[lo + 2000]
    assert (np.diff(ro[lo:lo + 2001]) == np.diff(R.indptr)).all(), "row nnz differs"
    assert (got.col_ids[a:b] == R.indices).all(), "col ids differ"
    assert np.allclose(got.data[a:b], R.data, rtol=1e-10, atol=1e-12), "values differ"
print("properties + sampled rows ok")
