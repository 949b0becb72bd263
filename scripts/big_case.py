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
