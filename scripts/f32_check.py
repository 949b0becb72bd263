"""fp32 vs fp64 multiply time on the same pattern (the reference instantiates both, Multiply.cu:1130)."""
import sys, time
import numpy as np, torch
import speck_amd as sa
wl = sys.argv[1] if len(sys.argv) > 1 else "scircuit"
A = sa.gen_matrix(wl, 1.0, 1)
for dt in (np.float64, np.float32):
    H = sa.HostCSR(A.rows, A.cols, A.row_offsets, A.col_ids, A.data.astype(dt))
    dA = sa.dCSR.from_host(H); dC = sa.dCSR(dt)
    cfg = sa.spECKConfig.initialize(0)
    for _ in range(8): sa.MultiplyspECK(dA, dA, dC, cfg)
    t = time.perf_counter()
    for _ in range(50): sa.MultiplyspECK(dA, dA, dC, cfg)
    print(wl, dt.__name__, round((time.perf_counter() - t) / 50 * 1e3, 4), "ms/step")
