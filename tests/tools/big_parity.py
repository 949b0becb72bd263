"""Full comparison against the oracle at sizes beyond the test suite (hand-run on a GPU box).
usage: python tests/tools/big_parity.py workload scale"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
import speck_amd as sa
from oracle import pyoracle as po
wl, scale = sys.argv[1], float(sys.argv[2])
A = sa.gen_matrix(wl, scale, 3, signed=True)
H = po.HostCSR(A.rows, A.cols, A.row_offsets, A.col_ids, A.data)
t = time.time(); R, ab = po.spgemm(H, H, threads=16); print("oracle", round(time.time() - t, 1), "s", R.nnz, flush=True)
cfg = sa.spECKConfig.initialize(0)
dA = sa.dCSR.from_host(A); dC = sa.dCSR(np.float64)
for i in range(3):
    sa.MultiplyspECK(dA, dA, dC, cfg)
got = dC.to_host()
assert got.nnz == R.nnz and (got.row_offsets == R.row_offsets).all(), "row_offsets differ"
assert (got.col_ids == R.col_ids).all(), "col_ids differ"
err = np.abs(got.data - R.data)
assert (err <= 1e-12 * ab + 1e-300).all(), "values differ"
print(wl, scale, "rows", A.rows, "nnzC", R.nnz, "bit-exact indices, values within 1e-12*sum|ab|")
print({k: v for k, v in cfg.last_stats()["num_bin_rows"].items() if v})
