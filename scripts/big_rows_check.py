"""Replay of problems with very many rows (scan tiles of 8 and 32 rows per thread, thousands of analysis blocks):
eager, capture + replay, replay -- each compared with the oracle.  usage (GPU box): python scripts/big_rows_check.py"""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import torch  # noqa
import speck_amd as sa
from oracle import pyoracle as po

def make(m, n, per, seed):
    rng = np.random.default_rng(seed)
    ln = rng.integers(0, per + 1, size=m)
    ro = np.zeros(m + 1, dtype=np.int64); ro[1:] = np.cumsum(ln)
    # ascending columns per row: a random start plus strictly increasing steps
    step = rng.integers(1, max(2, n // (4 * per)), size=int(ro[-1]))
    start = rng.integers(0, n // 2, size=m)
    col = np.empty(int(ro[-1]), dtype=np.int64)
    idx = np.repeat(np.arange(m), ln)
    csum = np.cumsum(step); first = ro[:-1][idx]
    base = np.where(first > 0, csum[np.maximum(first - 1, 0)], 0)
    col = start[idx] + (csum - base)
    col = np.minimum(col, n - 1)
    # enforce strict ascent after the clamp
    bad = np.zeros(len(col), dtype=bool); bad[1:] = (col[1:] <= col[:-1]) & (idx[1:] == idx[:-1])
    keep = ~bad
    cnt = np.bincount(idx[keep], minlength=m)
    ro2 = np.zeros(m + 1, dtype=np.int64); ro2[1:] = np.cumsum(cnt)
    val = rng.random(int(keep.sum())) + 0.5
    return po.HostCSR(m, n, ro2.astype(np.uint32), col[keep].astype(np.uint32), val)

cfg = sa.spECKConfig.initialize(0)
bad = 0
for m, per in ((700_000, 4), (9_000_000, 2)):
    A = make(m, m, per, m)
    R, ab = po.spgemm(A, A)
    dA, dC = sa.dCSR.from_host(sa.HostCSR(A.rows, A.cols, A.row_offsets, A.col_ids, A.data)), sa.dCSR()
    for rep in range(4):
        sa.MultiplyspECK(dA, dA, dC, cfg)
        got = dC.to_host(); st = cfg.last_stats()
        ok = got.nnz == R.nnz and (got.row_offsets == R.row_offsets).all() and (got.col_ids == R.col_ids).all() and \
            bool((np.abs(got.data - R.data) <= 1e-12 * ab + 1e-300).all())
        bad += 0 if ok else 1
        print(m, "rep", rep, "ok" if ok else "BAD", "replayed", st["replayed"], "pred", st["pred_stages"], "fused", st["esc_fused"], "nnzC", R.nnz, flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
