#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONPATH=$PWD
python - <<'PY'
import sys
sys.argv=["x","nlpkkt"]
import numpy as np, torch
import speck_amd as sa
cfg = sa.spECKConfig.initialize(0)
for k, v in (("use_graph", 0), ("merge_light", 0), ("concurrent_classes", 0)):
    cfg.set_option(k, v)
cfg.profile_kernels(1)
A = sa.gen_matrix("nlpkkt", 0.2, 1)
dA = sa.dCSR.from_host(A); dC = sa.dCSR(np.float64)
for i in range(3):
    sa.MultiplyspECK(dA, dA, dC, cfg)
    st = cfg.last_stats()
print("rows", A.rows, "P", st["sum_products"], "nnzC", st["nnz_c"], "analysis", st["analysis_ms"], "scan", st["scan_ms"])
for ph in ("sym", "num"):
    for c, ms in st[ph + "_bin_ms"].items():
        rows = st[ph + "_bin_rows"][c]
        if rows: print(f"  {ph}:{c:14s} rows {rows:8d}  {ms*1e3:9.1f} us  {ms*1e6/rows:7.2f} ns/row")
t = sa.Timings(measureCompleteTime=True)
cfg.profile_kernels(0)
for k, v in (("use_graph", 1), ("merge_light", 1), ("concurrent_classes", 1)):
    cfg.set_option(k, v)
for i in range(5):
    sa.MultiplyspECK(dA, dA, dC, cfg, t)
print("complete ms", t.complete)
PY
