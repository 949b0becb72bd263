"""Isolated time and row count of every symbolic / numeric class launch (eager, one class at a time).
usage: python scripts/class_times.py [workload]"""
import sys
import numpy as np, torch  # noqa
import speck_amd as sa
wl = sys.argv[1] if len(sys.argv) > 1 else "scircuit"
cfg = sa.spECKConfig.initialize(0)
for k, v in (("use_graph", 0), ("merge_light", 0), ("concurrent_classes", 0)):
    cfg.set_option(k, v)
cfg.profile_kernels(1)
A = sa.gen_matrix(wl, 1.0, 1)
dA = sa.dCSR.from_host(A); dC = sa.dCSR(np.float64)
acc = None
for i in range(6):
    sa.MultiplyspECK(dA, dA, dC, cfg)
    st = cfg.last_stats()
    if i == 0:
        continue
    if acc is None:
        acc = {k: dict(v) if isinstance(v, dict) else v for k, v in st.items()}
    else:
        for k in ("sym_bin_ms", "num_bin_ms"):
            for c in st[k]:
                acc[k][c] += st[k][c]
        for k in ("analysis_ms", "scan_ms"):
            acc[k] += st[k]
n = 5
print(wl, "analysis+scatter %.1f us, scan %.1f us" % (acc["analysis_ms"] / n * 1e3, acc["scan_ms"] / n * 1e3))
for ph in ("sym", "num"):
    for c, ms in acc[ph + "_bin_ms"].items():
        rows = st[ph + "_bin_rows"][c]
        if rows:
            print(f"  {ph}:{c:12s} rows {rows:8d}  {ms / n * 1e3:8.1f} us  {ms / n * 1e6 / rows:8.1f} ns/row")
