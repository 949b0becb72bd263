"""How the CPU oracle scales with threads on this host (bench.py picks the thread count).
Lives under tests/: the oracle is test infrastructure, nothing outside tests/, smoke() and the
cpu_baseline leg of bench.py may load it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import speck_amd as sa
from oracle import pyoracle as po
print("nproc affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count(), "omp max", po.max_threads())
A = sa.gen_matrix("scircuit", 1.0, 1)
H = po.HostCSR(A.rows, A.cols, A.row_offsets, A.col_ids, A.data)
P = po.analysis(H, H)["sum_products"]
for th in (1, 2, 4, 8, 16, 32, 64, 128):
    C, _ = po.spgemm(H, H, threads=th, with_abs=False)
    t = time.perf_counter(); n = 0
    while time.perf_counter() - t < 1.0:
        po.spgemm(H, H, threads=th, with_abs=False, out=C); n += 1
    dt = (time.perf_counter() - t) / n
    print(th, "threads", round(dt * 1e3, 2), "ms", round(2 * P / dt / 1e9, 3), "GFLOP/s")
