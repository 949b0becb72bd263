"""NUM_B8K rows in column slices (option slice_rows) against the two workgroup(512) launches: same C, and the time.
python scripts/dev/slice_check.py "webbase:0.2,webbase" """
import sys, time
import numpy as np
import speck_amd as sa

def product(kind, scale, opts, reuse, dtype=np.float64, calls=3):
    cfg = sa.spECKConfig.initialize(0)
    cfg.set_option("reuse", reuse)
    for k, v in opts.items():
        cfg.set_option(k, v)
    A = sa.gen_matrix(kind, scale, 1, signed=True)
    if dtype == np.float32:
        A = sa.HostCSR(A.rows, A.cols, A.row_offsets, A.col_ids, A.data.astype(np.float32))
    dA = sa.dCSR.from_host(A)
    dC = sa.dCSR(A.data.dtype)
    for _ in range(calls):
        sa.MultiplyspECK(dA, dA, dC, cfg)
    st = cfg.last_stats()
    C = dC.to_host()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            sa.MultiplyspECK(dA, dA, dC, cfg)
        best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
    cfg.cleanup()
    return C, st, best

if __name__ == "__main__":
    for w in sys.argv[1].split(","):
        scale = 1.0
        if ":" in w:
            w, sc = w.split(":")
            scale = float(sc)
        for dtype in (np.float64, np.float32):
            for reuse in (0, 1):
                C0, st0, t0 = product(w, scale, {"slice_rows": 0}, reuse, dtype)
                C1, st1, t1 = product(w, scale, {"slice_rows": 1}, reuse, dtype)
                same = (np.array_equal(C0.row_offsets, C1.row_offsets) and np.array_equal(C0.col_ids, C1.col_ids))
                tol = 1e-12 if dtype == np.float64 else 1e-4
                err = float(np.max(np.abs(C0.data - C1.data) / (np.abs(C0.data) + 1e-30))) if same and C0.data.size else -1.0
                print("%-8s %.2f %s reuse %d: structure %s  max rel diff %.2e  %.4f -> %.4f ms  (replayed %s / %s)" % (
                    w, scale, dtype.__name__, reuse, "same" if same else "DIFFERS", err, t0, t1,
                    st0.get("graph_replayed", st0.get("replayed")), st1.get("graph_replayed", st1.get("replayed"))), flush=True)
