"""one-walk complete call against the oracle (development check; the -m gpu tests carry the permanent version)"""
import sys, time
import numpy as np
import speck_amd as sa
from oracle import pyoracle as po

def run(kind, scale, reuse, calls=4, opts=()):
    cfg = sa.spECKConfig.initialize(0)
    cfg.set_option("reuse", reuse)
    for k, v in opts:
        cfg.set_option(k, v)
    A = sa.gen_matrix(kind, scale, 3, signed=True)
    H = po.HostCSR(A.rows, A.cols, A.row_offsets, A.col_ids, A.data)
    R, ab = po.spgemm(H, H)
    dA = sa.dCSR.from_host(A)
    dC = sa.dCSR()
    ok = True
    for i in range(calls):
        t = sa.Timings(measureCompleteTime=True)
        sa.MultiplyspECK(dA, dA, dC, cfg, t)
        st = cfg.last_stats()
        got = dC.to_host()
        good = (got.nnz == R.nnz and (got.row_offsets == R.row_offsets).all() and (got.col_ids == R.col_ids).all()
                and (np.abs(got.data - R.data) <= 1e-12 * ab + 1e-300).all())
        print(kind, scale, "reuse", reuse, "call", i, "one_walk", st["one_walk"], "misses", st["walk_misses"], "replayed", st["replayed"],
              "spec", st["eager_speculated"], "nnz", got.nnz, "ms %.4f" % t.complete, "OK" if good else "WRONG", flush=True)
        if not good:
            ok = False
            bad = np.nonzero(got.row_offsets != R.row_offsets)[0]
            print("  first bad offset row", bad[:5], got.row_offsets[bad[:5]], R.row_offsets[bad[:5]])
            if got.nnz == R.nnz:
                badc = np.nonzero(got.col_ids != R.col_ids)[0]
                print("  bad cols", badc.size, badc[:10])
                e = np.abs(got.data - R.data) > 1e-12 * ab + 1e-300
                print("  bad vals", e.sum())
    cfg.cleanup()
    return ok

if __name__ == "__main__":
    allok = True
    for kind, scale in [("nlpkkt", 0.01), ("nlpkkt", 0.1), ("scircuit", 0.05), ("mac_econ", 0.05), ("uniform", 1.0), ("webbase", 0.1), ("cant", 0.2)]:
        for reuse in (0, 1):
            allok &= run(kind, scale, reuse, opts=(('one_walk', 2),))
    print("ALL OK" if allok else "FAILURES")
    sys.exit(0 if allok else 1)
