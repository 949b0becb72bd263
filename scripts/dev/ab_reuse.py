"""A/B of options on the REUSE sequence and the complete call, alternating: python scripts/dev/ab_reuse.py webbase "b8k_full_first=0" "b8k_full_first=1" ..."""
import sys, time
import numpy as np
import speck_amd as sa

def run(kind, opts, reuse, steps=30, warm=8):
    cfg = sa.spECKConfig.initialize(0)
    cfg.set_option("reuse", reuse)
    for o in opts.split():
        k, v = o.split("=")
        cfg.set_option(k, int(v))
    A = sa.gen_matrix(kind, 1.0, 1, signed=True)
    dA = sa.dCSR.from_host(A)
    dC = sa.dCSR(A.data.dtype)
    for _ in range(warm):
        sa.MultiplyspECK(dA, dA, dC, cfg)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            sa.MultiplyspECK(dA, dA, dC, cfg)
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    st = cfg.last_stats()
    cfg.cleanup()
    return best, st["replayed"]

if __name__ == "__main__":
    for rnd in range(3):
        for o in sys.argv[2:]:
            c, _ = run(sys.argv[1], o, 0)
            r, rep = run(sys.argv[1], o, 1)
            print("%-9s %-22s complete %.4f ms   reuse %.4f ms (replayed %d)" % (sys.argv[1], o, c, r, rep), flush=True)
