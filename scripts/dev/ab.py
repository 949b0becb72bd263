"""A/B of library options on complete calls (reuse = 0): python scripts/dev/ab.py "w1,w2" "base" "one_walk=0" "one_walk=1 x=2" """
import sys, time
import numpy as np
import speck_amd as sa

def run(kind, scale, opts, steps=40, warm=10, dtype=np.float64):
    cfg = sa.spECKConfig.initialize(0)
    cfg.set_option("reuse", 0)
    for o in opts.split():
        if o == "base":
            continue
        k, v = o.split("=")
        cfg.set_option(k, int(v))
    A = sa.gen_matrix(kind, scale, 1, signed=True)
    if dtype == np.float32:
        A = sa.HostCSR(A.rows, A.cols, A.row_offsets, A.col_ids, A.data.astype(np.float32))
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
    print("%-9s %-28s %.4f ms  walk %d misses %d spec %d through %d  %.1f GF" % (kind, opts, best, st["one_walk"], st["walk_misses"], st["eager_speculated"], st["eager_through"],
          2.0 * st["sum_products"] / best / 1e6), flush=True)
    cfg.cleanup()

if __name__ == "__main__":
    wl = sys.argv[1].split(",")
    for w in wl:
        scale = 1.0
        if ":" in w:
            w, sc = w.split(":")
            scale = float(sc)
        for v in sys.argv[2:]:
            run(w, scale, v)
