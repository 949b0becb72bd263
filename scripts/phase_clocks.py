"""Per-phase cycle table of the numeric hash kernels (library built with `make PHASE_CLOCKS=1`).
usage: python scripts/phase_clocks.py [workload] [--opt name=value ...]"""
import ctypes as C
import sys

import numpy as np
import torch  # noqa: F401  (one HIP runtime)

import speck_amd as sa
from speck_amd import _lib

PHASES = ["init", "products", "sort(total)", "s:load", "s:l1 build", "s:l1 prefix", "s:rank", "s:l2 build",
          "s:l2 prefix", "s:emit", "p:meta+scan", "p:search+issue", "p:accumulate", "p:sync", "p:owner windows", "p:gather"]


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "scircuit"
    lib = _lib.load()
    fn = lib.speck_debug_phase_clocks
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p]
    cfg = sa.spECKConfig.initialize(0)
    for a in sys.argv[1:]:
        if "=" in a and not a.startswith("--"):
            k, v = a.split("=")
            cfg.set_option(k, int(v))
    cfg.set_option("reuse", 0)
    A = sa.gen_matrix(wl, 1.0, 1)
    dA = sa.dCSR.from_host(A)
    dC = sa.dCSR(np.float64)
    buf = np.zeros(16 * 16, dtype=np.uint64)   # kMaxClasses x 16 phases
    sa.MultiplyspECK(dA, dA, dC, cfg)
    fn(buf.ctypes.data)
    sa.MultiplyspECK(dA, dA, dC, cfg)
    fn(buf.ctypes.data)
    st = cfg.last_stats()
    t = buf.reshape(16, 16)
    for ci, name in enumerate(sa.api.NUM_CLASS_NAMES):
        rows = st["num_bin_rows"][name]
        if rows == 0 or t[ci].sum() == 0:
            continue
        print(f"{name}: rows {rows}")
        for pi, pn in enumerate(PHASES):
            print(f"   {pn:14s} {t[ci, pi] / rows:10.0f} cycles/row")


if __name__ == "__main__":
    main()
