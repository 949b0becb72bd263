"""Cycles per wave and phase of analysis_kernel (library built with `make PHASE_CLOCKS=1`): python scripts/analysis_clocks.py [workload]"""
import ctypes as C, numpy as np, torch, sys
import speck_amd as sa
from speck_amd import _lib
lib = _lib.load(); fn = lib.speck_debug_analysis_clocks; fn.restype = C.c_int; fn.argtypes = [C.c_void_p]
cfg = sa.spECKConfig.initialize(0); cfg.set_option("reuse", 0)
for o in sys.argv[2:]:
    n, v = o.split("="); cfg.set_option(n, int(v)); print(n, v)
A = sa.gen_matrix(sys.argv[1] if len(sys.argv) > 1 else "scircuit", 1.0, 1); dA = sa.dCSR.from_host(A); dC = sa.dCSR(np.float64)
buf = np.zeros(16, dtype=np.uint64)
for _ in range(3):
    sa.MultiplyspECK(dA, dA, dC, cfg); fn(buf.ctypes.data)
sa.MultiplyspECK(dA, dA, dC, cfg); fn(buf.ctypes.data)
avg = A.nnz / max(A.rows, 1)
waves = ((A.rows + 255) // 256) * (4 if avg <= 16 else 8)
names = ["s_ro load", "a_col+b_ro+stores", "b_col first/last", "row search", "atomics / row path", "coop rows", "reduce + barrier",
         "chain (look-back)", "binning pass"]
for n, v in zip(names, buf): print(f"{n:22s} {v / waves:10.0f} cycles/wave  (100 MHz s_memtime? clock64: shader clock)")
