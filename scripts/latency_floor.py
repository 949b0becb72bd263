import time, numpy as np, torch
import speck_amd as sa
cfg = sa.spECKConfig.initialize(0)
A = sa.gen_matrix("uniform", 0.01, 1)
dA = sa.dCSR.from_host(A); dC = sa.dCSR(np.float64)
for g in (1, 0):
    cfg.set_option("use_graph", g)
    for _ in range(20): sa.MultiplyspECK(dA, dA, dC, cfg)
    t = time.perf_counter()
    for _ in range(200): sa.MultiplyspECK(dA, dA, dC, cfg)
    print("graph" if g else "eager", "tiny multiply us/call:", (time.perf_counter() - t) / 200 * 1e6, A.rows, A.nnz)
lib = sa._lib.load()
import ctypes as C
a = C.c_int(); 
t = time.perf_counter()
for _ in range(2000): lib.speck_config_info(cfg._h, C.byref(a), C.byref(a), C.byref(a))
print("ctypes call us:", (time.perf_counter() - t) / 2000 * 1e6)
