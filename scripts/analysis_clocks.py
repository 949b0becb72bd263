import ctypes as C, numpy as np, torch, sys
import speck_amd as sa
from speck_amd import _lib
lib=_lib.load(); fn=lib.speck_debug_analysis_clocks; fn.restype=C.c_int; fn.argtypes=[C.c_void_p]
cfg=sa.spECKConfig.initialize(0); cfg.set_option("use_graph",0)
A=sa.gen_matrix(sys.argv[1] if len(sys.argv)>1 else "scircuit",1.0,1); dA=sa.dCSR.from_host(A); dC=sa.dCSR(np.float64)
buf=np.zeros(8,dtype=np.uint64)
sa.MultiplyspECK(dA,dA,dC,cfg); fn(buf.ctypes.data); sa.MultiplyspECK(dA,dA,dC,cfg); fn(buf.ctypes.data)
waves = ((A.rows+255)//256)*8
names=["s_ro load","a_col+b_ro+stores","b_col first/last","row search","atomics","finalize+stores","end barrier"]
for n,v in zip(names,buf): print(f"{n:20s} {v/waves:10.0f} cycles/wave")
