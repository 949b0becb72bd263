import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import speck_amd as sa
kind = sys.argv[1]; opt = sys.argv[2:] 
cfg = sa.spECKConfig.initialize(0)
for o in opt:
    n, v = o.split("="); cfg.set_option(n, int(v))
# something small first (as the test module does)
for k, sc in (("scircuit", 1.0), ("mac_econ", 1.0), ("cant", 1.0), (kind, 1.0)):
    A = sa.gen_matrix(k, sc, 1, signed=True)
    dA = sa.dCSR.from_host(A); dC = sa.dCSR()
    for i in range(7):
        sa.MultiplyspECK(dA, dA, dC, cfg)
    st = cfg.last_stats()
    print(k, "ok", st["pred_stages"], st["graph_replays"], flush=True)
cfg.cleanup()
print("done", flush=True)
