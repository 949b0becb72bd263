"""Time of the multi-window paths (NUM_D2 dense windows, SYM_BM2 bitmap windows) on the inputs of
tests/test_gpu_parity.py::test_symbolic_h3_and_dense_multiwindow (123 numeric windows) and
::test_dimensions_exactly_at_the_2_27_limit-like rows (128 symbolic windows).
usage: python scripts/multiwindow_time.py [repo_root]   (repo_root: which build of the library to load)"""
import os
import sys

root = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import speck_amd as sa  # noqa: E402


def rand_csr(rows, cols, k, seed):
    rng = np.random.default_rng(seed)
    c = np.sort(rng.integers(0, cols, size=(rows, k), dtype=np.int64), axis=1)
    keep = np.ones((rows, k), dtype=bool)
    keep[:, 1:] = c[:, 1:] != c[:, :-1]
    ro = np.zeros(rows + 1, dtype=np.uint32)
    ro[1:] = np.cumsum(keep.sum(axis=1))
    return sa.HostCSR(rows, cols, ro, c[keep].astype(np.uint32), 0.5 + rng.random(int(keep.sum())))


def best_ms(cfg, dA, dB, n=6):
    dC = sa.dCSR()
    best = 1e9
    for _ in range(n):
        t = sa.Timings(measureCompleteTime=True)
        sa.MultiplyspECK(dA, dB, dC, cfg, t)
        best = min(best, t.complete)
    return best, cfg.last_stats()


cfg = sa.spECKConfig.initialize(0)
cfg.set_option("reuse", 0)
A, B = rand_csr(48, 5000, 150, 5), rand_csr(5000, 2000000, 150, 6)
dA, dB = sa.dCSR.from_host(A), sa.dCSR.from_host(B)
cfg.set_option("sym_bitmap_ratio", 0)
cfg.set_option("num_global_passes", 1 << 30)
ms, st = best_ms(cfg, dA, dB)
print(f"{root}: NUM_D2, 123 windows of 16 Ki columns, 48 rows x 22.5 k products: {ms:.3f} ms  "
      f"(rows dense16k={st['num_bin_rows']['dense16k']})")
cfg.set_option("sym_bitmap_ratio", 32)
cfg.set_option("num_global_passes", 4)
A2, B2 = rand_csr(24, 300, 300, 7), rand_csr(300, 1 << 27, 200, 8)
dA2, dB2 = sa.dCSR.from_host(A2), sa.dCSR.from_host(B2)
for gh in (0, 8192):
    try:
        cfg.set_option("gh_per_window", gh)
    except Exception:
        if gh:
            break
    ms, st = best_ms(cfg, dA2, dB2)
    t = sa.Timings(measureAll=True)
    sa.MultiplyspECK(dA2, dB2, sa.dCSR(), cfg, t)
    k = cfg.last_stats()["sym_bin_ms"]
    print(f"{root}: 128 windows of 1 Mi columns, 24 rows x 60 k products (+ NUM_G), gh_per_window={gh}: {ms:.3f} ms  "
          f"(rows bitmap1m={st['sym_bin_rows']['bitmap1m']} global_hash={st['sym_bin_rows'].get('global_hash', 0)}; "
          f"symbolic kernel ms: { {n: round(v, 4) for n, v in k.items() if v} })")
cfg.cleanup()
