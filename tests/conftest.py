import os
import subprocess
import sys

import numpy as np
import torch  # noqa: F401  FIRST: torch and libspeck_amd.so each link a HIP runtime, the first one loaded serves the process
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The oracle is (re)built on demand; the HIP library must already be in-tree
    (built by `make` / __graft_entry__.build()) -- it is never silently replaced."""
    from oracle import pyoracle
    pyoracle.lib()
    lib = os.path.join(ROOT, "speck_amd", "libspeck_amd.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-C", ROOT, "speck_amd/libspeck_amd.so"])
    yield


def csr_from_dense(d):
    """Host CSR of a dense array keeping explicit pattern d != 0 (test helper)."""
    from oracle.pyoracle import HostCSR
    d = np.asarray(d, dtype=np.float64)
    ro = [0]
    ci, da = [], []
    for r in range(d.shape[0]):
        for c in range(d.shape[1]):
            if d[r, c] != 0:
                ci.append(c)
                da.append(d[r, c])
        ro.append(len(ci))
    return HostCSR(d.shape[0], d.shape[1], np.array(ro, dtype=np.uint32), np.array(ci, dtype=np.uint32),
                   np.array(da, dtype=np.float64))


def random_csr(rows, cols, density_per_row, seed, signed=True, empty_row_frac=0.0, dtype=np.float64):
    """Random CSR with sorted unique columns per row (test helper, numpy RNG)."""
    from oracle.pyoracle import HostCSR
    rng = np.random.default_rng(seed)
    ro = np.zeros(rows + 1, dtype=np.uint32)
    cis, das = [], []
    for r in range(rows):
        if rng.random() < empty_row_frac:
            k = 0
        else:
            k = int(min(cols, rng.poisson(density_per_row)))
        c = np.sort(rng.choice(cols, size=k, replace=False)).astype(np.uint32) if k else np.zeros(0, np.uint32)
        v = 0.5 + rng.random(k)
        if signed:
            v *= rng.choice([-1.0, 1.0], size=k)
        cis.append(c)
        das.append(v)
        ro[r + 1] = ro[r] + k
    ci = np.concatenate(cis) if cis else np.zeros(0, np.uint32)
    da = np.concatenate(das) if das else np.zeros(0)
    return HostCSR(rows, cols, ro, ci, da.astype(dtype))
