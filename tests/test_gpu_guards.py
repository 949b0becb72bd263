"""Out-of-bounds writes made visible (debug option guard_bytes / SPECK_GUARD_BYTES, speck_amd/csrc/guards.hpp).

The library's kernels walk B before the input check has spoken and place rows by predictions verified afterwards; what
keeps that safe is that every kernel stays inside its buffers whatever the inputs hold.  Here that claim is CHECKED:
canary zones around C's arrays, the arena's regions and the pools, compared after every call -- on the hostile inputs of
test_gpu_parity.py, on the randomised sequences, and per ROW: a rejected reuse sequence must not have written a byte
outside the room of the row that changed.
"""
import ctypes as C_
import os
import subprocess
import sys

import numpy as np
import pytest

import speck_amd as sa
from speck_amd import _lib
from oracle import pyoracle as po
from test_gpu_parity import TOL64, _assert_matches_oracle, fast_random_csr, to_sa

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_a_write_past_the_end_of_c_is_reported():
    """The detector itself: eight entries written behind C.col_ids (through the container's own update call with a
    doctored nnz) -> the next multiply returns SPECK_ERR_HIP; without the damage the same call is fine."""
    cfg = sa.spECKConfig.initialize(0)
    try:
        cfg.set_option("guard_bytes", 4096)
        cfg.set_option("reuse", 0)
        A = fast_random_csr(2000, 2000, 6, 3)
        dA = sa.dCSR.from_host(to_sa(A))
        dC = sa.dCSR()
        sa.MultiplyspECK(dA, dA, dC, cfg)
        sa.MultiplyspECK(dA, dA, dC, cfg)
        _assert_matches_oracle(dC, A, A)
        n = dC.nnz
        junk = np.zeros(n + 8, dtype=np.uint32)
        got = dC.to_host()
        junk[:n] = got.col_ids
        dC._c.nnz = n + 8
        assert _lib.load().speck_dcsr_update(C_.byref(dC._c), None, junk.ctypes.data, None, 8) == 0
        dC._c.nnz = n
        with pytest.raises(sa.SpeckError) as e:
            sa.MultiplyspECK(dA, dA, dC, cfg)
        assert e.value.status == 3
    finally:
        cfg.set_option("guard_bytes", 0)
        cfg.cleanup()


def test_row_room_of_a_rejected_reuse_sequence():
    """ONE row of C changes under the same pointers (a row of B that only one row of A references grows duplicates: the
    row's table / sort sees more products per column and other ids than its room was made for).  The reuse sequence
    walks it before anything is rejected; whatever it stored must lie inside THAT row's room: every other entry of C
    is what the previous call left -- column ids bit for bit, values within the bound."""
    rng = np.random.default_rng(17)
    rows, inner, cols = 6000, 3000, 9000
    for trial, (ka, kb) in enumerate([(6, 5), (12, 20), (3, 300)]):     # register classes, hash classes, workgroup rows
        A = fast_random_csr(rows, inner - 1, ka, 100 + trial)
        B = fast_random_csr(inner, cols, kb, 200 + trial)
        # the last row of B is referenced by exactly one row of A: r_star
        r_star = int(rng.integers(rows // 4, 3 * rows // 4))
        ro = A.row_offsets.astype(np.int64)
        a_col = A.col_ids.copy()
        assert ro[r_star + 1] > ro[r_star]
        a_col[ro[r_star + 1] - 1] = inner - 1          # (the largest id of the row: still ascending)
        A = po.HostCSR(rows, inner, A.row_offsets, a_col, A.data)
        cfg = sa.spECKConfig.initialize(0)
        try:
            cfg.set_option("guard_bytes", 4096)
            dA, dB = sa.dCSR.from_host(to_sa(A)), sa.dCSR.from_host(to_sa(B))
            dC = sa.dCSR()
            for _ in range(4):
                sa.MultiplyspECK(dA, dB, dC, cfg)
            assert cfg.last_stats()["replayed"] == 1
            before = dC.to_host()
            R, ab = po.spgemm(A, B)
            # hostile: the entries of B's last row all become its LAST column id (duplicates, not ascending)
            b_ro = B.row_offsets.astype(np.int64)
            bad = B.col_ids.copy()
            bad[b_ro[inner - 1]:b_ro[inner]] = B.col_ids[b_ro[inner] - 1]
            assert _lib.load().speck_dcsr_update(C_.byref(dB._c), None, np.ascontiguousarray(bad).ctypes.data, None, 8) == 0
            with pytest.raises(sa.SpeckError) as e:
                sa.MultiplyspECK(dA, dB, dC, cfg)
            assert e.value.status == 8                  # (not 3: no canary zone was touched either)
            after = dC.to_host()
            assert after.nnz == before.nnz and (after.row_offsets == before.row_offsets).all()
            lo, hi = int(before.row_offsets[r_star]), int(before.row_offsets[r_star + 1])
            outside = np.ones(before.nnz, dtype=bool)
            outside[lo:hi] = False
            assert (after.col_ids[outside] == before.col_ids[outside]).all(), "a store left the room of the row that changed"
            assert (np.abs(after.data[outside] - R.data[outside]) <= TOL64 * ab[outside] + 1e-300).all()
            # ... and the config serves the valid input again
            assert _lib.load().speck_dcsr_update(C_.byref(dB._c), None, np.ascontiguousarray(B.col_ids).ctypes.data, None, 8) == 0
            for _ in range(2):
                sa.MultiplyspECK(dA, dB, dC, cfg)
            _assert_matches_oracle(dC, A, B)
        finally:
            cfg.set_option("guard_bytes", 0)
            cfg.cleanup()


@pytest.mark.parametrize("select", ["test_every_kernel_family_survives_a_b_that_is_not_sorted",
                                    "test_randomised", "test_walk_call_survives or test_structure_changed"])
def test_hostile_and_randomised_inputs_under_canary_zones(select):
    """The tests that feed the kernels inputs they must survive, repeated in a process whose every device buffer
    carries canary zones (SPECK_GUARD_BYTES): a touched zone turns the call into SPECK_ERR_HIP and fails them."""
    env = dict(os.environ, SPECK_GUARD_BYTES="4096", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-k", select,
                        os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_walk.py")],
                       env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "guard_bytes" not in r.stderr, tail


def test_a_sequence_declared_void_while_its_workgroups_start():
    """A workgroup of a replayed sequence reads DeviceStats::capacity_miss when it starts; another workgroup of the same
    launch may raise it at that moment (nf_dense_kernel with direct placement on a B overwritten in place: a row whose nnz
    is no longer what its place in C was made for).  Round 6 found waves of ONE workgroup taking different decisions: the
    ones that stayed gathered B through staging entries the ones that left never wrote -- whatever the previous kernel had
    left in LDS, e.g. the 40 Mi-column ids of the global-key-set input -- and the process died of a memory fault once in
    ~10 runs of exactly this sequence of tests (row_groups.hpp, block_void: one decision per workgroup).  Repeated here,
    under canary zones, each time in a fresh process."""
    env = dict(os.environ, SPECK_GUARD_BYTES="4096", PYTHONPATH=ROOT)
    for attempt in range(6):
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-k",
                            "test_every_kernel_family_survives_a_b_that_is_not_sorted and (global_key_set or numeric_first)",
                            os.path.join(ROOT, "tests", "test_gpu_parity.py")], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "10 passed" in r.stdout, (attempt, (r.stdout + r.stderr)[-2000:])
