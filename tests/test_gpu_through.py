"""The THROUGH call: a complete two-phase multiply enqueued as ONE batch (speck_amd/csrc/pipeline.hip, option eager_through).

When a complete call of the same shapes has run on the config and matOut already holds buffers of the size it produced, the
numeric launches are queued right behind the scan -- no read-back in the middle.  What the host checks between the phases of
the reference's sequence (source/GPU/Multiply.cu:575-602: nnz(C) after the scan, then the allocation) the scan checks on the
device: nnz(C) = what the buffers hold, the classes with rows, the spill pool, the verdict of the input check.  A miss leaves C
untouched and the call re-runs with its read-back.  Same bar as every other path: row_offsets and col_ids bit-exact, values
within 1e-12 * sum|a*b|.
"""
import ctypes as C_

import numpy as np
import pytest

import speck_amd as sa
from speck_amd import _lib
from oracle import pyoracle as po
from test_gpu_parity import TOL32, TOL64, _assert_matches_oracle, _hostile_b, fast_random_csr, to_po, to_sa

pytestmark = pytest.mark.gpu


@pytest.fixture
def tcfg():
    c = sa.spECKConfig.initialize(0)
    c.set_option("reuse", 0)
    yield c
    c.cleanup()


def _scribble(dC, dtype=np.float64):
    n = dC.nnz
    junk_c = np.full(n, 0xDEADBEEF, dtype=np.uint32)
    junk_v = np.full(n, np.nan, dtype=dtype)
    assert _lib.load().speck_dcsr_update(C_.byref(dC._c), None, junk_c.ctypes.data, junk_v.ctypes.data, np.dtype(dtype).itemsize) == 0


@pytest.mark.parametrize("kind,scale", [("scircuit", 0.08), ("mac_econ", 0.08), ("cant", 0.1), ("webbase", 0.04), ("uniform", 0.3),
                                        ("nlpkkt", 0.004)])
def test_through_call_matches_the_oracle(tcfg, kind, scale):
    A = to_po(sa.gen_matrix(kind, scale, 7, signed=True))
    dA = sa.dCSR.from_host(to_sa(A))
    dC = sa.dCSR()
    sa.MultiplyspECK(dA, dA, dC, tcfg)           # the first call allocates C: two phases with a read-back in between
    assert tcfg.last_stats()["eager_through"] == 0
    _assert_matches_oracle(dC, A, A)
    for _ in range(3):
        _scribble(dC)                            # every entry of C is rewritten
        sa.MultiplyspECK(dA, dA, dC, tcfg)
        st = tcfg.last_stats()
        assert st["eager_through"] == 1 and st["eager_speculated"] == 1 and st["replayed"] == 0, st
        _assert_matches_oracle(dC, A, A)
    tcfg.set_option("eager_through", 0)
    sa.MultiplyspECK(dA, dA, dC, tcfg)
    two = tcfg.last_stats()
    assert two["eager_through"] == 0
    assert two["num_bin_rows"] == st["num_bin_rows"] and two["sym_bin_rows"] == st["sym_bin_rows"] and two["nnz_c"] == st["nnz_c"]


def test_through_call_in_float32(tcfg):
    A = to_po(sa.gen_matrix("mac_econ", 0.08, 3, signed=True))
    A32 = po.HostCSR(A.rows, A.cols, A.row_offsets, A.col_ids, A.data.astype(np.float32))
    R, ab = po.spgemm_f64_of(A32, A32)
    dA = sa.dCSR.from_host(to_sa(A32))
    dC = sa.dCSR(np.float32)
    for i in range(3):
        sa.MultiplyspECK(dA, dA, dC, tcfg)
        assert tcfg.last_stats()["eager_through"] == (1 if i else 0)
        got = dC.to_host()
        assert (got.row_offsets == R.row_offsets).all() and (got.col_ids == R.col_ids).all()
        assert (np.abs(got.data.astype(np.float64) - R.data) <= TOL32 * ab + 1e-300).all()


def test_another_product_size_is_noticed_on_the_device(tcfg):
    """Column ids of A rewritten in place (same row lengths): nnz(C), the classes and the pool slots move.  The scan finds
    nnz(C) != what the buffers hold: nothing of C is written by the batch, the call re-runs and reallocates."""
    rng = np.random.default_rng(5)
    A = fast_random_csr(8000, 8000, 7, 51)
    dA = sa.dCSR.from_host(to_sa(A))
    dC = sa.dCSR()
    seen = set()
    for step in range(6):
        sa.MultiplyspECK(dA, dA, dC, tcfg)
        seen.add(tcfg.last_stats()["eager_through"])
        _assert_matches_oracle(dC, A, A)
        sa.MultiplyspECK(dA, dA, dC, tcfg)
        assert tcfg.last_stats()["eager_through"] == 1
        _assert_matches_oracle(dC, A, A)
        col = A.col_ids.copy()
        ro = A.row_offsets.astype(np.int64)
        for r in rng.choice(A.rows, size=A.rows // 10, replace=False):
            n = int(ro[r + 1] - ro[r])
            if n:
                col[ro[r]:ro[r + 1]] = np.sort(rng.choice(A.cols, size=n, replace=False)).astype(np.uint32)
        A = po.HostCSR(A.rows, A.cols, A.row_offsets, col, A.data)
        assert _lib.load().speck_dcsr_update(C_.byref(dA._c), None, np.ascontiguousarray(col).ctypes.data, None, 8) == 0
    assert -1 in seen, seen


@pytest.mark.parametrize("how", ["ends_swapped", "reversed", "duplicates", "beyond_cols", "shuffled"])
def test_through_call_leaves_c_alone_on_a_b_that_is_not_sorted(how):
    """The numeric launches are queued before the host has seen the verdict of the input check -- the SCAN looks at it: C is
    bit for bit what the previous call left, the status is SPECK_ERR_UNSORTED, the config serves the valid input again."""
    A = to_po(sa.gen_matrix("scircuit", 0.06, 9, signed=True))
    Bx = _hostile_b(A, how, np.random.default_rng(5))
    cfg = sa.spECKConfig.initialize(0)
    try:
        cfg.set_option("reuse", 0)
        dA, dB = sa.dCSR.from_host(to_sa(A)), sa.dCSR.from_host(to_sa(A))
        dC = sa.dCSR()
        sa.MultiplyspECK(dA, dB, dC, cfg)
        sa.MultiplyspECK(dA, dB, dC, cfg)
        assert cfg.last_stats()["eager_through"] == 1
        before = dC.to_host()
        bad = np.ascontiguousarray(Bx.col_ids)
        assert _lib.load().speck_dcsr_update(C_.byref(dB._c), None, bad.ctypes.data, None, 8) == 0
        with pytest.raises(sa.SpeckError) as e:
            sa.MultiplyspECK(dA, dB, dC, cfg)
        assert e.value.status == 8
        after = dC.to_host()
        assert after.nnz == before.nnz and (after.row_offsets == before.row_offsets).all()
        assert (after.col_ids == before.col_ids).all() and (after.data == before.data).all()
        good = np.ascontiguousarray(A.col_ids)
        assert _lib.load().speck_dcsr_update(C_.byref(dB._c), None, good.ctypes.data, None, 8) == 0
        for _ in range(2):
            sa.MultiplyspECK(dA, dB, dC, cfg)
        assert cfg.last_stats()["eager_through"] == 1
        _assert_matches_oracle(dC, A, A)
    finally:
        cfg.cleanup()


def test_a_freed_or_smaller_matout_takes_the_two_phase_call(tcfg):
    A = fast_random_csr(4000, 4000, 6, 9)
    dA = sa.dCSR.from_host(to_sa(A))
    dC = sa.dCSR()
    for _ in range(2):
        sa.MultiplyspECK(dA, dA, dC, tcfg)
    assert tcfg.last_stats()["eager_through"] == 1
    fresh = sa.dCSR()                            # no buffers: nothing to run through into
    sa.MultiplyspECK(dA, dA, fresh, tcfg)
    assert tcfg.last_stats()["eager_through"] == 0
    _assert_matches_oracle(fresh, A, A)
    B = fast_random_csr(4000, 4000, 3, 10)       # another product on the same config and the same matOut: nnz(C) differs
    dB = sa.dCSR.from_host(to_sa(B))
    sa.MultiplyspECK(dA, dB, dC, tcfg)
    assert tcfg.last_stats()["eager_through"] in (0, -1)
    _assert_matches_oracle(dC, A, B)
