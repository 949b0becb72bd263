"""The ONE-WALK complete call (speck_amd/csrc/walk.hip, option one_walk) and the edges of the look-back chain.

The walk kernel finishes the rows of the register classes inside the kernel that places the rows (role of the reference's
symbolic pass + scan + numeric pass for those rows, source/GPU/Multiply.cu:488-602, 835-1014, in one walk).  Same bar as
every other path: row_offsets and col_ids bit-exact against the oracle, values within 1e-12 * sum|a*b|.
"""
import ctypes as C_

import numpy as np
import pytest

import speck_amd as sa
from speck_amd import _lib
from oracle import pyoracle as po
from test_gpu_parity import (TOL32, TOL64, _assert_matches_oracle, _hostile_b, fast_random_csr, to_po, to_sa)

pytestmark = pytest.mark.gpu


@pytest.fixture
def wcfg():
    c = sa.spECKConfig.initialize(0)
    c.set_option("reuse", 0)
    c.set_option("one_walk", 2)
    yield c
    c.cleanup()


def _matches(dC, R, ab, tol=TOL64):
    got = dC.to_host()
    assert got.nnz == R.nnz and (got.row_offsets == R.row_offsets).all(), "row_offsets differ"
    assert (got.col_ids == R.col_ids).all(), "col_ids differ"
    err = np.abs(got.data.astype(np.float64) - R.data.astype(np.float64))
    assert (err <= tol * ab + 1e-300).all()


def _scribble(dC, dtype=np.float64):
    """junk over C's col_ids / data between two calls: a walk call must rewrite every entry"""
    n = dC.nnz
    junk_c = np.full(n, 0xDEADBEEF, dtype=np.uint32)
    junk_v = np.full(n, np.nan, dtype=dtype)
    assert _lib.load().speck_dcsr_update(C_.byref(dC._c), None, junk_c.ctypes.data, junk_v.ctypes.data, np.dtype(dtype).itemsize) == 0


@pytest.mark.parametrize("kind,scale", [("scircuit", 0.08), ("mac_econ", 0.08), ("cant", 0.1), ("webbase", 0.04), ("uniform", 0.3)])
def test_walk_call_matches_the_oracle(wcfg, kind, scale):
    A = to_po(sa.gen_matrix(kind, scale, 7, signed=True))
    R, ab = po.spgemm(A, A)
    dA = sa.dCSR.from_host(to_sa(A))
    dC = sa.dCSR()
    sa.MultiplyspECK(dA, dA, dC, wcfg)           # the first call on a config is the two-phase one (it allocates C)
    assert wcfg.last_stats()["one_walk"] == 0
    _matches(dC, R, ab)
    for _ in range(3):
        _scribble(dC)
        sa.MultiplyspECK(dA, dA, dC, wcfg)
        st = wcfg.last_stats()
        assert st["one_walk"] == 1 and st["walk_misses"] == 0, st
        assert st["replayed"] == 0
        _matches(dC, R, ab)
    # the class statistics are those of a two-phase call (the reuse planner and bench.py read them)
    wcfg.set_option("one_walk", 0)
    sa.MultiplyspECK(dA, dA, dC, wcfg)
    two = wcfg.last_stats()
    assert two["one_walk"] == 0
    assert two["num_bin_rows"] == st["num_bin_rows"] and two["sym_bin_rows"] == st["sym_bin_rows"]
    assert two["nnz_c"] == st["nnz_c"] and two["sum_products"] == st["sum_products"] and two["max_row_nnz_c"] == st["max_row_nnz_c"]


def test_walk_call_in_float32(wcfg):
    A = to_po(sa.gen_matrix("mac_econ", 0.08, 3, signed=True))
    A32 = po.HostCSR(A.rows, A.cols, A.row_offsets, A.col_ids, A.data.astype(np.float32))
    R, ab = po.spgemm_f64_of(A32, A32)
    dA = sa.dCSR.from_host(to_sa(A32))
    dC = sa.dCSR(np.float32)
    for i in range(3):
        sa.MultiplyspECK(dA, dA, dC, wcfg)
        assert wcfg.last_stats()["one_walk"] == (1 if i else 0)
        _matches(dC, R, ab, TOL32)


def test_walk_call_on_a_row_view_and_a_rectangular_b(wcfg):
    A = fast_random_csr(6000, 900, 7, 21)
    B = fast_random_csr(900, 5000, 6, 22)
    dA, dB = sa.dCSR.from_host(to_sa(A)), sa.dCSR.from_host(to_sa(B))
    view = dA.row_view(1000, 4200)
    H = po.HostCSR(A.rows, A.cols, A.row_offsets, A.col_ids, A.data).row_slice(1000, 4200)
    R, ab = po.spgemm(H, B)
    dC = sa.dCSR()
    for i in range(3):
        sa.MultiplyspECK(view, dB, dC, wcfg)
        assert wcfg.last_stats()["one_walk"] == (1 if i else 0)
        _matches(dC, R, ab)


def test_buffers_that_do_not_hold_the_product_fall_back_and_stay_exact(wcfg):
    """Same shapes, another nnz(C): the walk call is declared void on the device (every tile checks that what it places
    ends inside the caller's buffers; the last one compares nnz(C) with them) and the two-phase call re-runs and
    re-allocates, as the reference does when nnz changes (source/GPU/Multiply.cu:589-592)."""
    A = fast_random_csr(5000, 5000, 6, 31)
    B1 = fast_random_csr(5000, 5000, 5, 32)
    B2 = fast_random_csr(5000, 5000, 9, 33)   # more entries of C
    B3 = fast_random_csr(5000, 5000, 3, 34)   # fewer
    dA = sa.dCSR.from_host(to_sa(A))
    dC = sa.dCSR()
    misses = 0
    for B in (B1, B1, B2, B2, B3, B3, B1):
        dB = sa.dCSR.from_host(to_sa(B))
        had = dC.nnz
        sa.MultiplyspECK(dA, dB, dC, wcfg)
        st = wcfg.last_stats()
        R, ab = po.spgemm(A, B)
        _matches(dC, R, ab)
        if had and had != R.nnz:
            misses += 1
            assert st["one_walk"] == 0 and st["walk_misses"] == misses, st
        elif had:
            assert st["one_walk"] == 1, st


def test_an_output_that_was_freed_takes_the_two_phase_call(wcfg):
    A = fast_random_csr(3000, 3000, 8, 41)
    dA = sa.dCSR.from_host(to_sa(A))
    R, ab = po.spgemm(A, A)
    dC = sa.dCSR()
    sa.MultiplyspECK(dA, dA, dC, wcfg)
    sa.MultiplyspECK(dA, dA, dC, wcfg)
    assert wcfg.last_stats()["one_walk"] == 1
    dC.reset()                                   # matOut freed between two calls: nothing to place rows into
    sa.MultiplyspECK(dA, dA, dC, wcfg)
    assert wcfg.last_stats()["one_walk"] == 0 and wcfg.last_stats()["walk_misses"] == 0
    _matches(dC, R, ab)
    fresh = sa.dCSR()                            # ... or another, empty matOut
    sa.MultiplyspECK(dA, dA, fresh, wcfg)
    assert wcfg.last_stats()["one_walk"] == 0
    _matches(fresh, R, ab)


def test_structure_changed_under_the_same_pointers(wcfg):
    """Column ids of A rewritten in place (same row lengths): classes, pool slots and nnz(C) move -- whatever the
    walk call sized from the previous call is checked on the device, the result is exact either way."""
    rng = np.random.default_rng(5)
    A = fast_random_csr(8000, 8000, 7, 51)
    dA = sa.dCSR.from_host(to_sa(A))
    dC = sa.dCSR()
    for step in range(6):
        sa.MultiplyspECK(dA, dA, dC, wcfg)
        _assert_matches_oracle(dC, A, A)
        # new sorted column ids for a tenth of the rows
        col = A.col_ids.copy()
        ro = A.row_offsets.astype(np.int64)
        for r in rng.choice(A.rows, size=A.rows // 10, replace=False):
            n = int(ro[r + 1] - ro[r])
            if n:
                col[ro[r]:ro[r + 1]] = np.sort(rng.choice(A.cols, size=n, replace=False)).astype(np.uint32)
        A = po.HostCSR(A.rows, A.cols, A.row_offsets, col, A.data)
        assert _lib.load().speck_dcsr_update(C_.byref(dA._c), None, np.ascontiguousarray(col).ctypes.data, None, 8) == 0


@pytest.mark.parametrize("how", ["ends_swapped", "reversed", "duplicates", "beyond_cols", "shuffled"])
def test_walk_call_survives_a_b_that_is_not_sorted(how):
    """The walk kernel finishes rows BEFORE the verdict of the input check is read: on a B whose rows are not strictly
    ascending it stays inside the pool slots (a slot holds the row's products) and inside C's buffers (checked per
    tile), the call returns SPECK_ERR_UNSORTED, and the config serves the valid input again."""
    A = to_po(sa.gen_matrix("scircuit", 0.06, 9, signed=True))
    Bx = _hostile_b(A, how, np.random.default_rng(5))
    cfg = sa.spECKConfig.initialize(0)
    try:
        cfg.set_option("reuse", 0)
        cfg.set_option("one_walk", 2)
        dA, dB = sa.dCSR.from_host(to_sa(A)), sa.dCSR.from_host(to_sa(A))
        dC = sa.dCSR()
        sa.MultiplyspECK(dA, dB, dC, cfg)
        sa.MultiplyspECK(dA, dB, dC, cfg)
        assert cfg.last_stats()["one_walk"] == 1
        bad = np.ascontiguousarray(Bx.col_ids)
        assert _lib.load().speck_dcsr_update(C_.byref(dB._c), None, bad.ctypes.data, None, 8) == 0
        with pytest.raises(sa.SpeckError) as e:
            sa.MultiplyspECK(dA, dB, dC, cfg)
        assert e.value.status == 8
        good = np.ascontiguousarray(A.col_ids)
        assert _lib.load().speck_dcsr_update(C_.byref(dB._c), None, good.ctypes.data, None, 8) == 0
        for _ in range(2):
            sa.MultiplyspECK(dA, dB, dC, cfg)
        assert cfg.last_stats()["one_walk"] == 1
        _assert_matches_oracle(dC, A, A)
    finally:
        cfg.cleanup()


@pytest.mark.parametrize("rows", [64 * 4096, 64 * 4096 + 1, 256 * 4096 + 1])
def test_walk_tiles_at_the_capacity_of_the_chain(wcfg, rows):
    """4096 tiles of 64 rows (kChainMaxBlocks, chain.hpp), one row more (the tiles double), and the first row count
    whose tiles are 512 rows (two rows per thread in the tile kernel)."""
    A = fast_random_csr(rows, 3000, 3, 61)
    B = fast_random_csr(3000, 4000, 4, 62)
    dA, dB = sa.dCSR.from_host(to_sa(A)), sa.dCSR.from_host(to_sa(B))
    R, ab = po.spgemm(A, B, threads=0)
    dC = sa.dCSR()
    for i in range(2):
        sa.MultiplyspECK(dA, dB, dC, wcfg)
        assert wcfg.last_stats()["one_walk"] == i
        _matches(dC, R, ab)


@pytest.mark.parametrize("rows", [(1 << 23), (1 << 23) + 1])
def test_scan_tiles_at_the_capacity_of_the_chain(rows):
    """The scan kernel of the two-phase call with exactly kChainMaxBlocks = 4096 tiles (2^23 rows at 2048 rows per
    tile) and with one row more (8192 rows per tile): VERDICT round 5, item 6."""
    rng = np.random.default_rng(3)
    cols = 50000
    # one entry per row of A (a scaled copy of a B row each), B with short rows
    a_ro = np.arange(rows + 1, dtype=np.uint32)
    a_col = rng.integers(0, 2000, size=rows).astype(np.uint32)
    A = po.HostCSR(rows, 2000, a_ro, a_col, 0.5 + rng.random(rows))
    B = fast_random_csr(2000, cols, 3, 72)
    cfg = sa.spECKConfig.initialize(0)
    try:
        cfg.set_option("reuse", 0)
        dA, dB = sa.dCSR.from_host(to_sa(A)), sa.dCSR.from_host(to_sa(B))
        dC = sa.dCSR()
        sa.MultiplyspECK(dA, dB, dC, cfg)
        got = dC.to_host()
        blen = np.diff(B.row_offsets.astype(np.int64))
        want = np.zeros(rows + 1, dtype=np.int64)
        want[1:] = np.cumsum(blen[a_col])
        assert got.nnz == want[-1] and (got.row_offsets.astype(np.int64) == want).all()
        # three sampled blocks of rows against the oracle
        for r0 in (0, rows // 2, rows - 5000):
            H = A.row_slice(r0, r0 + 5000)
            R, ab = po.spgemm(H, B)
            lo, hi = int(want[r0]), int(want[r0 + 5000])
            assert (got.col_ids[lo:hi] == R.col_ids).all()
            assert (np.abs(got.data[lo:hi] - R.data) <= TOL64 * ab + 1e-300).all()
    finally:
        cfg.cleanup()


def test_a_chain_that_times_out_is_reported_and_the_config_recovers():
    """chain.hpp: a workgroup that waits in vain (~1 s) raises Chain::error before it publishes anything made from the
    truncated prefix, places nothing, the last workgroup reports -> SPECK_ERR_HIP with C untouched; the flag is cleared
    and the next call succeeds.  The silent workgroup is a test hook (option chain_fault)."""
    A = fast_random_csr(200000, 5000, 5, 81)
    B = fast_random_csr(5000, 6000, 5, 82)
    cfg = sa.spECKConfig.initialize(0)
    try:
        cfg.set_option("reuse", 0)
        dA, dB = sa.dCSR.from_host(to_sa(A)), sa.dCSR.from_host(to_sa(B))
        dC = sa.dCSR()
        sa.MultiplyspECK(dA, dB, dC, cfg)
        before = dC.to_host()
        for _ in range(2):                       # the analysis' chain, then (after a clean call) again
            cfg.set_option("chain_fault", 3)
            with pytest.raises(sa.SpeckError) as e:
                sa.MultiplyspECK(dA, dB, dC, cfg)
            assert e.value.status == 3
            after = dC.to_host()
            assert after.nnz == before.nnz and (after.col_ids == before.col_ids).all() and (after.data == before.data).all()
            sa.MultiplyspECK(dA, dB, dC, cfg)
            _assert_matches_oracle(dC, A, B)
    finally:
        cfg.cleanup()


def test_the_chain_makes_progress_beside_a_stream_that_fills_the_chip():
    """Forward progress under contention: a long kernel of another stream holds wave slots on every CU while the chained
    kernels (analysis, scan, walk) run -- a workgroup only ever waits for workgroups dispatched before it."""
    torch = pytest.importorskip("torch")
    A = to_po(sa.gen_matrix("scircuit", 0.5, 4, signed=True))
    R, ab = po.spgemm(A, A)
    cfg = sa.spECKConfig.initialize(0)
    try:
        cfg.set_option("reuse", 0)
        dA = sa.dCSR.from_host(to_sa(A))
        dC = sa.dCSR()
        side = torch.cuda.Stream()
        x = torch.rand(1 << 26, device="cuda")
        for walk in (0, 2):
            cfg.set_option("one_walk", walk)
            for _ in range(6):
                with torch.cuda.stream(side):
                    for _ in range(4):
                        x = torch.sin(x) * 1.0001   # ~256 MB streamed per kernel: every CU busy
                sa.MultiplyspECK(dA, dA, dC, cfg)
                _matches(dC, R, ab)
            torch.cuda.synchronize()
    finally:
        cfg.cleanup()


# ---------------------------------------------------------------- the one-walk kernel of the hash classes (option one_walk_hash)
@pytest.fixture
def hcfg():
    c = sa.spECKConfig.initialize(0)
    c.set_option("reuse", 0)
    c.set_option("one_walk_hash", 2)
    yield c
    c.cleanup()


def test_hash_walk_call_matches_the_oracle(hcfg):
    """numeric.hip: walk_hash_kernel -- eight rows per workgroup accumulated in sub-wave tables sized from the product
    bound, placed through the three-level chain (chain3.hpp), sorted and stored: no symbolic pass, no scan."""
    A = to_po(sa.gen_matrix("nlpkkt", 0.01, 3, signed=True))
    R, ab = po.spgemm(A, A)
    dA = sa.dCSR.from_host(to_sa(A))
    dC = sa.dCSR()
    sa.MultiplyspECK(dA, dA, dC, hcfg)
    assert hcfg.last_stats()["one_walk"] == 0
    for _ in range(3):
        _scribble(dC)
        sa.MultiplyspECK(dA, dA, dC, hcfg)
        assert hcfg.last_stats()["one_walk"] == 2, hcfg.last_stats()
        _matches(dC, R, ab)
    # fp32, a rectangular B, rows of every small size incl. empty ones and single entries
    rng = np.random.default_rng(2)
    A2 = fast_random_csr(70001, 3000, 9, 91)
    B2 = fast_random_csr(3000, 40000, 12, 92)
    A32 = po.HostCSR(A2.rows, A2.cols, A2.row_offsets, A2.col_ids, A2.data.astype(np.float32))
    B32 = po.HostCSR(B2.rows, B2.cols, B2.row_offsets, B2.col_ids, B2.data.astype(np.float32))
    R32, ab32 = po.spgemm_f64_of(A32, B32)
    dA2, dB2 = sa.dCSR.from_host(to_sa(A32)), sa.dCSR.from_host(to_sa(B32))
    dC2 = sa.dCSR(np.float32)
    for i in range(3):
        sa.MultiplyspECK(dA2, dB2, dC2, hcfg)
        assert hcfg.last_stats()["one_walk"] == (2 if i else 0)
        _matches(dC2, R32, ab32, TOL32)


def test_hash_walk_rows_that_outgrow_the_table_fall_back(hcfg):
    """The path is CHOSEN from the previous call's longest row of C (<= 170: the 256-entry class); a row that has more
    distinct columns than its table has slots now is noticed by the bounded probing, the call is declared void and the
    two-phase call re-runs: exact either way."""
    A = fast_random_csr(20000, 4000, 8, 101)
    B1 = fast_random_csr(4000, 60000, 10, 102)               # rows of C up to ~100 entries
    dA = sa.dCSR.from_host(to_sa(A))
    dC = sa.dCSR()
    for i in range(3):
        dB = sa.dCSR.from_host(to_sa(B1))
        sa.MultiplyspECK(dA, dB, dC, hcfg)
        assert hcfg.last_stats()["one_walk"] == (2 if i else 0)
        _assert_matches_oracle(dC, A, B1)
    B2 = fast_random_csr(4000, 60000, 60, 103)               # same shapes, rows of C of several hundred entries
    dB = sa.dCSR.from_host(to_sa(B2))
    sa.MultiplyspECK(dA, dB, dC, hcfg)
    st = hcfg.last_stats()
    assert st["one_walk"] == 0 and st["walk_misses"] == 1
    _assert_matches_oracle(dC, A, B2)
    sa.MultiplyspECK(dA, dB, dC, hcfg)                        # ... and the figures now rule the path out
    assert hcfg.last_stats()["one_walk"] == 0 and hcfg.last_stats()["walk_misses"] == 1
    _assert_matches_oracle(dC, A, B2)


@pytest.mark.parametrize("how", ["ends_swapped", "duplicates", "beyond_cols", "shuffled"])
def test_hash_walk_call_survives_a_b_that_is_not_sorted(how):
    A = to_po(sa.gen_matrix("nlpkkt", 0.005, 5, signed=True))
    Bx = _hostile_b(A, how, np.random.default_rng(5))
    cfg = sa.spECKConfig.initialize(0)
    try:
        cfg.set_option("reuse", 0)
        cfg.set_option("one_walk_hash", 2)
        dA, dB = sa.dCSR.from_host(to_sa(A)), sa.dCSR.from_host(to_sa(A))
        dC = sa.dCSR()
        sa.MultiplyspECK(dA, dB, dC, cfg)
        sa.MultiplyspECK(dA, dB, dC, cfg)
        assert cfg.last_stats()["one_walk"] == 2
        assert _lib.load().speck_dcsr_update(C_.byref(dB._c), None, np.ascontiguousarray(Bx.col_ids).ctypes.data, None, 8) == 0
        with pytest.raises(sa.SpeckError) as e:
            sa.MultiplyspECK(dA, dB, dC, cfg)
        assert e.value.status == 8
        assert _lib.load().speck_dcsr_update(C_.byref(dB._c), None, np.ascontiguousarray(A.col_ids).ctypes.data, None, 8) == 0
        for _ in range(2):
            sa.MultiplyspECK(dA, dB, dC, cfg)
        _assert_matches_oracle(dC, A, A)
    finally:
        cfg.cleanup()
