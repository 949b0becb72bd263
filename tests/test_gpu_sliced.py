"""NUM_B8K rows produced in COLUMN SLICES of the 2 Ki table (speck_amd/csrc/numeric.hip, num_sliced_body; option slice_rows).

Role of the reference's largest shared-memory maps for rows of a few thousand entries
(include/GPU/spECK_HashSpGEMM.cuh:1300-1436, source/GPU/Multiply.cu:700-760).  Same bar as every other path: row_offsets and
col_ids bit-exact against the oracle, values within 1e-12 * sum|a*b| -- complete calls and the replayed sequence (whose
launch verifies the row lengths itself).  (Hostile B under this option: test_gpu_parity.py, case block_hash_8k_sliced.)
"""
import numpy as np
import pytest

import speck_amd as sa
from oracle import pyoracle as po
from test_gpu_parity import TOL32, TOL64, fast_random_csr, to_sa

pytestmark = pytest.mark.gpu


def _matches(dC, R, ab, tol=TOL64):
    got = dC.to_host()
    assert got.nnz == R.nnz and (got.row_offsets == R.row_offsets).all(), "row_offsets differ"
    assert (got.col_ids == R.col_ids).all(), "col_ids differ"
    err = np.abs(got.data.astype(np.float64) - R.data.astype(np.float64))
    assert (err <= tol * ab + 1e-300).all()


def _run(A, B, R, ab, dtype=np.float64, tol=TOL64, want_b8k=True, slice_rows=1):
    cfg = sa.spECKConfig.initialize(0)
    try:
        cfg.set_option("slice_rows", slice_rows)
        dA, dB = sa.dCSR.from_host(to_sa(A)), sa.dCSR.from_host(to_sa(B))
        dC = sa.dCSR(dtype)
        cfg.set_option("reuse", 0)
        for _ in range(2):                      # the second call is sized from the first
            sa.MultiplyspECK(dA, dB, dC, cfg)
            _matches(dC, R, ab, tol)
        st = cfg.last_stats()
        assert (st["num_bin_rows"]["block8k"] > 0) == want_b8k, st["num_bin_rows"]
        cfg.set_option("reuse", 1)
        for _ in range(4):
            sa.MultiplyspECK(dA, dB, dC, cfg)
        assert cfg.last_stats()["replayed"] == 1
        _matches(dC, R, ab, tol)
        return st
    finally:
        cfg.cleanup()


def _clustered(rows_a, len_a, rows_b, len_b, cols, centre_span, spread, seed):
    """B rows whose columns sit in a window of `spread` columns around a per-row centre, A rows that pick B rows with
    NEIGHBOURING centres: many products per distinct column (ops >> nnz), bins of very different weight."""
    rng = np.random.default_rng(seed)
    centre = np.sort(rng.integers(0, centre_span, size=rows_b)) + (cols - centre_span - spread) // 2
    bcol = np.sort(centre[:, None] + rng.integers(0, spread, size=(rows_b, len_b)), axis=1)
    keep = np.ones(bcol.shape, dtype=bool)
    keep[:, 1:] = bcol[:, 1:] != bcol[:, :-1]
    bro = np.zeros(rows_b + 1, dtype=np.uint32)
    bro[1:] = np.cumsum(keep.sum(axis=1))
    bc = bcol[keep].astype(np.uint32)
    B = po.HostCSR(rows_b, cols, bro, bc, (0.5 + rng.random(bc.size)) * rng.choice([-1.0, 1.0], size=bc.size))
    first = rng.integers(0, rows_b - 4 * len_a, size=rows_a)
    acol = np.sort(first[:, None] + rng.choice(4 * len_a, size=(rows_a, len_a)), axis=1)
    keep = np.ones(acol.shape, dtype=bool)
    keep[:, 1:] = acol[:, 1:] != acol[:, :-1]
    aro = np.zeros(rows_a + 1, dtype=np.uint32)
    aro[1:] = np.cumsum(keep.sum(axis=1))
    ac = acol[keep].astype(np.uint32)
    return po.HostCSR(rows_a, rows_b, aro, ac, 0.5 + rng.random(ac.size)), B


@pytest.mark.parametrize("cols", [9000, 300000, 2 << 20])
def test_sliced_rows_match_the_oracle(cols):
    """Rows of ~2-4 k entries over a narrow range (a bin is ONE column), a wide one, and the widest the option takes
    (512 columns per bin)."""
    A = fast_random_csr(300, 6000, 64, 3)
    B = fast_random_csr(6000, cols, 64, 4)
    R, ab = po.spgemm(A, B)
    st = _run(A, B, R, ab)
    assert st["max_row_nnz_c"] > 1740


def test_sliced_rows_with_many_products_per_column():
    A, B = _clustered(160, 120, 4000, 90, 1 << 20, 40000, 600, 21)
    R, ab = po.spgemm(A, B)
    nnz = np.diff(R.row_offsets.astype(np.int64))
    assert ((nnz > 1740) & (nnz <= 6963)).sum() > 20
    _run(A, B, R, ab)


def test_sliced_rows_in_float32():
    A = fast_random_csr(200, 6000, 64, 3)
    B = fast_random_csr(6000, 300000, 64, 4)
    A32 = po.HostCSR(A.rows, A.cols, A.row_offsets, A.col_ids, A.data.astype(np.float32))
    B32 = po.HostCSR(B.rows, B.cols, B.row_offsets, B.col_ids, B.data.astype(np.float32))
    R, ab = po.spgemm_f64_of(A32, B32)
    _run(A32, B32, R, ab, np.float32, TOL32)


def test_option_is_inert_beyond_two_mi_columns_and_for_rows_of_too_many_products():
    """cols(B) > kSliceMaxCols: the two workgroup(512) launches as before.  A row with more products than the cut arrays
    could take slices for (kSliceMaxOps = 49 152) is not a NUM_B8K row under the option: dense windows / spill."""
    A = fast_random_csr(120, 6000, 64, 3)
    B = fast_random_csr(6000, (2 << 20) + 1, 64, 4)
    R, ab = po.spgemm(A, B)
    _run(A, B, R, ab)
    # 700 entries x 90 products = 63 k products onto <= 6 k columns
    A2, B2 = _clustered(24, 700, 6000, 90, 1 << 20, 12000, 600, 5)
    R2, ab2 = po.spgemm(A2, B2)
    nnz = np.diff(R2.row_offsets.astype(np.int64))
    assert ((nnz > 1740) & (nnz <= 6963)).any()
    with_opt = _run(A2, B2, R2, ab2, want_b8k=False)
    without = _run(A2, B2, R2, ab2, want_b8k=True, slice_rows=0)
    assert with_opt["nnz_c"] == without["nnz_c"]
