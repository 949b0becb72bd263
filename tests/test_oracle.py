"""The oracle against its pins: the known-answer fixture of the config-#1 generator
(SURVEY.md 8d), the hand-checkable tiny cases, and scipy on cancellation-free inputs."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import csr_from_dense, random_csr
from oracle import pyoracle as po

G = os.path.join(os.path.dirname(__file__), "golden")


def sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def test_golden_synth10k():
    g = json.load(open(os.path.join(G, "synth10k.json")))
    A = po.gen_uniform(g["n"], g["seed"])
    assert A.nnz == g["nnzA"] == 199976
    assert [int(x) for x in A.col_ids[:6]] == g["a_row0_cols"] == [1398, 1495, 2005, 2263, 2291, 2861]
    assert np.allclose(A.data[:3], [1.28990827, 1.34051648, 1.14709463], atol=1e-8)
    an = po.analysis(A, A)
    assert an["sum_products"] == g["P"] == 3994059
    assert an["max_row_ops"] == g["max_row_ops"] == 697
    assert int(an["row_ops"].min()) == g["min_row_ops"] == 141
    C, ab = po.spgemm(A, A)
    assert C.nnz == g["nnzC"] == 3911793
    assert int(np.diff(C.row_offsets.astype(np.int64)).max()) == 672
    assert [int(x) for x in C.col_ids[:5]] == [13, 44, 98, 116, 122]
    assert np.allclose(C.data[:3], [0.64159851, 0.63060413, 0.66275535], atol=1e-8)
    # sha256 prefixes of C.col_ids / C.row_offsets as u32 LE
    assert sha16(C.col_ids) == g["sha_c_col_ids"] == "f35356355bb1a863"
    assert sha16(C.row_offsets) == g["sha_c_row_offsets"] == "0ac160c3fd6016e1"
    assert (ab >= np.abs(C.data) - 1e-15).all()


def test_tiny_cases_structural_contract():
    for case in json.load(open(os.path.join(G, "tiny_cases.json"))):
        A, B = csr_from_dense(case["a"]), csr_from_dense(case["b"])
        C, _ = po.spgemm(A, B)
        pat = np.array(case["pattern"])
        dense = np.zeros(pat.shape)
        got_pat = np.zeros(pat.shape, dtype=int)
        for r in range(C.rows):
            cols = C.col_ids[C.row_offsets[r]:C.row_offsets[r + 1]]
            assert (np.diff(cols.astype(np.int64)) > 0).all(), case["name"]
            got_pat[r, cols] = 1
            dense[r, cols] = C.data[C.row_offsets[r]:C.row_offsets[r + 1]]
        assert (got_pat == pat).all(), case["name"]            # cancelled entries are kept
        assert np.allclose(dense, np.array(case["c"])), case["name"]


def test_cancellation_kept_where_scipy_drops():
    A = csr_from_dense([[1, 1], [0, 2]])
    B = csr_from_dense([[1, 3], [-1, 0]])
    C, _ = po.spgemm(A, B)
    assert list(C.row_offsets) == [0, 2, 3]
    assert list(C.col_ids) == [0, 1, 0] and C.data[0] == 0.0


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_matches_scipy_on_positive_values(seed):
    A = random_csr(300, 200, 6, seed, signed=False, empty_row_frac=0.1)
    B = random_csr(200, 250, 5, seed + 100, signed=False, empty_row_frac=0.1)
    C, _ = po.spgemm(A, B)
    R = (A.to_scipy() @ B.to_scipy()).tocsr()
    R.sort_indices()
    assert (R.indptr == C.row_offsets).all() and (R.indices == C.col_ids).all()
    assert np.allclose(R.data, C.data, rtol=1e-13, atol=0)


def test_analysis_quantities():
    A = random_csr(100, 80, 4, 7, empty_row_frac=0.2)
    B = random_csr(80, 90, 3, 8, empty_row_frac=0.2)
    an = po.analysis(A, B)
    blen = np.diff(B.row_offsets.astype(np.int64))
    for r in range(A.rows):
        ks = A.col_ids[A.row_offsets[r]:A.row_offsets[r + 1]]
        assert an["row_ops"][r] == blen[ks].sum()
        assert an["row_max_ops"][r] == (blen[ks].max() if len(ks) else 0)
        ne = [k for k in ks if blen[k]]
        if ne:
            assert an["row_col_min"][r] == min(B.col_ids[B.row_offsets[k]] for k in ne)
            assert an["row_col_max"][r] == max(B.col_ids[B.row_offsets[k + 1] - 1] for k in ne)
        else:
            assert an["row_col_min"][r] == 0xFFFFFFFF and an["row_col_max"][r] == 0
    assert an["sum_products"] == int(an["row_ops"].astype(np.int64).sum())


def test_row_shards_concatenate_to_unsharded():
    A = random_csr(257, 257, 5, 11)
    C, _ = po.spgemm(A, A)
    cols, vals, cnts = [], [], []
    for r0, r1 in [(0, 100), (100, 101), (101, 257)]:
        S, _ = po.spgemm(A.row_slice(r0, r1), A)
        cols.append(S.col_ids)
        vals.append(S.data)
        cnts.append(np.diff(S.row_offsets.astype(np.int64)))
    assert (np.concatenate(cols) == C.col_ids).all()
    assert (np.concatenate(vals) == C.data).all()
    assert (np.concatenate(cnts) == np.diff(C.row_offsets.astype(np.int64))).all()


def test_threads_do_not_change_results():
    A = random_csr(500, 500, 8, 5)
    C1, _ = po.spgemm(A, A, threads=1)
    C8, _ = po.spgemm(A, A, threads=0)
    assert (C1.col_ids == C8.col_ids).all() and (C1.data == C8.data).all()


def test_transpose_and_f32():
    A = random_csr(60, 90, 4, 3)
    T = po.transpose(A)
    assert np.allclose(T.to_scipy().toarray(), A.to_scipy().toarray().T)
    A32 = po.HostCSR(A.rows, A.cols, A.row_offsets, A.col_ids, A.data.astype(np.float32))
    B32 = po.HostCSR(T.rows, T.cols, T.row_offsets, T.col_ids, T.data.astype(np.float32))
    C32, _ = po.spgemm(A32, B32)
    C64, _ = po.spgemm(A, T)
    assert C32.data.dtype == np.float32 and (C32.col_ids == C64.col_ids).all()
    assert np.allclose(C32.data, C64.data, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("kind", ["uniform", "scircuit", "mac_econ", "webbase", "cant", "nlpkkt"])
def test_oracle_matches_the_rocsparse_golden_vectors(kind):
    """A third-party pin that travels without a GPU: rocSPARSE's C = A*A of small stand-in inputs
    (tests/golden/make_rocsparse_golden.py wrote the vectors on an MI355X through apps/runspECK's compare path,
    the stand-in of the reference's cuSPARSE check, source/Executor.cpp:29-40).  Structure bit-exact
    (SHA-256 of row_offsets / col_ids), values within 1e-12 * sum|a*b| per entry."""
    import speck_amd as sa   # host-side generator only (speck_gen_matrix): no GPU call
    g = np.load(os.path.join(G, "rocsparse", kind + ".npz"))
    A = sa.gen_matrix(kind, float(g["scale"]), int(g["seed"]), signed=True)
    Ao = po.HostCSR(A.rows, A.cols, A.row_offsets, A.col_ids, A.data)
    C, ab = po.spgemm(Ao, Ao)
    assert (C.rows, C.cols, C.nnz) == (int(g["rows"]), int(g["cols"]), int(g["nnz"]))
    assert hashlib.sha256(np.ascontiguousarray(C.row_offsets, dtype=np.uint32).tobytes()).hexdigest() == str(g["sha_row_offsets"])
    assert hashlib.sha256(np.ascontiguousarray(C.col_ids, dtype=np.uint32).tobytes()).hexdigest() == str(g["sha_col_ids"])
    assert (np.abs(C.data - g["data"]) <= 1e-12 * ab + 1e-300).all()
