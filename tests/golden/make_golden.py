"""Generates the committed golden fixtures under tests/golden/.

Run in the BUILD container (needs /root/reference for the format fixtures):
    python tests/golden/make_golden.py

* synth10k.json   -- known answers of the config-#1 generator; the figures
                     were first produced by the survey's independent script and are
                     re-derived here with the oracle and cross-checked with scipy.
* tiny_cases.json -- hand-checkable SpGEMM cases (cancellation kept, empty rows, ...),
                     expected results computed densely with numpy (independent of the oracle).
* formats/*.mtx, *.hicsr, formats.json -- MatrixMarket inputs written by this script and
                     the bytes/arrays the REFERENCE's own loadMTX/convert/storeCSR produce for
                     them (oracle/_ref/ref_formats, compiled from /root/reference/source).
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402


def sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def synth10k():
    A = po.gen_uniform(10000, 42)
    an = po.analysis(A, A)
    C, _ = po.spgemm(A, A, threads=1)
    S = A.to_scipy()
    R = (S @ S).tocsr()
    R.sort_indices()
    assert (R.indices == C.col_ids).all() and (R.indptr == C.row_offsets).all()
    out = dict(
        n=10000, seed=42, nnzA=A.nnz, P=an["sum_products"], max_row_ops=an["max_row_ops"],
        min_row_ops=int(an["row_ops"].min()), nnzC=C.nnz,
        max_row_nnzC=int(np.diff(C.row_offsets.astype(np.int64)).max()),
        a_row0_cols=[int(x) for x in A.col_ids[:6]], a_row0_vals=[float(x) for x in A.data[:3]],
        c_row0_cols=[int(x) for x in C.col_ids[:5]], c_row0_vals=[float(x) for x in C.data[:3]],
        sha_c_col_ids=sha16(C.col_ids), sha_c_row_offsets=sha16(C.row_offsets),
        sha_a_col_ids=sha16(A.col_ids), sha_row_ops=sha16(an["row_ops"]),
        sum_c_values=float(np.sum(C.data)),
    )
    json.dump(out, open(os.path.join(HERE, "synth10k.json"), "w"), indent=1)


def tiny_cases():
    cases = []

    def add(name, a, b):
        a, b = np.array(a, dtype=float), np.array(b, dtype=float)
        pat = ((a != 0).astype(int) @ (b != 0).astype(int)) > 0   # structural pattern
        cases.append(dict(name=name, a=a.tolist(), b=b.tolist(), c=(a @ b).tolist(),
                          pattern=pat.astype(int).tolist()))

    # cancelled entry must be KEPT (scipy drops it)
    add("cancellation", [[1, 1], [0, 2]], [[1, 3], [-1, 0]])
    add("empty_rows", [[0, 0, 0], [1, 0, 2], [0, 0, 0]], [[1, 2, 0], [0, 0, 0], [0, 3, 4]])
    add("single_entry_rows", [[0, 2, 0], [3, 0, 0], [0, 0, 4]], [[1, 0, 5], [0, 6, 7], [8, 9, 0]])
    add("b_rows_empty", [[1, 1, 0], [0, 1, 0]], [[0, 0], [0, 0], [1, 1]])
    add("rect", [[1, 2, 0, 0], [0, 0, 3, 4]], [[1, 0], [0, 1], [1, 1], [0, 2]])
    add("dense4", np.arange(1, 17).reshape(4, 4), np.arange(16, 0, -1).reshape(4, 4))
    json.dump(cases, open(os.path.join(HERE, "tiny_cases.json"), "w"), indent=1)


MTX = {
    "general_real.mtx": """%%MatrixMarket matrix coordinate real general
% comment line
4 5 7
1 1 1.5
3 2 -2.25
1 4 3.0
2 5 4.0
4 1 5.5
3 3 6.0
2 2 7.125
""",
    "symmetric_real.mtx": """%%MatrixMarket matrix coordinate real symmetric
4 4 5
1 1 2.0
2 1 -1.0
3 2 -1.5
4 4 9.0
4 1 0.25
""",
    "pattern_general.mtx": """%%MatrixMarket matrix coordinate pattern general
3 3 4
1 2
2 3
3 1
3 3
""",
    "integer_symmetric.mtx": """%%MatrixMarket matrix coordinate integer symmetric
3 3 3
1 1 4
3 1 -2
2 2 7
""",
}


def formats():
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_formats")
    if not os.path.exists(ref):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    fdir = os.path.join(HERE, "formats")
    os.makedirs(fdir, exist_ok=True)
    meta = {}
    for name, text in MTX.items():
        mtx = os.path.join(fdir, name)
        open(mtx, "w").write(text)
        hic = mtx[:-4] + ".hicsr"
        subprocess.check_call([ref, "mtx2hicsr", mtx, hic], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dump = subprocess.check_output([ref, "dump", hic]).decode().split("\n")
        rows, cols, nnz = [int(x) for x in dump[0].split()]
        meta[name] = dict(rows=rows, cols=cols, nnz=nnz, row_offsets=[int(x) for x in dump[1].split()],
                          col_ids=[int(x) for x in dump[2].split()], data=[float(x) for x in dump[3].split()],
                          hicsr=os.path.basename(hic))
    json.dump(meta, open(os.path.join(HERE, "formats.json"), "w"), indent=1)


if __name__ == "__main__":
    synth10k()
    tiny_cases()
    formats()
    print("golden fixtures written to", HERE)
