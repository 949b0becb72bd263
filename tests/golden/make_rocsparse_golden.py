#!/usr/bin/env python3
"""Golden vectors from a THIRD-PARTY SpGEMM for the CPU-side oracle tests: rocSPARSE's product C = A*A of small
stand-in inputs (the reference itself ships no vectors and its own compare path is cuSPARSE, source/Executor.cpp:29-40).
Run on a GPU box (needs apps/runspECK built):   python tests/golden/make_rocsparse_golden.py
Writes tests/golden/rocsparse/<case>.npz: SHA-256 of C.row_offsets / C.col_ids and the values (float64).
The inputs are regenerated from (kind, scale, seed) by speck_gen_matrix, signed values -- not stored."""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import speck_amd as sa  # noqa: E402

CASES = [("uniform", 0.01, 7), ("scircuit", 0.006, 5), ("mac_econ", 0.005, 5), ("webbase", 0.0008, 5),
         ("cant", 0.004, 5), ("nlpkkt", 0.00005, 5)]


def main():
    out = os.path.join(ROOT, "tests", "golden", "rocsparse")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(ROOT, "apps", "runspECK")
    with tempfile.TemporaryDirectory() as tmp:
        ini = os.path.join(tmp, "config.ini")
        with open(ini, "w") as f:
            f.write("CompareResult=true\nIterationsWarmUp=1\nIterationsExecution=1\n")
        for kind, scale, seed in CASES:
            dump = os.path.join(tmp, f"{kind}.hicsr")
            env = dict(os.environ, SPECK_DUMP_ROCSPARSE=dump)
            p = subprocess.run([exe, f"gen:{kind}:{scale}:{seed}", ini], cwd=tmp, env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT)
            text = p.stdout.decode()
            assert p.returncode == 0 and "compare vs rocSPARSE: ok" in text, text
            c = sa.load_hicsr(dump)
            np.savez_compressed(
                os.path.join(out, f"{kind}.npz"), kind=kind, scale=scale, seed=seed, rows=c.rows, cols=c.cols, nnz=c.nnz,
                sha_row_offsets=hashlib.sha256(np.ascontiguousarray(c.row_offsets, dtype=np.uint32).tobytes()).hexdigest(),
                sha_col_ids=hashlib.sha256(np.ascontiguousarray(c.col_ids, dtype=np.uint32).tobytes()).hexdigest(),
                data=np.asarray(c.data, dtype=np.float64))
            print(kind, scale, seed, "rows", c.rows, "nnz", c.nnz)


if __name__ == "__main__":
    main()
