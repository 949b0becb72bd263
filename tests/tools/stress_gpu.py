"""Randomised parity stress of the HIP path against the oracle (run by hand on a GPU box):
power-law / uniform / clustered row lengths, empty rows and columns, rectangular shapes, fp32 and fp64,
repeated calls (graph replay) with changing values.
usage: python tests/tools/stress_gpu.py [cases] [seed] [option=value ...]   (first=N: only multiply cases >= N;
       interleave=K: keep K problems alive on the one config and multiply a random one of them each step -- captured
       sequences, predictions and scratch of several problems taking turns)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import speck_amd as sa  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def rand_csr(rng, rows, cols, mean_len, kind, dtype):
    if kind == "uniform":
        ln = rng.integers(0, 2 * mean_len + 1, size=rows)
    elif kind == "powerlaw":
        ln = np.minimum((rng.pareto(1.2, size=rows) * mean_len * 0.5).astype(np.int64), cols)
    elif kind == "sparse_empty":
        ln = np.where(rng.random(rows) < 0.7, 0, rng.integers(1, 3 * mean_len + 2, size=rows))
    else:  # "dense_band"
        ln = rng.integers(mean_len, 2 * mean_len + 2, size=rows)
    ln = np.minimum(ln, cols).astype(np.int64)
    ro = np.zeros(rows + 1, dtype=np.int64)
    ro[1:] = np.cumsum(ln)
    col = np.empty(int(ro[-1]), dtype=np.uint32)
    for r in range(rows):
        n = int(ln[r])
        if n == 0:
            continue
        if kind == "dense_band":
            lo = int(rng.integers(0, max(1, cols - 4 * n)))
            pool = np.arange(lo, min(cols, lo + 4 * n))
            col[ro[r]:ro[r + 1]] = np.sort(rng.choice(pool, size=n, replace=False))
        else:
            col[ro[r]:ro[r + 1]] = np.sort(rng.choice(cols, size=n, replace=False))
    val = ((0.5 + rng.random(col.size)) * rng.choice([-1.0, 1.0], size=col.size)).astype(dtype)
    return po.HostCSR(rows, cols, ro.astype(np.uint32), col, val)


def to_sa(h):
    return sa.HostCSR(h.rows, h.cols, h.row_offsets, h.col_ids, h.data)


def random_problem(rng):
    dtype = np.float64 if rng.random() < 0.8 else np.float32
    scale = int(os.environ.get("STRESS_SCALE", "1"))   # rows of A and B up to 3000 x scale
    m = int(rng.integers(1, 3000 * scale))
    k = int(rng.integers(1, 3000 * scale))
    n = int(rng.choice([50, 1000, 20000, 300000, 3000000]))
    ka = rng.choice(["uniform", "powerlaw", "sparse_empty", "dense_band"])
    kb = rng.choice(["uniform", "powerlaw", "sparse_empty", "dense_band"])
    A = rand_csr(rng, m, k, int(rng.choice([1, 3, 10, 40])), ka, dtype)
    B = rand_csr(rng, k, n, int(rng.choice([1, 3, 10, 40, 150])), kb, dtype)
    return dict(A=A, B=B, R=None, ab=None, dtype=dtype, dA=None, dB=None, dC=None,
                name=f"{m}x{k}x{n} {ka}/{kb} {dtype.__name__}", calls=0)


def materialise(p):
    if p["dA"] is None:
        p["R"], p["ab"] = po.spgemm_f64_of(p["A"], p["B"])
        p["dA"], p["dB"], p["dC"] = sa.dCSR.from_host(to_sa(p["A"])), sa.dCSR.from_host(to_sa(p["B"])), sa.dCSR(p["dtype"])


def interleaved(cfg, rng, steps, alive):
    probs = [random_problem(rng) for _ in range(alive)]
    bad = 0
    hostile = 0
    for it in range(steps):
        j = int(rng.integers(0, alive))
        if rng.random() < 0.08:          # now and then a problem is replaced by a new one (buffers freed, reused)
            probs[j] = random_problem(rng)
        p = probs[j]
        if it < int(os.environ.get("STRESS_SKIP_BEFORE", "0")):   # fast-forward to a step of a long run (same draws)
            p["calls"] += 1
            continue
        materialise(p)
        if rng.random() < 0.1:           # new VALUES under the same pointers and structure: a replay stays valid
            import ctypes
            from speck_amd import _lib
            A = p["A"]
            newv = (A.data * (0.5 + rng.random(A.data.size))).astype(A.data.dtype)
            p["A"] = po.HostCSR(A.rows, A.cols, A.row_offsets, A.col_ids, newv)
            assert _lib.load().speck_dcsr_update(ctypes.byref(p["dA"]._c), None, None, np.ascontiguousarray(newv).ctypes.data,
                                                 newv.dtype.itemsize) == 0
            p["R"], p["ab"] = po.spgemm_f64_of(p["A"], p["B"])
        if rng.random() < 0.06 and p["A"].nnz:   # new STRUCTURE under the same pointers (same row lengths of A): a
            import ctypes                          #   replay must notice -- whatever it predicted -- and re-run eagerly
            from speck_amd import _lib
            A = p["A"]
            ro = A.row_offsets.astype(np.int64)
            col = A.col_ids.copy()
            rows = rng.choice(A.rows, size=max(1, A.rows // 20), replace=False)
            for r in rows:
                n = int(ro[r + 1] - ro[r])
                if n:
                    col[ro[r]:ro[r + 1]] = np.sort(rng.choice(A.cols, size=n, replace=False)).astype(np.uint32)
            p["A"] = po.HostCSR(A.rows, A.cols, A.row_offsets, col, A.data)
            assert _lib.load().speck_dcsr_update(ctypes.byref(p["dA"]._c), None, np.ascontiguousarray(col).ctypes.data,
                                                 None, A.data.dtype.itemsize) == 0
            p["R"], p["ab"] = po.spgemm_f64_of(p["A"], p["B"])
        if rng.random() < 0.04 and p["B"].nnz:   # ... and the same for rows of B
            import ctypes
            from speck_amd import _lib
            B = p["B"]
            ro = B.row_offsets.astype(np.int64)
            col = B.col_ids.copy()
            for r in rng.choice(B.rows, size=max(1, B.rows // 20), replace=False):
                n = int(ro[r + 1] - ro[r])
                if n:
                    col[ro[r]:ro[r + 1]] = np.sort(rng.choice(B.cols, size=n, replace=False)).astype(np.uint32)
            p["B"] = po.HostCSR(B.rows, B.cols, B.row_offsets, col, B.data)
            assert _lib.load().speck_dcsr_update(ctypes.byref(p["dB"]._c), None, np.ascontiguousarray(col).ctypes.data,
                                                 None, B.data.dtype.itemsize) == 0
            p["R"], p["ab"] = po.spgemm_f64_of(p["A"], p["B"])
        if os.environ.get("STRESS_HOSTILE") and rng.random() < 0.08 and p["B"].nnz > 4 and p["calls"] > 0:
            # a B that violates the precondition, under the same pointers, whatever sequence the problem is in: rejected
            # (SPECK_ERR_UNSORTED), nothing out of bounds on the way, and the valid B is served again afterwards
            import ctypes
            from speck_amd import _lib
            B = p["B"]
            kind = int(rng.integers(0, 4))
            col = B.col_ids.copy()
            ro = B.row_offsets.astype(np.int64)
            if kind == 0:
                col = rng.permutation(col)
            elif kind == 1:
                pick = rng.random(col.size) < 0.2
                col[pick] = rng.integers(0xF0000000, 0xFFFFFFFF, size=int(pick.sum()), dtype=np.int64).astype(np.uint32)
            elif kind == 2:
                ln = np.diff(ro)
                col[:] = np.repeat(B.col_ids[ro[:-1][ln > 0]], ln[ln > 0])
            else:
                for r in np.flatnonzero(np.diff(ro) >= 2)[:2000]:
                    col[ro[r]:ro[r + 1]] = col[ro[r]:ro[r + 1]][::-1]
            if not (col == B.col_ids).all() and (np.diff(ro) >= 2).any():
                assert _lib.load().speck_dcsr_update(ctypes.byref(p["dB"]._c), None, np.ascontiguousarray(col).ctypes.data,
                                                     None, B.data.dtype.itemsize) == 0
                try:
                    sa.MultiplyspECK(p["dA"], p["dB"], p["dC"], cfg)
                    hostile_ok = False
                except sa.SpeckError as e:
                    hostile_ok = e.status == 8
                assert _lib.load().speck_dcsr_update(ctypes.byref(p["dB"]._c), None,
                                                     np.ascontiguousarray(B.col_ids).ctypes.data, None,
                                                     B.data.dtype.itemsize) == 0
                hostile += 1
                if not hostile_ok:
                    bad += 1
                    print(f"{it:4d} BAD hostile B kind {kind} was not rejected: problem {j} {p['name']}", flush=True)
        if os.environ.get("STRESS_VERBOSE"):
            print(f"     step {it}: problem {j} call {p['calls'] + 1} {p['name']}", flush=True)
        ok = True
        # (STRESS_REPEAT=K: the chosen problem K times in a row -- the later calls of a run are replays without a scan and,
        #  option num_verify, without a symbolic pass, and the NEXT in-place change of that problem meets such a sequence)
        for _ in range(int(os.environ.get("STRESS_REPEAT", "1"))):
            sa.MultiplyspECK(p["dA"], p["dB"], p["dC"], cfg)
            p["calls"] += 1
            got = p["dC"].to_host()
            R, ab = p["R"], p["ab"]
            tol = 1e-12 if p["dtype"] == np.float64 else 4.0 * 2.0 ** -23
            ok = ok and got.nnz == R.nnz and (got.row_offsets == R.row_offsets).all() and (got.col_ids == R.col_ids).all() and \
                bool((np.abs(got.data.astype(np.float64) - R.data.astype(np.float64)) <= tol * ab + 1e-300).all())
        st = cfg.last_stats()
        bad += 0 if ok else 1
        print(f"{it:4d} {'ok ' if ok else 'BAD'} problem {j} call {p['calls']} replayed={int(st['replayed'])} "
              f"pred={st['pred_stages']} {p['name']} nnzC={R.nnz}", flush=True)
        if os.environ.get("STRESS_VERBOSE"):
            print("       sym", {k: v for k, v in st["sym_bin_rows"].items() if v}, "num",
                  {k: v for k, v in st["num_bin_rows"].items() if v}, "pool", st["scratch_pool_bytes"], flush=True)
    if hostile:
        print("hostile B rejected:", hostile)
    print("failures:", bad)
    sys.exit(1 if bad else 0)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    cfg = sa.spECKConfig.initialize(0)
    first, alive = 0, 0
    for opt in sys.argv[3:]:   # library options name=value, e.g. nf_min_ops=1 num_global_passes=1000000 xcd_aware=7
        name, value = opt.split("=")
        if name == "first":
            first = int(value)
            continue
        if name == "interleave":
            alive = int(value)
            continue
        if name == "user_stream":     # run the pipeline on a stream of the caller (torch's current one)
            if int(value):
                cfg.set_stream(torch.cuda.current_stream().cuda_stream)
            continue
        cfg.set_option(name, int(value))
    if alive:
        return interleaved(cfg, rng, cases, alive)
    bad = 0
    for it in range(cases):
        dtype = np.float64 if rng.random() < 0.8 else np.float32
        m = int(rng.integers(1, 3000))
        k = int(rng.integers(1, 3000))
        n = int(rng.choice([50, 1000, 20000, 300000, 3000000]))
        ka = rng.choice(["uniform", "powerlaw", "sparse_empty", "dense_band"])
        kb = rng.choice(["uniform", "powerlaw", "sparse_empty", "dense_band"])
        A = rand_csr(rng, m, k, int(rng.choice([1, 3, 10, 40])), ka, dtype)
        B = rand_csr(rng, k, n, int(rng.choice([1, 3, 10, 40, 150])), kb, dtype)
        if it < first:
            continue
        R, ab = po.spgemm_f64_of(A, B)
        dA, dB, dC = sa.dCSR.from_host(to_sa(A)), sa.dCSR.from_host(to_sa(B)), sa.dCSR(dtype)
        ok = True
        for rep in range(3):  # the third call is a graph replay
            sa.MultiplyspECK(dA, dB, dC, cfg)
            got = dC.to_host()
            tol = 1e-12 if dtype == np.float64 else 4.0 * 2.0 ** -23
            ok = ok and got.nnz == R.nnz and (got.row_offsets == R.row_offsets).all() and \
                (got.col_ids == R.col_ids).all() and \
                bool((np.abs(got.data.astype(np.float64) - R.data.astype(np.float64)) <= tol * ab + 1e-300).all())
        st = cfg.last_stats()
        cls = {k: v for k, v in st["num_bin_rows"].items() if v}
        cls.update({"sym:" + k: v for k, v in st["sym_bin_rows"].items() if v and k in ("bitmap1m", "global_hash")})
        if not ok:
            bad += 1
        print(f"{it:4d} {'ok ' if ok else 'BAD'} {m}x{k}x{n} {ka}/{kb} {dtype.__name__} nnzC={R.nnz} {cls}", flush=True)
    print("failures:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
