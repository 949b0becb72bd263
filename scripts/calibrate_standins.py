#!/usr/bin/env python3
"""Fit check of the SuiteSparse stand-ins against SURVEY.md section 8's table:
n, nnz(A), longest row, P (intermediate products) and nnz(C) of A*A, computed with the CPU oracle.

    python scripts/calibrate_standins.py [kind ...]          # print the table
    SPECK_GEN_PARAMS=block=12,p_local=0.7 python scripts/calibrate_standins.py scircuit
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import speck_amd as sa  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

# SURVEY.md section 8 (P / nnzC recalled from the SpGEMM literature)
TARGET = {
    "scircuit": dict(n=170998, nnzA=958936, maxrow=353, P=8.68e6, nnzC=5.22e6),
    "webbase": dict(n=1000005, nnzA=3105536, maxrow=4700, P=69.5e6, nnzC=51.1e6),
    "mac_econ": dict(n=206500, nnzA=1273389, maxrow=44, P=7.56e6, nnzC=6.70e6),
    "cant": dict(n=62451, nnzA=4007383, maxrow=78, P=269.5e6, nnzC=17.4e6),
}


def measure(kind, scale=1.0, seed=1):
    A = sa.gen_matrix(kind, scale, seed, signed=True)
    H = po.HostCSR(A.rows, A.cols, A.row_offsets, A.col_ids, A.data)
    an = po.analysis(H, H)
    cnt, nnzc = po.symbolic(H, H)
    ln = np.diff(A.row_offsets.astype(np.int64))
    ln = np.diff(A.row_offsets.astype(np.int64))
    if os.environ.get("SPECK_CAL_BUCKETS"):
        ops = an["row_ops"].astype(np.int64)
        c = cnt[:-1].astype(np.int64)
        for lo, hi in ((1, 1), (2, 3), (4, 7), (8, 31), (32, 255), (256, 1 << 30)):
            sel = (ln >= lo) & (ln <= hi)
            if sel.any():
                print(f"   len {lo}-{hi}: rows={sel.sum()} nnzA={ln[sel].sum()} P={ops[sel].sum()} nnzC={c[sel].sum()} "
                      f"ratio={ops[sel].sum() / max(1, c[sel].sum()):.3f}")
    return dict(n=A.rows, nnzA=A.nnz, maxrow=int(ln.max()), P=an["sum_products"], nnzC=nnzc,
                max_ops=an["max_row_ops"], max_nnzc=int(cnt[:-1].max()), rows1=int((ln == 1).sum()))


def loss(kind, params, seed=1):
    os.environ["SPECK_GEN_PARAMS"] = ",".join(f"{k}={v}" for k, v in params.items())
    got, want = measure(kind, seed=seed), TARGET[kind]
    e = sum((got[k] / want[k] - 1) ** 2 for k in ("nnzA", "P", "nnzC"))
    return e + 0.02 * (got["maxrow"] / want["maxrow"] - 1) ** 2, got


def fit(kind, start, steps, rounds=6):
    """Coordinate descent with shrinking steps over the generator parameters (SPECK_GEN_PARAMS)."""
    cur = dict(start)
    best, got = loss(kind, cur)
    for rnd in range(rounds):
        for k, st in steps.items():
            for sign in (+1, -1):
                trial = dict(cur)
                trial[k] = type(cur[k])(max(0, cur[k] + sign * st))
                if trial[k] == cur[k]:
                    continue
                l, g = loss(kind, trial)
                if l < best:
                    best, cur, got = l, trial, g
                    break
        print(f"round {rnd}: loss={best:.5f} {cur} -> nnzA={got['nnzA']} P={got['P']} nnzC={got['nnzC']} "
              f"maxrow={got['maxrow']}", flush=True)
        steps = {k: (max(1, v // 2) if isinstance(v, int) else v * 0.6) for k, v in steps.items()}
    return cur


def random_search(kind, space, n, seed=0):
    """Random search over `space` = {name: {"lo": a, "hi": b} | [choices]}; prints every new best."""
    rng = np.random.default_rng(seed)
    best = 1e9
    for i in range(n):
        params = {}
        for k, v in space.items():
            if isinstance(v, list):
                params[k] = v[rng.integers(len(v))]
            else:
                params[k] = round(float(rng.uniform(v["lo"], v["hi"])), 4)
        l, got = loss(kind, params)
        if l < best:
            best = l
            print(f"[{i}] loss={l:.5f} {params} -> nnzA={got['nnzA']} P={got['P']} nnzC={got['nnzC']} "
                  f"maxrow={got['maxrow']} ratio={got['P'] / got['nnzC']:.3f}", flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--random":
        import json
        random_search(sys.argv[2], json.loads(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]) if len(sys.argv) > 5 else 0)
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--fit":
        import json
        fit(sys.argv[2], json.loads(sys.argv[3]), json.loads(sys.argv[4]), int(sys.argv[5]) if len(sys.argv) > 5 else 6)
        return
    kinds = sys.argv[1:] or list(TARGET)
    for k in kinds:
        t0 = time.time()
        got, want = measure(k), TARGET[k]
        line = [f"{k:9s}"]
        for key in ("n", "nnzA", "maxrow", "P", "nnzC"):
            line.append(f"{key}={got[key]:.4g} ({got[key] / want[key] - 1:+.1%})")
        r_got, r_want = got["P"] / got["nnzC"], want["P"] / want["nnzC"]
        line.append(f"P/nnzC={r_got:.3f} ({r_got / r_want - 1:+.1%})")
        line.append(f"max_ops={got['max_ops']} max_nnzc={got['max_nnzc']} rows1={got['rows1']} [{time.time() - t0:.1f}s]")
        print("  ".join(line), flush=True)


if __name__ == "__main__":
    main()
