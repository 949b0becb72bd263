"""Result checks shared by tests/ and bench.py (TEST INFRASTRUCTURE ONLY, like everything under oracle/).

The reference compares the product after every iteration of its benchmark loop (source/Executor.cpp:51-55,
67-71, against cuSPARSE); here the output of the LAST timed step (of the complete call, and of the structure-reuse mode beside it) is compared
with the CPU oracle: row_offsets and col_ids bit-exact, values |c - c_ref| <= tol * sum|a*b| per entry.
Inputs too large for the oracle in seconds (the full-size nlpkkt stand-in) are checked through size-independent
properties on the device (row offsets consistent, every row strictly ascending and in range, row sums
C*1 == A*(A*1)) plus the oracle on sampled row blocks.
PARITY UNPINNED BY THE REFERENCE (see oracle/speck_oracle.h).
"""
import numpy as np

from . import pyoracle as po

TOL64 = 1e-12
# fp32: every product is rounded to float once (eps32 / 2 each), the sum is accumulated in fp64 cells and rounded to float
# once -- |c - c_ref| <= eps32 * sum|a*b| against the product computed in fp64; 4 eps32 asserted (eps32 = 2^-23)
TOL32 = 4.0 * 2.0 ** -23


def _as_po(A):
    return A if isinstance(A, po.HostCSR) else po.HostCSR(A.rows, A.cols, A.row_offsets, A.col_ids, A.data)


def compare_with_reference(ref, got_ro, got_col, got_val, tol):
    """`ref` = (R, ab): the oracle's product and its per-entry sum|a*b| (pyoracle.spgemm_f64_of).  Returns (ok, detail)."""
    R, ab = ref
    d = {"oracle_nnz": int(R.nnz), "got_nnz": int(len(got_col)), "tol": tol}
    if len(got_col) != R.nnz or len(got_ro) != len(R.row_offsets):
        return False, dict(d, why="nnz / rows differ")
    if not (np.asarray(got_ro) == R.row_offsets).all():
        return False, dict(d, why="row_offsets differ")
    if not (np.asarray(got_col) == R.col_ids).all():
        return False, dict(d, why="col_ids differ")
    err = np.abs(np.asarray(got_val, dtype=np.float64) - R.data.astype(np.float64))
    bound = tol * ab.astype(np.float64) + 1e-300
    worst = float(np.max(err / bound)) if err.size else 0.0
    d["max_err_over_bound"] = round(worst, 4)
    return bool(worst <= 1.0), d


def compare_with_oracle(A, B, got_ro, got_col, got_val, tol=None, threads=0):
    """Full comparison.  Returns (ok, detail dict).  fp32 inputs are compared with their product in fp64 (TOL32)."""
    A, B = _as_po(A), _as_po(B)
    if tol is None:
        tol = TOL32 if A.data.dtype == np.float32 else TOL64
    return compare_with_reference(po.spgemm_f64_of(A, B, threads=threads), got_ro, got_col, got_val, tol)


def _row_ids(torch, ro, nnz):
    counts = (ro[1:] - ro[:-1]).to(torch.int64)
    return torch.repeat_interleave(torch.arange(counts.numel(), device=ro.device), counts, output_size=int(nnz))


def device_properties(torch, a_ro, a_col, a_val, r0, r1, c_ro, c_col, c_val, cols, tol=1e-11):
    """Size-independent checks of C = A[r0:r1, :] * A on device tensors (int32 views of the u32 arrays of the
    FULL square A and of this row shard of C; values fp64).  Returns (ok, detail)."""
    n = a_ro.numel() - 1
    m = r1 - r0
    a_ro64 = a_ro.to(torch.int64) & 0xFFFFFFFF
    c_ro64 = c_ro.to(torch.int64) & 0xFFFFFFFF
    nnz_a, nnz_c = int(a_ro64[-1].item()), int(c_ro64[-1].item())
    d = {"rows": int(m), "nnz_c": nnz_c}
    if c_ro.numel() != m + 1 or int(c_ro64[0].item()) != 0 or nnz_c != c_col.numel():
        return False, dict(d, why="row_offsets inconsistent with nnz")
    if not bool((c_ro64[1:] >= c_ro64[:-1]).all().item()):
        return False, dict(d, why="row_offsets not monotone")
    cc = c_col.to(torch.int64) & 0xFFFFFFFF
    if nnz_c and not bool((cc < cols).all().item()):
        return False, dict(d, why="column id out of range")
    if nnz_c > 1:
        starts = torch.zeros(nnz_c + 1, dtype=torch.bool, device=c_ro.device)
        starts[c_ro64[:-1]] = True
        asc = (cc[1:] > cc[:-1]) | starts[1:nnz_c]
        if not bool(asc.all().item()):
            return False, dict(d, why="a row is not strictly ascending")
        del starts, asc
    if n != cols:
        return True, dict(d, note="row-sum identity skipped (rectangular)")
    # row sums: C*1 == A[r0:r1]*(A*1), scaled by |A|[r0:r1]*(|A|*1)
    ac = a_col.to(torch.int64) & 0xFFFFFFFF
    rid_a = _row_ids(torch, a_ro64, nnz_a)
    ones = torch.ones(cols, dtype=torch.float64, device=a_val.device)

    def spmv(vals, x, lo, hi):
        e0, e1 = int(a_ro64[lo].item()), int(a_ro64[hi].item())
        y = torch.zeros(hi - lo, dtype=torch.float64, device=vals.device)
        y.index_add_(0, rid_a[e0:e1] - lo, vals[e0:e1] * x[ac[e0:e1]])
        return y

    ref = spmv(a_val, spmv(a_val, ones, 0, n), r0, r1)
    aabs = a_val.abs()
    scale = spmv(aabs, spmv(aabs, ones, 0, n), r0, r1) + 1e-300
    rid_c = _row_ids(torch, c_ro64, nnz_c)
    csum = torch.zeros(m, dtype=torch.float64, device=a_val.device)
    csum.index_add_(0, rid_c, c_val)
    worst = float(((csum - ref).abs() / scale).max().item())
    d["row_sum_rel_err"] = worst
    return bool(worst < tol), d


def sampled_blocks(torch, A, c_ro, c_col, c_val, blocks=3, rows_per_block=20000, tol=TOL64, seed=0):
    """The oracle on `blocks` row blocks of A*A (first, last and random ones) against the same rows of C on the
    device.  Returns (ok, detail)."""
    H = _as_po(A)
    m = H.rows
    rpb = min(rows_per_block, m)
    rng = np.random.default_rng(seed)
    starts = [0, max(0, m - rpb)] + [int(x) for x in rng.integers(0, max(1, m - rpb), size=max(0, blocks - 2))]
    out = []
    for r0 in starts[:max(blocks, 1)]:
        r1 = min(m, r0 + rpb)
        R, ab = po.spgemm(H.row_slice(r0, r1), H)
        ro = (c_ro[r0:r1 + 1].to(torch.int64) & 0xFFFFFFFF).cpu().numpy()
        e0, e1 = int(ro[0]), int(ro[-1])
        col = c_col[e0:e1].cpu().numpy().view(np.uint32)
        val = c_val[e0:e1].cpu().numpy()
        ok = (e1 - e0 == R.nnz and ((ro - e0) == R.row_offsets).all() and (col == R.col_ids).all())
        worst = None
        if ok:
            err = np.abs(val - R.data)
            worst = float(np.max(err / (tol * ab + 1e-300))) if err.size else 0.0
            ok = worst <= 1.0
        out.append({"rows": [r0, r1], "ok": bool(ok), "max_err_over_bound": worst})
        if not ok:
            return False, {"blocks": out}
    return True, {"blocks": out}
