"""ctypes/numpy front-end of the CPU oracle (oracle/libspeck_oracle.so).

TEST INFRASTRUCTURE ONLY -- may be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg, never by the product package speck_amd.
PARITY UNPINNED BY THE REFERENCE (no golden vectors exist upstream); see
oracle/speck_oracle.h for what pins it instead.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


def _opt(ptr_type):
    """ndpointer that also accepts None."""
    base = ptr_type

    class _Opt(base):
        @classmethod
        def from_param(cls, obj):
            if obj is None:
                return None
            return base.from_param(obj)

    return _Opt


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libspeck_oracle.so"])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "libspeck_oracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    L.orc_gen_uniform.restype = C.c_uint64
    L.orc_gen_uniform.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int,
                                  _u32p, _u32p, _f64p]
    L.orc_analysis.restype = None
    L.orc_analysis.argtypes = [C.c_uint64, _u32p, _u32p, _u32p, _u32p, _u32p, _u32p, _u32p, _u32p,
                               C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    L.orc_symbolic.restype = C.c_uint64
    L.orc_symbolic.argtypes = [C.c_uint64, C.c_uint64, _u32p, _u32p, _u32p, _u32p, _u32p, C.c_int]
    L.orc_numeric.restype = None
    L.orc_numeric.argtypes = [C.c_uint64, C.c_uint64, _u32p, _u32p, _f64p, _u32p, _u32p, _f64p,
                              _u32p, _u32p, _f64p, _opt(_f64p), C.c_int]
    L.orc_numeric_f32.restype = None
    L.orc_numeric_f32.argtypes = [C.c_uint64, C.c_uint64, _u32p, _u32p, _f32p, _u32p, _u32p, _f32p,
                                  _u32p, _u32p, _f32p, _opt(_f32p), C.c_int]
    L.orc_exclusive_scan.restype = C.c_uint64
    L.orc_exclusive_scan.argtypes = [_u32p, C.c_uint64]
    L.orc_transpose.restype = None
    L.orc_transpose.argtypes = [C.c_uint64, C.c_uint64, _u32p, _u32p, _f64p, _u32p, _u32p, _f64p]
    L.orc_max_threads.restype = C.c_int
    _LIB = L
    return L


class HostCSR:
    """Host CSR with the reference's layout (include/CSR.h:57-61): u32 offsets/ids."""

    def __init__(self, rows, cols, row_offsets, col_ids, data):
        self.rows = int(rows)
        self.cols = int(cols)
        self.row_offsets = np.ascontiguousarray(row_offsets, dtype=np.uint32)
        self.col_ids = np.ascontiguousarray(col_ids, dtype=np.uint32)
        self.data = np.ascontiguousarray(data)
        assert self.row_offsets.shape == (self.rows + 1,)

    @property
    def nnz(self):
        return int(self.row_offsets[-1]) - int(self.row_offsets[0])

    def row_slice(self, r0, r1):
        """View of rows [r0, r1): absolute offsets into the shared col_ids/data."""
        return HostCSR(r1 - r0, self.cols, self.row_offsets[r0:r1 + 1].copy(), self.col_ids, self.data)

    def to_scipy(self):
        import scipy.sparse as sp
        base = int(self.row_offsets[0])
        return sp.csr_matrix((self.data[base:base + self.nnz], self.col_ids[base:base + self.nnz],
                              self.row_offsets.astype(np.int64) - base), shape=(self.rows, self.cols))

    @staticmethod
    def from_scipy(m):
        m = m.tocsr()
        m.sort_indices()
        return HostCSR(m.shape[0], m.shape[1], m.indptr.astype(np.uint32), m.indices.astype(np.uint32),
                       m.data.astype(np.float64))


def gen_uniform(n, seed=42, kmin=10, kspan=21, signed=False):
    ro = np.zeros(n + 1, dtype=np.uint32)
    cap = n * (kmin + kspan - 1)
    ci = np.zeros(cap, dtype=np.uint32)
    da = np.zeros(cap, dtype=np.float64)
    nnz = lib().orc_gen_uniform(n, seed, kmin, kspan, int(signed), ro, ci, da)
    return HostCSR(n, n, ro, ci[:nnz].copy(), da[:nnz].copy())


def analysis(A, B):
    m = A.rows
    ops = np.zeros(m, dtype=np.uint32)
    mx = np.zeros(m, dtype=np.uint32)
    cmin = np.zeros(m, dtype=np.uint32)
    cmax = np.zeros(m, dtype=np.uint32)
    P = C.c_uint64(0)
    M = C.c_uint32(0)
    lib().orc_analysis(m, A.row_offsets, A.col_ids, B.row_offsets, B.col_ids, ops, mx, cmin, cmax,
                       C.byref(P), C.byref(M))
    return dict(row_ops=ops, row_max_ops=mx, row_col_min=cmin, row_col_max=cmax,
                sum_products=int(P.value), max_row_ops=int(M.value))


def symbolic(A, B, threads=0):
    cnt = np.zeros(A.rows + 1, dtype=np.uint32)
    total = lib().orc_symbolic(A.rows, B.cols, A.row_offsets, A.col_ids, B.row_offsets, B.col_ids,
                               cnt, threads)
    return cnt, int(total)


def spgemm(A, B, threads=0, with_abs=True, out=None):
    """Full oracle SpGEMM. Returns (HostCSR C, abs_sum or None).  `out` = a previous result of the
    same product: its col_ids / data arrays are reused (the timed baseline must not measure the page
    faults of fresh output arrays -- the GPU path reuses C as well)."""
    cnt, total = symbolic(A, B, threads)
    if total > 0xFFFFFFFF:
        raise OverflowError("nnz(C) exceeds the u32 row_offsets of the dCSR layout")
    lib().orc_exclusive_scan(cnt, A.rows)
    if out is not None and out.col_ids.size == total and out.data.dtype == A.data.dtype and not with_abs:
        lib_fn = lib().orc_numeric_f32 if A.data.dtype == np.float32 else lib().orc_numeric
        lib_fn(A.rows, B.cols, A.row_offsets, A.col_ids, A.data, B.row_offsets, B.col_ids, B.data, cnt,
               out.col_ids, out.data, None, threads)
        return HostCSR(A.rows, B.cols, cnt, out.col_ids, out.data), None
    ci = np.zeros(total, dtype=np.uint32)
    if A.data.dtype == np.float32:
        da = np.zeros(total, dtype=np.float32)
        ab = np.zeros(total, dtype=np.float32) if with_abs else None
        lib().orc_numeric_f32(A.rows, B.cols, A.row_offsets, A.col_ids, A.data, B.row_offsets,
                              B.col_ids, B.data, cnt, ci, da, ab, threads)
    else:
        da = np.zeros(total, dtype=np.float64)
        ab = np.zeros(total, dtype=np.float64) if with_abs else None
        lib().orc_numeric(A.rows, B.cols, A.row_offsets, A.col_ids, A.data, B.row_offsets,
                          B.col_ids, B.data, cnt, ci, da, ab, threads)
    return HostCSR(A.rows, B.cols, cnt, ci, da), ab


def spgemm_f64_of(A, B, threads=0):
    """The product of fp32 inputs computed in fp64 (the inputs convert exactly): the reference value a float
    implementation is judged against -- |c - c_ref| <= k * eps32 * sum|a*b| -- without the error of the checker's
    own float accumulation.  fp64 inputs: spgemm itself."""
    if A.data.dtype == np.float64 and B.data.dtype == np.float64:
        return spgemm(A, B, threads=threads)
    A64 = HostCSR(A.rows, A.cols, A.row_offsets, A.col_ids, A.data.astype(np.float64))
    B64 = A64 if B is A else HostCSR(B.rows, B.cols, B.row_offsets, B.col_ids, B.data.astype(np.float64))
    return spgemm(A64, B64, threads=threads)


def transpose(A):
    tro = np.zeros(A.cols + 1, dtype=np.uint32)
    tci = np.zeros(A.nnz, dtype=np.uint32)
    tda = np.zeros(A.nnz, dtype=np.float64)
    lib().orc_transpose(A.rows, A.cols, A.row_offsets, A.col_ids, A.data, tro, tci, tda)
    return HostCSR(A.cols, A.rows, tro, tci, tda)


def max_threads():
    return int(lib().orc_max_threads())
