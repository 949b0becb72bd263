"""Host-side mirror of the reference's interface for the SpGEMM path, over the C ABI.

Names, argument meaning and error behaviour follow the reference:
  MultiplyspECK(A, B, matOut, config, timings)  -- include/Multiply.h:15-16
  dCSR / convert()                              -- include/dCSR.h, source/dCSR.cpp
  spECKConfig.initialize / cleanup              -- include/spECKConfig.h:15-43
  Timings (+=, /=)                              -- include/Timings.h:4-49
  CSR (host)                                    -- include/CSR.h:57-65
All compute goes through libspeck_amd.so; nothing here has a CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import CStats, CTimings, DCsr, NUM_NUM_BINS, NUM_SYM_BINS

SYM_CLASS_NAMES = ["g16", "wave256", "wave1k", "block4k", "block16k", "block32k", "bitmap256k", "bitmap1m",
                   "numeric_first", "global_hash", "g8", "wave128", "r32", "r64"]
NUM_CLASS_NAMES = ["direct", "g16", "wave128", "wave512", "block2k", "block8k", "dense4k", "dense16k", "global",
                   "wave256", "nfcopy", "g8", "r32", "r64"]


class SpeckError(RuntimeError):
    def __init__(self, status, where=""):
        self.status = status
        msg = _lib.load().speck_status_string(status).decode()
        super().__init__(f"{where}: {msg} (status {status})" if where else f"{msg} (status {status})")


def _check(status, where=""):
    if status != 0:
        raise SpeckError(status, where)


def lib_path():
    return _lib.LIB_PATH


class HostCSR:
    """Host CSR<T> (reference include/CSR.h): u32 row_offsets[rows+1], u32 col_ids, T data."""

    def __init__(self, rows, cols, row_offsets, col_ids, data):
        self.rows = int(rows)
        self.cols = int(cols)
        self.row_offsets = np.ascontiguousarray(row_offsets, dtype=np.uint32)
        self.col_ids = np.ascontiguousarray(col_ids, dtype=np.uint32)
        self.data = np.ascontiguousarray(data)
        if self.row_offsets.shape != (self.rows + 1,):
            raise ValueError("row_offsets must have rows+1 entries")

    @property
    def nnz(self):
        return int(self.row_offsets[-1]) - int(self.row_offsets[0])

    @staticmethod
    def _from_handle(h):
        L = _lib.load()
        r, c, n = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(L.speck_host_csr_dims(h, C.byref(r), C.byref(c), C.byref(n)))
        ro = np.empty(r.value + 1, dtype=np.uint32)
        ci = np.empty(n.value, dtype=np.uint32)
        da = np.empty(n.value, dtype=np.float64)
        _check(L.speck_host_csr_copy(h, ro.ctypes.data, ci.ctypes.data, da.ctypes.data))
        L.speck_host_csr_free(h)
        return HostCSR(r.value, c.value, ro, ci, da)

    def _to_handle(self):
        L = _lib.load()
        h = C.c_void_p()
        d = np.ascontiguousarray(self.data, dtype=np.float64)
        _check(L.speck_host_csr_from_arrays(self.rows, self.cols, self.nnz, self.row_offsets.ctypes.data,
                                            self.col_ids.ctypes.data, d.ctypes.data, C.byref(h)))
        return h


def gen_matrix(kind, scale=1.0, seed=42, signed=False):
    """Synthetic stand-ins of SURVEY.md 8d: uniform|scircuit|webbase|mac_econ|cant|nlpkkt."""
    h = C.c_void_p()
    _check(_lib.load().speck_gen_matrix(kind.encode(), float(scale), int(seed), int(bool(signed)),
                                        C.byref(h)), f"gen_matrix({kind})")
    return HostCSR._from_handle(h)


def load_mtx(path):
    h = C.c_void_p()
    _check(_lib.load().speck_load_mtx(str(path).encode(), C.byref(h)), f"load_mtx({path})")
    return HostCSR._from_handle(h)


def store_mtx(mat, path, symmetric_lower=False):
    """MatrixMarket coordinate real; symmetric_lower: only the lower triangle, banner `symmetric`."""
    h = mat._to_handle()
    try:
        _check(_lib.load().speck_store_mtx(h, str(path).encode(), int(bool(symmetric_lower))), f"store_mtx({path})")
    finally:
        _lib.load().speck_host_csr_free(h)


def load_hicsr(path):
    h = C.c_void_p()
    _check(_lib.load().speck_load_hicsr(str(path).encode(), C.byref(h)), f"load_hicsr({path})")
    return HostCSR._from_handle(h)


def store_hicsr(mat, path):
    h = mat._to_handle()
    try:
        _check(_lib.load().speck_store_hicsr(h, str(path).encode()), f"store_hicsr({path})")
    finally:
        _lib.load().speck_host_csr_free(h)


def load_matrix(path, write_cache=True):
    """DataLoader rule (source/DataLoader.cpp:24-58): '<path>d_.hicsr' cache, else .mtx."""
    h = C.c_void_p()
    _check(_lib.load().speck_load_matrix(str(path).encode(), int(write_cache), C.byref(h)),
           f"load_matrix({path})")
    return HostCSR._from_handle(h)


class Timings:
    """reference include/Timings.h:4-49 (milliseconds)."""
    FIELDS = ("init", "countProducts", "loadBalanceCounting", "globalMapsCounting", "spGEMMCounting",
              "allocC", "loadBalanceNumeric", "globalMapsNumeric", "spGEMMNumeric", "sorting",
              "cleanup", "complete")

    def __init__(self, measureAll=False, measureCompleteTime=False):
        self.measureAll = measureAll
        self.measureCompleteTime = measureCompleteTime
        for f in self.FIELDS:
            setattr(self, f, 0.0)

    def __iadd__(self, b):
        for f in self.FIELDS:
            setattr(self, f, getattr(self, f) + getattr(b, f))
        return self

    def __itruediv__(self, x):
        for f in self.FIELDS:
            setattr(self, f, getattr(self, f) / x)
        return self

    def _to_c(self):
        t = CTimings()
        t.measureAll = int(self.measureAll)
        t.measureCompleteTime = int(self.measureCompleteTime)
        return t

    def _from_c(self, t):
        for f in self.FIELDS:
            setattr(self, f, float(getattr(t, f)))


class dCSR:
    """Device CSR (reference include/dCSR.h:9-22).  Owns its buffers unless built as a view."""

    def __init__(self, dtype=np.float64):
        self._c = DCsr()
        self._owner = True
        self._keep = None
        self.dtype = np.dtype(dtype)
        self._host_row_offsets = None

    rows = property(lambda s: int(s._c.rows))
    cols = property(lambda s: int(s._c.cols))
    nnz = property(lambda s: int(s._c.nnz))

    def alloc(self, rows, cols, nnz, allocOffsets=True):
        _check(_lib.load().speck_dcsr_alloc(C.byref(self._c), rows, cols, nnz, int(allocOffsets),
                                            self.dtype.itemsize), "dCSR.alloc")

    def reset(self):
        if self._owner:
            _lib.load().speck_dcsr_free(C.byref(self._c))
        else:
            # in place: a BoundMultiply (or any other holder of byref(self._c)) keeps pointing at THIS struct
            C.memset(C.byref(self._c), 0, C.sizeof(DCsr))
        self._keep = None

    def __del__(self):
        try:
            self.reset()
        except Exception:
            pass

    # convert(dCSR <- CSR), source/dCSR.cpp:51-65
    @staticmethod
    def from_host(h):
        d = dCSR(h.data.dtype)
        base = int(h.row_offsets[0])
        ro = (h.row_offsets - np.uint32(base)).astype(np.uint32) if base else h.row_offsets
        ci = np.ascontiguousarray(h.col_ids[base:base + h.nnz])
        da = np.ascontiguousarray(h.data[base:base + h.nnz])
        _check(_lib.load().speck_dcsr_upload(C.byref(d._c), h.rows, h.cols, h.nnz, ro.ctypes.data,
                                             ci.ctypes.data, da.ctypes.data, d.dtype.itemsize),
               "convert(dCSR<-CSR)")
        d._host_row_offsets = np.array(ro, dtype=np.uint32)
        return d

    # convert(CSR <- dCSR), source/dCSR.cpp:67-76
    def to_host(self):
        ro = np.zeros(self.rows + 1, dtype=np.uint32)
        ci = np.zeros(self.nnz, dtype=np.uint32)
        da = np.zeros(self.nnz, dtype=self.dtype)
        if self._c.row_offsets:
            _check(_lib.load().speck_dcsr_download(C.byref(self._c), ro.ctypes.data, ci.ctypes.data,
                                                   da.ctypes.data, self.dtype.itemsize),
                   "convert(CSR<-dCSR)")
        return HostCSR(self.rows, self.cols, ro, ci, da)

    # convert(dCSR <- dCSR, padding), source/dCSR.cpp:81-89: device to device
    def copy(self, padding=0):
        d = dCSR(self.dtype)
        _check(_lib.load().speck_dcsr_copy(C.byref(d._c), C.byref(self._c), self.dtype.itemsize, int(padding)),
               "convert(dCSR<-dCSR)")
        if self._host_row_offsets is not None:
            d._host_row_offsets = (self._host_row_offsets - self._host_row_offsets[0]).astype(np.uint32)
        return d

    def row_view(self, r0, r1):
        """Non-owning view of rows [r0, r1): row_offsets stay absolute (shard of A)."""
        if self._host_row_offsets is None:
            raise ValueError("row_view needs the host row offsets (build with from_host/from_device)")
        v = dCSR(self.dtype)
        v._owner = False
        v._keep = self
        v._c.rows = r1 - r0
        v._c.cols = self._c.cols
        v._c.nnz = int(self._host_row_offsets[r1]) - int(self._host_row_offsets[r0])
        v._c.data = self._c.data
        v._c.col_ids = self._c.col_ids
        v._c.row_offsets = (self._c.row_offsets or 0) + 4 * r0
        v._host_row_offsets = self._host_row_offsets[r0:r1 + 1]
        return v

    @staticmethod
    def from_device(rows, cols, nnz, row_offsets_ptr, col_ids_ptr, data_ptr, dtype=np.float64,
                    keep=None, host_row_offsets=None):
        """Non-owning wrapper of caller-owned device buffers (e.g. torch tensors)."""
        v = dCSR(dtype)
        v._owner = False
        v._keep = keep
        v._c.rows, v._c.cols, v._c.nnz = rows, cols, nnz
        v._c.row_offsets, v._c.col_ids, v._c.data = row_offsets_ptr, col_ids_ptr, data_ptr
        v._host_row_offsets = host_row_offsets
        return v


class spECKConfig:
    """reference include/spECKConfig.h:8-53."""

    def __init__(self):
        raise TypeError("use spECKConfig.initialize(device)")  # private ctor in the reference

    @classmethod
    def initialize(cls, device=0):
        self = object.__new__(cls)
        self._h = C.c_void_p()
        _check(_lib.load().speck_config_create(int(device), C.byref(self._h)), "spECKConfig.initialize")
        sm, st, dy = C.c_int(), C.c_int(), C.c_int()
        _lib.load().speck_config_info(self._h, C.byref(sm), C.byref(st), C.byref(dy))
        self.sm = sm.value
        self.maxStaticSharedMemoryPerBlock = st.value
        self.maxDynamicSharedMemoryPerBlock = dy.value
        self.device = device
        return self

    def cleanup(self):
        if getattr(self, "_h", None):
            _lib.load().speck_config_destroy(self._h)
            self._h = None

    def set_stream(self, hip_stream_ptr):
        _check(_lib.load().speck_config_set_stream(self._h, hip_stream_ptr))

    def set_option(self, name, value):
        _check(_lib.load().speck_config_set_option(self._h, name.encode(), int(value)), f"set_option({name})")

    def profile_kernels(self, enable=True):
        _check(_lib.load().speck_config_profile_kernels(self._h, int(enable)))

    def last_stats(self):
        s = CStats()
        _check(_lib.load().speck_last_stats(self._h, C.byref(s)))
        return dict(
            sum_products=int(s.sum_products), nnz_c=int(s.nnz_c), max_row_ops=int(s.max_row_ops),
            max_row_nnz_c=int(s.max_row_nnz_c),
            sym_bin_rows=dict(zip(SYM_CLASS_NAMES, list(s.sym_bin_rows))),
            num_bin_rows=dict(zip(NUM_CLASS_NAMES, list(s.num_bin_rows))),
            sym_bin_bytes=dict(zip(SYM_CLASS_NAMES, list(s.sym_bin_bytes))),
            num_bin_bytes=dict(zip(NUM_CLASS_NAMES, list(s.num_bin_bytes))),
            sym_bin_ms=dict(zip(SYM_CLASS_NAMES, list(s.sym_bin_ms))),
            num_bin_ms=dict(zip(NUM_CLASS_NAMES, list(s.num_bin_ms))),
            analysis_ms=float(s.analysis_ms), scan_ms=float(s.scan_ms),
            sym_light_ms=float(s.sym_light_ms), num_light_ms=float(s.num_light_ms),
            sym_tiny_ms=float(s.sym_tiny_ms), num_tiny_ms=float(s.num_tiny_ms),
            kernel_events_valid=bool(s.kernel_events_valid), numeric_reruns=int(s.numeric_reruns),
            graph_replays=int(s.graph_replays), graph_captures=int(s.graph_captures),
            sym_phase_ms=float(s.sym_phase_ms), num_phase_ms=float(s.num_phase_ms),
            replayed=bool(s.replayed), nf_direct=bool(s.nf_direct), esc_fused=bool(s.esc_fused), pool_fallbacks=int(s.pool_fallbacks),
            scratch_pool_bytes=int(s.scratch_pool_bytes), pred_stages=int(s.pred_stages),
            eager_speculated=int(s.eager_speculated), one_walk=int(s.one_walk), walk_misses=int(s.walk_misses),
            eager_through=int(s.eager_through))


_NO_TIMINGS = CTimings()  # scratch for calls that do not ask for stage times


def MultiplyspECK(A, B, matOut, config, timings=None):
    """spECK::MultiplyspECK<T,...>(A, B, matOut, config, timings), include/Multiply.h:15-16."""
    L = _lib.load()
    t = timings._to_c() if timings is not None else _NO_TIMINGS
    if A.dtype != B.dtype:
        raise TypeError("A and B must share a value type")
    fn = L.speck_multiply_f64 if A.dtype == np.float64 else L.speck_multiply_f32
    if matOut.dtype != A.dtype:
        matOut.reset()
        matOut.dtype = A.dtype
    _check(fn(config._h, C.byref(A._c), C.byref(B._c), C.byref(matOut._c), C.byref(t)), "MultiplyspECK")
    if timings is not None:
        timings._from_c(t)
    return matOut


class BoundMultiply:
    """MultiplyspECK(A, B, matOut, config) with its arguments bound once: the benchmark loop of the reference calls
    the SAME multiply again and again (source/Executor.cpp:59-72), and at ~90 us per call the interpreter's share of a
    call -- attribute lookups, four byref objects, a status check through two frames -- is worth measuring out.
    Each __call__ is still exactly one speck_multiply_* call of the C ABI."""

    def __init__(self, A, B, matOut, config):
        if A.dtype != B.dtype:
            raise TypeError("A and B must share a value type")
        L = _lib.load()
        self._fn = L.speck_multiply_f64 if A.dtype == np.float64 else L.speck_multiply_f32
        if matOut.dtype != A.dtype:
            matOut.reset()
            matOut.dtype = A.dtype
        self._keep = (A, B, matOut, config)
        self._args = (config._h, C.byref(A._c), C.byref(B._c), C.byref(matOut._c), C.byref(_NO_TIMINGS))

    def __call__(self):
        rc = self._fn(*self._args)
        if rc:
            _check(rc, "MultiplyspECK")


def _dev_u32(n):
    """Scratch device array through the library's own allocator (a 1 x n dCSR col_ids buffer)."""
    d = dCSR()
    d.alloc(0, 0, max(int(n), 1), allocOffsets=False)
    return d


def analysis(A, B, config):
    """Stage entry point: the reference's readOperations quantities (include/common.cuh:321-459)."""
    L = _lib.load()
    m = A.rows
    bufs = [_dev_u32(m) for _ in range(4)]
    P, M = C.c_uint64(), C.c_uint32()
    _check(L.speck_analysis(config._h, C.byref(A._c), C.byref(B._c), *[b._c.col_ids for b in bufs],
                            C.byref(P), C.byref(M)), "analysis")
    out = {}
    for name, b in zip(("row_ops", "row_max_ops", "row_col_min", "row_col_max"), bufs):
        h = np.zeros(max(m, 1), dtype=np.uint32)
        b._c.nnz = max(m, 1)
        b._c.rows = 0
        _check(L.speck_dcsr_download(C.byref(b._c), None, h.ctypes.data, None, 8))
        out[name] = h[:m]
    out["sum_products"] = int(P.value)
    out["max_row_ops"] = int(M.value)
    return out


def symbolic(A, B, config):
    """Stage entry point: C.row_offsets (host copy) and nnz(C)."""
    L = _lib.load()
    buf = _dev_u32(A.rows + 1)
    n = C.c_uint64()
    _check(L.speck_symbolic(config._h, C.byref(A._c), C.byref(B._c), buf._c.col_ids, C.byref(n)), "symbolic")
    h = np.zeros(A.rows + 1, dtype=np.uint32)
    buf._c.nnz = A.rows + 1
    _check(L.speck_dcsr_download(C.byref(buf._c), None, h.ctypes.data, None, 8))
    return h, int(n.value)


def partition_rows(A, B, config, parts):
    b = (C.c_uint64 * (parts + 1))()
    _check(_lib.load().speck_partition_rows(config._h, C.byref(A._c), C.byref(B._c), parts, b), "partition_rows")
    return [int(x) for x in b]


def compare(ref, cmp, config, compare_data=False, rel_tol=1e-12):
    """spECK::Compare(reference_mat, compare_mat, compare_data) -> bool, include/Compare.h:5-6."""
    n = C.c_uint64()
    fn = _lib.load().speck_compare_f32 if ref.dtype == np.float32 else _lib.load().speck_compare_f64
    _check(fn(config._h, C.byref(ref._c), C.byref(cmp._c), int(compare_data), float(rel_tol), C.byref(n)), "Compare")
    return n.value == 0


def compare_bounded(ref, cmp, abs_products, config, tol=1e-12):
    """(rows differing in structure, rows with |ref - cmp| > tol * sum|a*b|); abs_products = |A|*|B|."""
    ns, nv = C.c_uint64(), C.c_uint64()
    _check(_lib.load().speck_compare_bounded_f64(config._h, C.byref(ref._c), C.byref(cmp._c),
                                                 C.byref(abs_products._c), float(tol), C.byref(ns), C.byref(nv)),
           "compare_bounded")
    return int(ns.value), int(nv.value)


def transpose(A, config):
    """spECK::Transpose(matIn, matTransposeOut), include/Transpose.h (float and double)."""
    At = dCSR(A.dtype)
    fn = _lib.load().speck_transpose_f32 if A.dtype == np.float32 else _lib.load().speck_transpose_f64
    _check(fn(config._h, C.byref(A._c), C.byref(At._c)), "Transpose")
    return At
