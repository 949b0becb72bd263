"""ctypes loader for libspeck_amd.so.  Fails loudly: there is no fallback path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SPECK_LIB: another build of the same library, e.g. the sanitizer build of `make ASAN=1`)
LIB_PATH = os.environ.get("SPECK_LIB") or os.path.join(_HERE, "libspeck_amd.so")

NUM_SYM_BINS = 16
NUM_NUM_BINS = 16


class DCsr(C.Structure):
    # field-for-field include/speck_c_api.h: speck_dcsr (reference dCSR<T>, include/dCSR.h:9-16)
    _fields_ = [("rows", C.c_uint64), ("cols", C.c_uint64), ("nnz", C.c_uint64),
                ("data", C.c_void_p), ("row_offsets", C.c_void_p), ("col_ids", C.c_void_p)]


class CTimings(C.Structure):
    _fields_ = [("measureAll", C.c_int32), ("measureCompleteTime", C.c_int32)] + [
        (n, C.c_float) for n in ("init", "countProducts", "loadBalanceCounting", "globalMapsCounting",
                                 "spGEMMCounting", "allocC", "loadBalanceNumeric",
                                 "globalMapsNumeric", "spGEMMNumeric", "sorting", "cleanup",
                                 "complete")]


class CStats(C.Structure):
    _fields_ = [("sum_products", C.c_uint64), ("nnz_c", C.c_uint64), ("max_row_ops", C.c_uint32),
                ("max_row_nnz_c", C.c_uint32),
                ("sym_bin_rows", C.c_uint32 * NUM_SYM_BINS), ("num_bin_rows", C.c_uint32 * NUM_NUM_BINS),
                ("num_bin_bytes", C.c_uint64 * NUM_NUM_BINS), ("sym_bin_bytes", C.c_uint64 * NUM_SYM_BINS),
                ("num_bin_ms", C.c_float * NUM_NUM_BINS), ("sym_bin_ms", C.c_float * NUM_SYM_BINS),
                ("analysis_ms", C.c_float), ("scan_ms", C.c_float),
                ("sym_light_ms", C.c_float), ("num_light_ms", C.c_float),
                ("sym_tiny_ms", C.c_float), ("num_tiny_ms", C.c_float),
                ("kernel_events_valid", C.c_int32), ("numeric_reruns", C.c_int32),
                ("graph_replays", C.c_int32), ("graph_captures", C.c_int32),
                ("sym_phase_ms", C.c_float), ("num_phase_ms", C.c_float),
                ("replayed", C.c_int32), ("nf_direct", C.c_int32), ("pool_fallbacks", C.c_int32),
                ("esc_fused", C.c_int32), ("scratch_pool_bytes", C.c_uint64),
                ("pred_stages", C.c_int32), ("eager_speculated", C.c_int32), ("one_walk", C.c_int32),
                ("walk_misses", C.c_int32), ("eager_through", C.c_int32)]


# every symbol include/speck_c_api.h declares, with its ctypes signature
_P = C.POINTER
_SIGS = {
    "speck_config_create": (C.c_int, [C.c_int, _P(C.c_void_p)]),
    "speck_config_destroy": (C.c_int, [C.c_void_p]),
    "speck_config_info": (C.c_int, [C.c_void_p, _P(C.c_int), _P(C.c_int), _P(C.c_int)]),
    "speck_config_handles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "speck_config_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "speck_config_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "speck_config_profile_kernels": (C.c_int, [C.c_void_p, C.c_int]),
    "speck_last_stats": (C.c_int, [C.c_void_p, _P(CStats)]),
    "speck_multiply_f64": (C.c_int, [C.c_void_p, _P(DCsr), _P(DCsr), _P(DCsr), _P(CTimings)]),
    "speck_multiply_f32": (C.c_int, [C.c_void_p, _P(DCsr), _P(DCsr), _P(DCsr), _P(CTimings)]),
    "speck_analysis": (C.c_int, [C.c_void_p, _P(DCsr), _P(DCsr), C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, _P(C.c_uint64), _P(C.c_uint32)]),
    "speck_symbolic": (C.c_int, [C.c_void_p, _P(DCsr), _P(DCsr), C.c_void_p, _P(C.c_uint64)]),
    "speck_partition_rows": (C.c_int, [C.c_void_p, _P(DCsr), _P(DCsr), C.c_int, _P(C.c_uint64)]),
    "speck_dcsr_alloc": (C.c_int, [_P(DCsr), C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_size_t]),
    "speck_dcsr_free": (C.c_int, [_P(DCsr)]),
    "speck_dcsr_upload": (C.c_int, [_P(DCsr), C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_size_t]),
    "speck_dcsr_upload_padded": (C.c_int, [_P(DCsr), C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32]),
    "speck_dcsr_copy": (C.c_int, [_P(DCsr), _P(DCsr), C.c_size_t, C.c_uint32]),
    "speck_dcsr_download": (C.c_int, [_P(DCsr), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "speck_dcsr_update": (C.c_int, [_P(DCsr), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "speck_compare_f64": (C.c_int, [C.c_void_p, _P(DCsr), _P(DCsr), C.c_int, C.c_double, _P(C.c_uint64)]),
    "speck_transpose_f64": (C.c_int, [C.c_void_p, _P(DCsr), _P(DCsr)]),
    "speck_transpose_f32": (C.c_int, [C.c_void_p, _P(DCsr), _P(DCsr)]),
    "speck_compare_f32": (C.c_int, [C.c_void_p, _P(DCsr), _P(DCsr), C.c_int, C.c_double, _P(C.c_uint64)]),
    "speck_compare_bounded_f64": (C.c_int, [C.c_void_p, _P(DCsr), _P(DCsr), _P(DCsr), C.c_double, _P(C.c_uint64),
                                            _P(C.c_uint64)]),
    "speck_gen_matrix": (C.c_int, [C.c_char_p, C.c_double, C.c_uint64, C.c_int, _P(C.c_void_p)]),
    "speck_load_mtx": (C.c_int, [C.c_char_p, _P(C.c_void_p)]),
    "speck_store_mtx": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "speck_load_hicsr": (C.c_int, [C.c_char_p, _P(C.c_void_p)]),
    "speck_store_hicsr": (C.c_int, [C.c_void_p, C.c_char_p]),
    "speck_load_matrix": (C.c_int, [C.c_char_p, C.c_int, _P(C.c_void_p)]),
    "speck_host_csr_dims": (C.c_int, [C.c_void_p, _P(C.c_uint64), _P(C.c_uint64), _P(C.c_uint64)]),
    "speck_host_csr_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "speck_host_csr_from_arrays": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p,
                                             C.c_void_p, _P(C.c_void_p)]),
    "speck_host_csr_free": (C.c_int, [C.c_void_p]),
    "speck_comm_unique_id": (C.c_int, [C.c_int, C.c_void_p]),
    "speck_comm_init": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, _P(C.c_void_p)]),
    "speck_comm_destroy": (C.c_int, [C.c_void_p]),
    "speck_comm_info": (C.c_int, [C.c_void_p, _P(C.c_int), _P(C.c_int), _P(C.c_int)]),
    "speck_gatherv_csr": (C.c_int, [C.c_void_p, C.c_int, _P(DCsr), C.c_uint64, C.c_size_t, _P(DCsr)]),
    "speck_gather_plan_create": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_size_t,
                                           C.c_int, _P(C.c_void_p)]),
    "speck_gather_start": (C.c_int, [C.c_void_p, C.c_int, _P(DCsr)]),
    "speck_gather_wait": (C.c_int, [C.c_void_p, C.c_int, _P(DCsr)]),
    "speck_gather_plan_layout": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "speck_gather_plan_destroy": (C.c_int, [C.c_void_p]),
    "speck_status_string": (C.c_char_p, [C.c_int]),
    "speck_version": (C.c_char_p, []),
}

_LIB = None


def load():
    """Load libspeck_amd.so (built in-tree by `make` / __graft_entry__.build())."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP extension first (`make` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError = a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def declared_symbols():
    return sorted(_SIGS)
