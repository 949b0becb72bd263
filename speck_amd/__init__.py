"""speck_amd -- MI355X-native SpGEMM (C = A*B, CSR in / CSR out) behind the spECK host API.

The product is the C-ABI shared library ``libspeck_amd.so`` (hand-written HIP for gfx950,
sources under ``speck_amd/csrc``); this package is only the thin ctypes mirror of the
reference's host interface used by the tests and the benchmark harness.
"""
from .api import (  # noqa: F401
    SpeckError, Timings, dCSR, spECKConfig, HostCSR, MultiplyspECK, BoundMultiply, analysis, symbolic,
    partition_rows, compare, compare_bounded, transpose, gen_matrix, load_matrix, load_mtx, store_mtx, load_hicsr,
    store_hicsr, lib_path,
)
