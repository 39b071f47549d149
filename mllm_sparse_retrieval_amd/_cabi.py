"""ctypes bindings of include/msr.h (libmsr.so). No torch, no pybind: plain pointers and sizes.

The library is the only scoring path: if it cannot be loaded, or no HIP device is usable, searching raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libmsr.so")

MSR_OK = 0
MSR_F_DROP_DF_EQ_N = 1
MSR_KMAX = 1024
MSR_COMM_ID_BYTES = 128

_ERR_NAMES = {
    -1: "MSR_E_INVAL", -2: "MSR_E_IO", -3: "MSR_E_FORMAT", -4: "MSR_E_NOMEM", -5: "MSR_E_NODEVICE",
    -6: "MSR_E_HIP", -7: "MSR_E_OVERFLOW", -8: "MSR_E_RANGE", -9: "MSR_E_COMM",
}


class MsrError(RuntimeError):
    def __init__(self, code, message):
        self.code = code
        super().__init__(f"{_ERR_NAMES.get(code, code)}: {message}")


class NoDeviceError(MsrError):
    """No usable HIP device / handle opened without one. There is deliberately no CPU fallback."""


class MsrInfo(C.Structure):
    _fields_ = [
        ("n_docs", C.c_uint64), ("n_postings", C.c_uint64), ("n_vecs", C.c_uint64),
        ("n_terms", C.c_uint32), ("tile_docs", C.c_uint32), ("n_tiles", C.c_uint32), ("max_weight", C.c_uint32),
        ("shard_tile0", C.c_uint32), ("shard_ntiles", C.c_uint32), ("device", C.c_int32), ("n_dense", C.c_uint32),
        ("term_lo", C.c_uint32), ("term_hi", C.c_uint32), ("resident_bytes", C.c_uint64),
    ]


# every symbol include/msr.h declares: (name, restype, argtypes)
_VP, _CP, _I, _U32, _U64 = C.c_void_p, C.c_char_p, C.c_int, C.c_uint32, C.c_uint64
SYMBOLS = [
    ("msr_set_build_option", _I, [_CP, C.c_double]),
    ("msr_index_build", _I, [_CP, _CP, _I, _U32]),
    ("msr_index_build_csr", _I, [_CP, _U64, _U32, _VP, _VP, _VP, _VP, _VP, _I, _U32]),
    ("msr_index_open", _I, [_CP, _I, C.POINTER(_VP)]),
    ("msr_index_open_shard", _I, [_CP, _I, _I, _I, C.POINTER(_VP)]),
    ("msr_index_open_termshard", _I, [_CP, _I, _I, _I, C.POINTER(_VP)]),
    ("msr_index_close", None, [_VP]),
    ("msr_index_info", _I, [_VP, C.POINTER(MsrInfo)]),
    ("msr_term_lookup", _I, [_VP, _VP, _I, _VP]),
    ("msr_term_df", _I, [_VP, _VP, _I, _VP]),
    ("msr_term_str", _I, [_VP, _U32, C.POINTER(_CP)]),
    ("msr_docid_str", _I, [_VP, _U32, C.POINTER(_CP)]),
    ("msr_search_csr", _I, [_VP, _VP, _VP, _VP, _I, _I, _U32, _VP, _VP, _VP, _VP]),
    ("msr_search_laps", _I, [_VP]),
    ("msr_search_text", _I, [_VP, _VP, _I, _I, _U32, _VP, _VP, _VP, _VP]),
    ("msr_encode_queries", _I, [_VP, _VP, _I, _VP, _VP, _VP, C.c_int64, C.POINTER(C.c_int64)]),
    ("msr_batch_create", _I, [_VP, _VP, _VP, _VP, _I, _I, _U32, C.POINTER(_VP)]),
    ("msr_batch_search", _I, [_VP, _I]),
    ("msr_batch_sync", _I, [_VP]),
    ("msr_batch_fetch", _I, [_VP, _VP, _VP, _VP, _VP]),
    ("msr_batch_kernel_ms", _I, [_VP, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("msr_batch_timing_reset", _I, [_VP]),
    ("msr_batch_timing_sum", _I, [_VP, C.POINTER(_I), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("msr_batch_debug_stamps", _I, [_VP, _VP]),
    ("msr_batch_algo_bytes", _I, [_VP, _I, C.POINTER(_U64), C.POINTER(_U64)]),
    ("msr_batch_work", _I, [_VP, _VP]),
    ("msr_batch_destroy", None, [_VP]),
    ("msr_comm_unique_id", _I, [_VP]),
    ("msr_comm_init", _I, [_VP, _I, _I, _VP]),
    ("msr_batch_search_sharded", _I, [_VP, _I]),
    ("msr_comm_destroy", _I, [_VP]),
    ("msr_batch_create_termshard", _I, [_VP, _VP, _VP, _VP, _I, _I, _U32, _I, _I, C.POINTER(_VP)]),
    ("msr_batch_search_termshard", _I, [_VP, _I]),
    ("msr_search_termshard_emulated", _I, [_VP, _VP, _VP, _VP, _I, _I, _U32, _I, _VP, _VP, _VP, _VP]),
    ("msr_search_termshard_emulated_handles", _I, [_VP, _I, _VP, _VP, _VP, _I, _I, _U32, _VP, _VP, _VP, _VP]),
    ("msr_comm_info", _I, [_VP, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    ("msr_runtime_info", _I, [_VP, _I]),
    ("msr_device_sync", _I, [_I]),
    ("msr_device_copy_gbs", _I, [_I, _U64, _I, C.POINTER(C.c_double)]),
    ("msr_device_peak_rates", _I, [_I, _VP]),
    ("msr_merge_lists", _I, [_VP, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    ("msr_dense_open", _I, [_VP, _U64, _U32, _I, C.POINTER(_VP)]),
    ("msr_dense_search", _I, [_VP, _VP, _I, _I, _VP, _VP, _VP, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("msr_dense_close", None, [_VP]),
    ("msr_dense_stats", _I, [_VP, _VP]),
    ("msr_f32_to_f16", _I, [_VP, _VP, _U64, _I]),
    ("msr_hybrid_search", _I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, C.c_float, _U32, _VP, _VP, _VP, _VP, _VP, _VP]),
    ("msr_sparsify", _I, [_VP, _I, _I, _I, _U32, _I, _I, _VP, _VP, _VP]),
    ("msr_synth_vectors", _I, [_U64, _U32, _U32, C.c_double, _U64, _I, _VP, _VP, _VP]),
    ("msr_last_error", _CP, []),
    ("msr_version", _CP, []),
]

_lib = None


def lib():
    """Load libmsr.so (built in-tree by __graft_entry__.build() / csrc/Makefile). Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C mllm_sparse_retrieval_amd/csrc`). There is no fallback scorer."
            )
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != MSR_OK:
        msg = lib().msr_last_error().decode("utf-8", "replace")
        raise (NoDeviceError if rc == -5 else MsrError)(rc, msg)


def ptr(a):
    return None if a is None else a.ctypes.data


def c_str_array(strings):
    """list[str|bytes] -> (ctypes array of char*, keep-alive list)."""
    bs = [s if isinstance(s, bytes) else s.encode("utf-8") for s in strings]
    arr = (C.c_char_p * len(bs))(*bs)
    return arr, bs


def as_csr(q_ptr, q_term, q_w):
    q_ptr = np.ascontiguousarray(q_ptr, dtype=np.int64)
    q_term = np.ascontiguousarray(q_term, dtype=np.int32)
    q_w = np.ascontiguousarray(q_w, dtype=np.int32)
    if q_ptr.ndim != 1 or len(q_ptr) < 1 or q_ptr[0] != 0 or q_ptr[-1] != len(q_term) or len(q_term) != len(q_w):
        raise ValueError("malformed CSR query arrays")
    return q_ptr, q_term, q_w
