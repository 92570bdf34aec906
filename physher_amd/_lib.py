"""Loader for the in-tree C-ABI library (physher_amd/libphysher_amd.so).

There is no CPU fallback: if the HIP library is missing or does not load, importing the engine fails
loudly.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C physher_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PHYAMD_LIB: load another build of the same ABI (A/B experiments with kernel variants); default = the in-tree library
LIB_PATH = os.environ.get("PHYAMD_LIB") or os.path.join(_HERE, "libphysher_amd.so")

ABI_VERSION = 5

OK, EINVAL, EDEVICE, ENOMEM, EUNSUPPORTED = 0, -1, -2, -3, -4
RESCALE_NEVER, RESCALE_ALWAYS, RESCALE_AUTO = 0, 1, 2
GRAD_FOLD_ROOT_FREQS, GRAD_COMPAT_SCALED = 1, 2


class Config(C.Structure):
    _fields_ = [
        ("tip_count", C.c_int32), ("pattern_count", C.c_int32), ("state_count", C.c_int32),
        ("category_count", C.c_int32), ("device", C.c_int32), ("rescale", C.c_int32),
        ("max_device_bytes", C.c_int64), ("stream", C.c_void_p),
    ]


class Profile(C.Structure):
    _fields_ = [
        ("matrices_ms", C.c_double), ("lower_ms", C.c_double), ("upper_ms", C.c_double), ("reduce_ms", C.c_double),
        ("lower_launches", C.c_int32), ("upper_launches", C.c_int32), ("device_bytes", C.c_int64), ("tiles", C.c_int32),
    ]


# every symbol include/physher_amd.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("phyamd_create", C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    ("phyamd_create_sharded", C.c_int, [C.POINTER(Config), C.c_int32, _P, C.POINTER(_P)]),
    ("phyamd_shard_count", C.c_int, [_P]),
    ("phyamd_destroy", None, [_P]),
    ("phyamd_last_error", C.c_char_p, []),
    ("phyamd_abi_version", C.c_int, []),
    ("phyamd_set_tip_states", C.c_int, [_P, C.c_int, _P]),
    ("phyamd_set_tip_partials", C.c_int, [_P, C.c_int, _P]),
    ("phyamd_set_pattern_weights", C.c_int, [_P, _P]),
    ("phyamd_set_topology", C.c_int, [_P, _P, _P, C.c_int]),
    ("phyamd_set_branch_lengths", C.c_int, [_P, _P]),
    ("phyamd_set_branch_length", C.c_int, [_P, C.c_int, C.c_double]),
    ("phyamd_update_all_nodes", C.c_int, [_P]),
    ("phyamd_set_eigen", C.c_int, [_P, _P, _P, _P]),
    ("phyamd_set_frequencies", C.c_int, [_P, _P]),
    ("phyamd_set_category_rates", C.c_int, [_P, _P, _P]),
    ("phyamd_set_node_matrices", C.c_int, [_P, C.c_int, _P]),
    ("phyamd_set_matrices", C.c_int, [_P, _P]),
    ("phyamd_set_rate_matrix", C.c_int, [_P, _P]),
    ("phyamd_log_likelihood", C.c_int, [_P, C.POINTER(C.c_double)]),
    ("phyamd_gradient", C.c_int, [_P, C.c_int, C.POINTER(C.c_double), _P]),
    ("phyamd_branch_gradient", C.c_int, [_P, C.c_int, _P, C.POINTER(C.c_double), _P]),
    ("phyamd_gradient_device", C.c_int, [_P, C.c_int, _P]),
    ("phyamd_log_likelihood_device", C.c_int, [_P, _P]),
    ("phyamd_root_invariant_term", C.c_int, [_P, C.POINTER(C.c_double)]),
    ("phyamd_set_rate_matrix_derivatives", C.c_int, [_P, C.c_int, _P]),
    ("phyamd_parameter_gradient", C.c_int, [_P, C.c_int, C.POINTER(C.c_double), _P, _P]),
    ("phyamd_parameter_gradient_device", C.c_int, [_P, C.c_int, _P]),
    ("phyamd_root_frequency_term", C.c_int, [_P, _P]),
    ("phyamd_branch_log_likelihood", C.c_int, [_P, C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("phyamd_synchronize", C.c_int, [_P]),
    ("phyamd_get_pattern_log_likelihoods", C.c_int, [_P, _P]),
    ("phyamd_get_partials", C.c_int, [_P, C.c_int, C.c_int, _P]),
    ("phyamd_get_node_matrices", C.c_int, [_P, C.c_int, C.c_int, _P]),
    ("phyamd_is_rescaling", C.c_int, [_P]),
    ("phyamd_set_rescaling", C.c_int, [_P, C.c_int]),
    ("phyamd_set_keep_partials", C.c_int, [_P, C.c_int]),
    ("phyamd_set_reduction_levels", C.c_int, [_P, C.c_int]),
    ("phyamd_set_profiling", C.c_int, [_P, C.c_int]),
    ("phyamd_get_profile", C.c_int, [_P, C.POINTER(Profile)]),
    ("phyamd_store", C.c_int, [_P]),
    ("phyamd_restore", C.c_int, [_P]),
    ("phyamd_compress_patterns", C.c_int, [C.c_int, C.c_int32, C.c_int64, _P, _P, C.POINTER(C.c_int32), _P, _P]),
]

_lib = None


def load():
    """dlopen the engine library and bind every declared symbol (raises if anything is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build the HIP engine first (make -C physher_amd/csrc); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.phyamd_abi_version() != ABI_VERSION:
        raise ImportError(f"ABI mismatch: library {lib.phyamd_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


class EngineError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"physher_amd engine error {code}: {message}")
        self.code = code
