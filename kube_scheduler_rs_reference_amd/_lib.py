"""ctypes binding of libksched_hip.so (the C ABI declared in include/ksched.h).

The product path has no CPU implementation: if the HIP library is missing this module raises at
import time of the symbol table (`load()`), and `ksched_create` returns KSCHED_E_NODEVICE on a box
without a GPU.  Nothing in this package imports `oracle/`.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# KSCHED_LIB: an alternative build of the SAME library (tools/build_variants.sh, A/B timing of kernel variants)
LIB_PATH = os.environ.get("KSCHED_LIB") or os.path.join(_PKG_DIR, "libksched_hip.so")

# --- constants mirrored from include/ksched.h --------------------------------------------------
ABI_VERSION = 6
MAX_KEYS = 32
MAX_ATTEMPTS = 64
SEL_NEVER = 0xFFFFFFFF

OK = 0
E_INVAL = -1
E_NODEVICE = -2
E_HIP = -3
E_NOMEM = -4
E_STATE = -5
E_UNSUPPORTED = -6
E_RCCL = -7
COMM_ID_BYTES = 128

FIT = 0x01
SEL = 0x02
TAINT = 0x04
PICK_SAMPLED = 0x08
PICK_BESTFIT = 0x10
WANT_FIT_MASK = 0x20

REASON_OK = 0
REASON_NOT_ENOUGH_RESOURCES = 1
REASON_NODE_SELECTOR_MISMATCH = 2
REASON_TAINT_NOT_TOLERATED = 3
REASON_NAMES = {0: "Ok", 1: "NotEnoughResources", 2: "NodeSelectorMismatch", 3: "TaintNotTolerated"}

OPT_KERNEL = 1
OPT_TIMING = 2
OPT_DEBUG = 3
OPT_TRACE = 4
OPT_PICK_FROM_MASK = 5
OPT_INDEX_BUILD = 6
OPT_BESTFIT_STAGES = 7
OPT_SNAPSHOT_STREAM = 8
OPT_FUSED_PICK = 9
OPT_FAULT = 10
OPT_PIPE_MODE = 11
PIPE_MAX_STREAMS = 8
OPT_GRID_CUS = 12
OPT_MASK_PROBE = 13
OPT_ROUND_ORDER = 14
TRACE_WORDS = 8
KERNEL_AUTO = 0
KERNEL_DIRECT = 1
KERNEL_FUSED = 3
MASK_ALLOC_AUTO = 0
MASK_ALLOC_PLAIN = 1
MASK_ALLOC_VMM = 2
MASK_ALLOC_VMM_MIN = 4
MASK_ALLOC_CONTIGUOUS = 5
MASK_ALLOC_SCATTER_2M = 8
MASK_ALLOC_SCATTER_16M = 9
MASK_ALLOC_PROBE = 11
MASK_ALLOC_LAST = 11
MASK_ALLOC_NAMES = {0: "auto", 1: "plain", 2: "vmm", 4: "vmm-min", 5: "contiguous", 8: "scatter-2m", 9: "scatter-16m", 11: "probe"}
MASK_PROBE_MIN_BYTES = 128 << 20
MASK_ALLOC_NAMES = {0: "auto", 1: "plain", 2: "vmm", 3: "vmm-1g", 4: "vmm-min", 5: "contiguous", 6: "uncached", 7: "pool", 8: "scatter-2m", 9: "scatter-16m", 10: "scatter-64k", 11: "probe"}

# every symbol include/ksched.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
_u32 = C.c_uint32
SYMBOLS = {
    "ksched_create": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "ksched_destroy": (None, [_vp]),
    "ksched_abi_version": (_u32, []),
    "ksched_device_count": (C.c_int, []),
    "ksched_strerror": (C.c_char_p, [C.c_int]),
    "ksched_last_error": (C.c_char_p, [_vp]),
    "ksched_mask_words": (_u32, [_u32]),
    "ksched_set_option": (C.c_int, [_vp, C.c_int, C.c_int64]),
    "ksched_set_nodes": (C.c_int, [_vp, _u32, _vp, _vp, _vp, _u32, _vp]),
    "ksched_update_nodes": (C.c_int, [_vp, _u32, _vp, _vp, _vp]),
    "ksched_forget_stream": (C.c_int, [_vp, _vp]),
    "ksched_num_nodes": (_u32, [_vp]),
    "ksched_num_keys": (_u32, [_vp]),
    "ksched_eval": (C.c_int, [_vp, _u32, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp]),
    "ksched_shard_bounds": (None, [_u32, _u32, _u32, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32)]),
    "ksched_eval_begin": (C.c_int, [_vp, _u32, _vp, _vp, _vp, _u32, _vp, _vp, _u32, _u32, _vp, _vp, _u32, C.POINTER(_vp), C.POINTER(_vp)]),
    "ksched_gather_buffer": (C.c_int, [_vp, _u32, C.POINTER(_vp)]),
    "ksched_eval_end": (C.c_int, [_vp, _vp, _u32, _vp]),
    "ksched_eval_device": (C.c_int, [_vp, _u32, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp]),
    "ksched_eval_device_pitched": (C.c_int, [_vp, _u32, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _u32, _vp]),
    "ksched_mask_pitch": (_u32, [_u32]),
    "ksched_mask_alloc": (C.c_int, [_vp, _u32, _u32, C.POINTER(_vp), C.POINTER(_u32)]),
    "ksched_mask_free": (C.c_int, [_vp, _vp]),
    "ksched_mask_probe_report": (C.c_int, [_vp, _vp, _u32]),
    "ksched_pick_device": (C.c_int, [_vp, _u32, _vp, _u32, _vp, _vp, _u32, _u32, _vp, _vp]),
    "ksched_pick": (C.c_int, [_vp, _u32, _vp, _vp, _vp, _u32, _u32, _vp]),
    "ksched_pipe_create": (C.c_int, [_vp, _u32, C.POINTER(_vp)]),
    "ksched_pipe_destroy": (None, [_vp]),
    "ksched_pipe_submit": (C.c_int, [_vp, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _u32, _vp]),
    "ksched_pipe_wait": (C.c_int, [_vp, _u32, _vp]),
    "ksched_pipe_wait_mask": (C.c_int, [_vp, _u32, _vp]),
    "ksched_pipe_stream": (_vp, [_vp, C.c_int]),
    "ksched_pipe_slot_stream": (_vp, [_vp, _u32]),
    "ksched_reason": (C.c_int, [_vp, _vp, _u32, _u32]),
    "ksched_comm_unique_id": (C.c_int, [_vp]),
    "ksched_comm_create": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "ksched_comm_create_local": (C.c_int, [_vp, C.c_int, _vp]),
    "ksched_comm_destroy": (None, [_vp]),
    "ksched_comm_rank": (C.c_int, [_vp]),
    "ksched_comm_size": (C.c_int, [_vp]),
    "ksched_allgather_bindings": (C.c_int, [_vp, _vp, _vp, _u32, _vp]),
    "ksched_allgather_bindings_local": (C.c_int, [_vp, C.c_int, _vp, _vp, _u32, _vp]),
    "ksched_comm_last_error": (C.c_char_p, []),
    "ksched_kernel_time_ms": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "ksched_kernel_time_samples": (C.c_int, [_vp, _vp, _u32]),
    "ksched_explain": (C.c_int, [_vp, _u32, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _u32, _vp]),
    "ksched_index_checksum": (C.c_int, [_vp, _vp]),
    "ksched_trace_read": (C.c_int, [_vp, _vp, C.c_uint32]),
    "ksched_last_kernel": (C.c_char_p, [_vp]),
    "ksched_last_pick": (C.c_char_p, [_vp]),
}

_lib = None


class KschedError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str = ""):
        self.code = code
        msg = f"{where}: ksched error {code}"
        try:
            msg += f" ({load().ksched_strerror(code).decode()})"
        except Exception:  # pragma: no cover
            pass
        if detail:
            msg += f": {detail}"
        super().__init__(msg)


def load() -> C.CDLL:
    """Load libksched_hip.so and bind every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension has not been built (run `make lib` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    # (the TEST build -- tests/cpp/hooks/libksched_hip.so, reached through $KSCHED_LIB -- exports two more functions; the shipped library does not)
    if hasattr(lib, "ksched_test_hooks_linked"):
        lib.ksched_test_hooks_linked.restype = C.c_int
        lib.ksched_test_hooks_linked.argtypes = []
    if lib.ksched_abi_version() != ABI_VERSION:
        raise ImportError(f"ABI mismatch: library {lib.ksched_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib
