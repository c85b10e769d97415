"""ctypes binding of libmfar_hip.so (the C ABI in include/mfar_hip.h).

The product path has NO CPU fallback: if the library is missing or a call fails, this module raises.
"""
import ctypes
import os
import subprocess

# The pipelined searcher keeps the scans, the tail kernels and (multi-GPU) RCCL's collectives on separate HIP streams.  The
# ROCm runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): when the scan stream shares a queue
# with RCCL's internal stream, the next launch's scan sits behind the all-gather that waits for the previous launch's tail,
# and the two streams run strictly one after the other (measured at the 8-GPU shard size: 107 k -> 130 k queries/s once
# every stream has its own queue).  Must be set before the HIP runtime initialises, hence at import; an explicit setting wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # .../multifield-adaptive-retrieval_amd
LIB_PATH = os.environ.get("MFAR_LIB_PATH") or os.path.join(_PKG, "lib", "libmfar_hip.so")   # (override: experiment builds)
CSRC = os.path.join(_PKG, "csrc")

MFAR_OK = 0
ERR_NAMES = {-1: "MFAR_ERR_INVALID", -2: "MFAR_ERR_HIP", -3: "MFAR_ERR_NOMEM", -4: "MFAR_ERR_UNSUPPORTED"}
DTYPE_F32, DTYPE_BF16 = 0, 1
MAX_K, MAX_FIELDS = 128, 32


class MfarError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 with hipcc (works without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["make", "-C", CSRC, "-s", "all"])
    return LIB_PATH


_lib = None

_c = ctypes
_vp, _i, _i64, _f32p = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_void_p

# name -> (restype, argtypes); must list every symbol declared in include/mfar_hip.h
SIGNATURES = {
    "mfar_version": (_i, []),
    "mfar_last_error": (_c.c_char_p, []),
    "mfar_device_count": (_i, [_c.POINTER(_i)]),
    "mfar_index_create": (_i, [_c.POINTER(_vp), _i, _i64, _i64, _i, _i, _i]),
    "mfar_index_destroy": (None, [_vp]),
    "mfar_index_info": (_i, [_vp, _c.POINTER(_i64), _c.POINTER(_i64), _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i64)]),
    "mfar_index_resident_bytes": (_i, [_vp, _c.POINTER(_i64), _c.POINTER(_i64), _c.POINTER(_i64), _c.POINTER(_i64), _c.POINTER(_i64)]),
    "mfar_index_write_rows": (_i, [_vp, _i, _i64, _i64, _vp, _i, _vp]),
    "mfar_index_read_rows": (_i, [_vp, _i, _i64, _i64, _vp, _i, _vp]),
    "mfar_retrieve_fields": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "mfar_retrieve_field": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "mfar_score_candidates": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp]),
    "mfar_mix_topk": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "mfar_search_two_stage": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "mfar_search_fused": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp]),
    "mfar_payload_bytes": (_i64, [_i, _i, _i]),
    "mfar_search_stage2": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "mfar_search_stage2_masks": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "mfar_search_local": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _vp]),
    "mfar_merge_workspace_bytes": (_i64, [_i, _i, _i]),
    "mfar_merge_payloads": (_i, [_i, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i64, _i, _vp]),
    "mfar_lists_bytes": (_i64, [_i, _i, _i]),
    "mfar_topk_bytes": (_i64, [_i, _i]),
    "mfar_retrieve_lists": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "mfar_search_owned": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mfar_search_owned_masks": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mfar_merge_topk": (_i, [_i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "mfar_stream_wait_stage1_start": (_i, [_vp, _vp]),
    "mfar_set_timing": (_i, [_vp, _i]),
    "mfar_stage1_timing": (_i, [_vp, _c.POINTER(_c.c_double), _c.POINTER(_i)]),
    "mfar_last_stage1_kernel": (_c.c_char_p, [_vp]),
    "mfar_set_wgs_per_cu": (_i, [_vp, _i]),
    "mfar_stage1_begin": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mfar_stage1_finish": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "mfar_max_split_batch": (_i, [_vp, _i]),
    "mfar_set_wide": (_i, [_vp, _i]),
    "mfar_set_repair_mode": (_i, [_vp, _i]),
    "mfar_set_screen": (_i, [_vp, _i, _c.c_float]),
    "mfar_get_screen": (_i, [_vp, _c.POINTER(_i), _c.POINTER(_c.c_float)]),
    "mfar_screen_field_info": (_i, [_vp, _i, _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int64)]),
    "mfar_screen_stats": (_i, [_vp, _c.POINTER(_i), _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int64)]),
    "mfar_set_stage2_mode": (_i, [_vp, _i]),
    "mfar_set_row_mode": (_i, [_vp, _i]),
    "mfar_row_mode_activate": (_i, [_vp]),
    "mfar_set_auto_off": (_i, [_vp, _i, _i, _i]),
    "mfar_debug_fail_allocations_above": (_i, [_i64]),
    "mfar_set_tier2": (_i, [_vp, _i]),
    "mfar_set_stage2_kernels": (_i, [_vp, _i]),
    "mfar_set_deep_scan": (_i, [_vp, _i]),
    "mfar_deep_scan_info": (_i, [_vp, _c.POINTER(_c.c_uint32), _c.POINTER(_i64)]),
    "mfar_tier2_stats": (_i, [_vp, _c.POINTER(_i), _c.POINTER(_i64), _c.POINTER(_i64), _c.POINTER(_i64)]),
    "mfar_tier2_rescan_stats": (_i, [_vp, _c.POINTER(_i64), _c.POINTER(_i64)]),
    "mfar_auto_off_info": (_i, [_vp, _c.POINTER(_c.c_uint32), _c.POINTER(_i64), _c.POINTER(_i64), _c.POINTER(_i64), _c.POINTER(_i)]),
    "mfar_row_mode_info": (_i, [_vp, _c.POINTER(_c.c_uint32), _c.POINTER(_c.c_uint32)]),
    "mfar_set_stage2_dump": (_i, [_vp, _i]),
    "mfar_stage2_dump_info": (_i, [_vp, _i, _c.POINTER(_i), _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int64)]),
    "mfar_stage2_stats": (_i, [_vp, _c.POINTER(_i), _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int64)]),
    "mfar_pipeline_create": (_i, [_c.POINTER(_vp), _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i]),
    "mfar_pipeline_destroy": (None, [_vp]),
    "mfar_pipeline_info": (_i, [_vp, _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i64)]),
    "mfar_pipeline_set_weights": (_i, [_vp, _vp, _vp, _i]),
    "mfar_pipeline_submit": (_i, [_vp, _vp, _i, _i, _vp, _c.POINTER(_i64)]),
    "mfar_pipeline_flush": (_i, [_vp]),
    "mfar_pipeline_result": (_i, [_vp, _i64, _vp, _vp, _vp, _i, _vp]),
    "mfar_pipeline_lists": (_i, [_vp, _i64, _vp, _vp, _i, _vp]),
    "mfar_pipeline_result_view": (_i, [_vp, _i64, _c.POINTER(_vp), _c.POINTER(_vp), _c.POINTER(_vp), _c.POINTER(_vp), _c.POINTER(_vp)]),
}

# MFAR_ABI_VERSION of include/mfar_hip.h these signatures were written against.  A library that reports another value has
# different argument lists behind the same names (pointers would land in the wrong slots): lib() refuses it.
ABI_VERSION = 107


def lib():
    """Load libmfar_hip.so. Raises (loudly) when it has not been built: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP extension first "
                f"(python -c 'import __graft_entry__ as g; g.build()' or make -C {CSRC})")
        # One HIP runtime per process: the PyTorch-ROCm wheel bundles its own libamdhip64/libhsa-runtime64 and a
        # second runtime initialised later in the same process finds no GPU.  Importing torch first makes the loader
        # resolve this library's libamdhip64.so.7 dependency to the copy torch already mapped (same SONAME), so device
        # pointers and streams are shared with torch.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        got = L.mfar_version()
        if got != ABI_VERSION:
            raise ImportError(f"{LIB_PATH} reports ABI version {got}, this package binds version {ABI_VERSION}: "
                              f"rebuild the library (make -C {CSRC})")
        _lib = L
    return _lib


def check(rc: int):
    if rc != MFAR_OK:
        raise MfarError(rc, lib().mfar_last_error().decode("utf-8", "replace"))


def device_count() -> int:
    n = _i(0)
    rc = lib().mfar_device_count(ctypes.byref(n))
    if rc != MFAR_OK:
        return 0
    return n.value
