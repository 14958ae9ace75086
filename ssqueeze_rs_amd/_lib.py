"""ctypes binding of libssq_hip.so (C-ABI declared in include/ssq_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails, the
caller gets an exception.  The product path never computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SSQ_HIP_LIB") or os.path.join(_HERE, "libssq_hip.so")   # env: diagnostic builds only
HEADER_PATH = os.path.join(_HERE, "..", "include", "ssq_hip.h")

SSQ_F32, SSQ_F64 = 0, 1
PAD = {"reflect": 0, "zero": 1}
SQUEEZE = {"sum": 0, "lebesgue": 1}
WAVELET = {"gmw": 0, "morlet": 1}
OUT_TX, OUT_SX, OUT_DSX, OUT_WK = 0, 1, 2, 3


class SsqHipError(RuntimeError):
    """A libssq_hip call returned a non-zero status."""


_lock = threading.Lock()
_lib = None

i64 = C.c_int64
vp = C.c_void_p
dp = C.POINTER(C.c_double)

# name -> (restype, argtypes); mirrors include/ssq_hip.h one to one
_SIGNATURES = {
    "ssq_last_error": (C.c_char_p, []),
    "ssq_hello_from_bin": (C.c_char_p, []),
    "ssq_build_has_tuning": (C.c_int, []),
    "ssq_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "ssq_set_device": (C.c_int, [C.c_int]),
    "ssq_device_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(i64), C.c_char_p, C.c_int]),
    "ssq_stft_shape": (C.c_int, [i64, i64, i64, C.POINTER(i64), C.POINTER(i64)]),
    "ssq_cwt_pad_len": (C.c_int, [i64, C.POINTER(i64), C.POINTER(i64)]),
    "ssq_log_scales": (C.c_int, [i64, i64, C.c_int, C.POINTER(i64), vp]),
    "ssq_size_window": (C.c_int, [vp, i64, i64, vp]),
    "ssq_diff_window": (C.c_int, [vp, i64, vp]),
    "ssq_cwt_ssq_freqs": (C.c_int, [vp, i64, i64, C.c_double, C.c_int, C.c_int, vp]),
    "ssq_stft_host": (C.c_int, [C.c_int, vp, i64, i64, vp, i64, i64, C.c_int, vp, vp]),
    "ssq_ssq_stft_host": (C.c_int, [C.c_int, vp, i64, i64, vp, i64, i64, C.c_double, C.c_int, C.c_int,
                                    C.c_double, vp, vp, vp, vp, vp]),
    "ssq_cwt_host": (C.c_int, [C.c_int, vp, i64, i64, C.c_int, vp, i64, C.c_double, C.c_int, C.c_int,
                               C.c_int, vp, vp]),
    "ssq_ssq_cwt_host": (C.c_int, [C.c_int, vp, i64, i64, C.c_int, vp, i64, C.c_double, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_double, vp, vp, vp, vp, vp]),
    "ssq_icwt_host": (C.c_int, [C.c_int, vp, i64, i64, C.c_int, vp, i64, C.c_int, i64, C.c_double, C.c_int, vp]),
    "ssq_morlet": (C.c_int, [vp, i64, C.c_double, vp]),
    "ssq_morlet_freq": (C.c_int, [i64, C.c_double, C.c_double, vp]),
    "ssq_morlet_time": (C.c_int, [i64, C.c_double, C.c_double, vp]),
    "ssq_gmw": (C.c_int, [vp, i64, C.c_double, C.c_double, C.c_char_p, C.c_int, vp]),
    "ssq_gmw_freq": (C.c_int, [i64, C.c_double, C.c_double, C.c_double, C.c_char_p, C.c_int, vp]),
    "ssq_gmw_time": (C.c_int, [i64, C.c_double, C.c_double, C.c_double, C.c_char_p, C.c_int, vp]),
    "ssq_gmw_center_frequency": (C.c_int, [C.c_double, C.c_double, C.c_char_p, C.POINTER(C.c_double)]),
    "ssq_stft_plan_create": (C.c_int, [C.POINTER(vp), C.c_int, i64, vp, i64, i64, C.c_double, C.c_int,
                                       C.c_int, C.c_double, C.c_int]),
    "ssq_stft_plan_create_v": (C.c_int, [C.POINTER(vp), C.c_int, i64, vp, i64, i64, C.c_double, C.c_int,
                                         C.c_int, C.c_double, C.c_int, C.c_int]),
    "ssq_stft_host_v": (C.c_int, [C.c_int, vp, i64, i64, vp, i64, i64, C.c_double, C.c_int, C.c_int, vp, vp]),
    "ssq_ssq_stft_host_v": (C.c_int, [C.c_int, vp, i64, i64, vp, i64, i64, C.c_double, C.c_int, C.c_int, C.c_double,
                                      C.c_int, vp, vp, vp, vp, vp]),
    "ssq_istft_host": (C.c_int, [C.c_int, vp, i64, vp, i64, i64, i64, C.c_int, C.c_int, vp]),
    "ssq_issq_host": (C.c_int, [C.c_int, vp, i64, i64, C.c_double, vp, vp]),
    "ssq_upstream_adm": (C.c_int, [C.c_int, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double)]),
    "ssq_upstream_center_frequency": (C.c_int, [C.c_int, C.c_double, C.c_double, C.c_double, i64, C.POINTER(C.c_double)]),
    "ssq_upstream_p2up": (C.c_int, [i64, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]),
    "ssq_cwt_host_v": (C.c_int, [C.c_int, vp, i64, i64, C.c_int, C.c_double, C.c_double, vp, i64, C.c_double, C.c_int,
                                 C.c_int, C.c_int, C.c_int, vp, vp]),
    "ssq_ssq_cwt_host_v": (C.c_int, [C.c_int, vp, i64, i64, C.c_int, C.c_double, C.c_double, vp, i64, C.c_double, C.c_int,
                                     vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, vp, vp, vp, vp]),
    "ssq_cwt_plan_create_v": (C.c_int, [C.POINTER(vp), C.c_int, i64, C.c_int, C.c_double, C.c_double, vp, i64,
                                        C.c_double, C.c_int, C.c_int]),
    "ssq_stft_plan_destroy": (C.c_int, [vp]),
    "ssq_stft_plan_is_fused": (C.c_int, [vp]),
    "ssq_stft_plan_workspace_bytes": (i64, [vp, i64, C.c_int]),
    "ssq_stft_plan_exec": (C.c_int, [vp, C.c_int, vp, i64, vp, vp, i64, vp]),
    "ssq_stft_plan_exec_strided": (C.c_int, [vp, C.c_int, vp, i64, i64, i64, i64, vp, vp, i64, vp]),
    "ssq_chunk_halo_fill": (C.c_int, [C.c_int, vp, i64, i64, i64, C.c_int, vp]),
    "ssq_chunks_relayout": (C.c_int, [C.c_int, vp, i64, i64, i64, i64, i64, i64, vp, i64, i64, i64, i64, vp]),
    "ssq_cwt_plan_create": (C.c_int, [C.POINTER(vp), C.c_int, i64, C.c_int, vp, i64, C.c_double, C.c_int]),
    "ssq_cwt_plan_destroy": (C.c_int, [vp]),
    "ssq_cwt_plan_workspace_bytes": (i64, [vp, i64]),
    "ssq_cwt_plan_exec_cwt": (C.c_int, [vp, vp, i64, C.c_int, C.c_int, vp, vp, vp, i64, vp]),
    "ssq_cwt_plan_exec_ssq": (C.c_int, [vp, vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                        vp, vp, vp, vp, vp, i64, vp]),
    "ssq_rccl_available": (C.c_int, []),
    "ssq_rccl_unique_id": (C.c_int, [vp]),
    "ssq_rccl_comm_init": (C.c_int, [C.POINTER(vp), C.c_int, vp, C.c_int]),
    "ssq_rccl_comm_info": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ssq_rccl_comm_destroy": (C.c_int, [vp]),
    "ssq_gather_shards": (C.c_int, [vp, vp, vp, i64, vp]),
    "ssq_pinned_alloc": (C.c_int, [C.POINTER(vp), i64]),
    "ssq_pinned_free": (C.c_int, [vp]),
    "ssq_host_cache_limit": (C.c_int, [i64]),
    "ssq_host_cache_stats": (C.c_int, [C.POINTER(i64), C.POINTER(i64)]),
    "ssq_host_cache_clear": (C.c_int, []),
    "ssq_dev_malloc": (C.c_int, [C.POINTER(vp), i64]),
    "ssq_dev_free": (C.c_int, [vp]),
    "ssq_dev_memset": (C.c_int, [vp, C.c_int, i64, vp]),
    "ssq_memcpy_h2d": (C.c_int, [vp, vp, i64, vp]),
    "ssq_memcpy_d2h": (C.c_int, [vp, vp, i64, vp]),
    "ssq_memcpy_d2d": (C.c_int, [vp, vp, i64, vp]),
    "ssq_stream_create": (C.c_int, [C.POINTER(vp)]),
    "ssq_stream_destroy": (C.c_int, [vp]),
    "ssq_stream_sync": (C.c_int, [vp]),
    "ssq_device_sync": (C.c_int, []),
    "ssq_event_create": (C.c_int, [C.POINTER(vp)]),
    "ssq_event_destroy": (C.c_int, [vp]),
    "ssq_event_record": (C.c_int, [vp, vp]),
    "ssq_event_sync": (C.c_int, [vp]),
    "ssq_event_elapsed_ms": (C.c_int, [vp, vp, C.POINTER(C.c_float)]),
    "ssq_graph_capture_begin": (C.c_int, [vp]),
    "ssq_graph_capture_end": (C.c_int, [vp, C.POINTER(vp)]),
    "ssq_graph_launch": (C.c_int, [vp, vp]),
    "ssq_graph_destroy": (C.c_int, [vp]),
}


def header_symbols() -> list:
    """Every function name include/ssq_hip.h declares."""
    with open(HEADER_PATH) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ssq_[a-z0-9_]+)\s*\(", src)))


def load():
    """Load libssq_hip.so (once) and set the ctypes prototypes.  Raises ImportError if absent."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m ssqueeze_rs_amd.build` "
                "(hipcc, gfx950).  ssqueeze_rs_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(rc: int, exc=SsqHipError):
    if rc != 0:
        msg = load().ssq_last_error().decode("utf-8", "replace")
        raise exc(msg)


def device_count() -> int:
    n = C.c_int(0)
    rc = load().ssq_device_count(C.byref(n))
    return n.value if rc == 0 else 0


class _PinnedBlock:
    """Owner of one block of the library's pinned pool; returns it when the NumPy array on top of it dies."""

    def __init__(self, nbytes: int):
        self.ptr = C.c_void_p()
        check(load().ssq_pinned_alloc(C.byref(self.ptr), int(nbytes)))

    def __del__(self):
        try:
            if self.ptr:
                _lib.ssq_pinned_free(self.ptr)
        except Exception:
            pass
        self.ptr = None


# Results above this size come from pageable memory: page-locking tens of GB (C5: 17.2 GB of Tx per fp64 signal) can
# fail or stall the host.  Override with the environment (bytes); 0 switches the pinned pool off for results.
PINNED_RESULT_LIMIT = int(os.environ.get("SSQ_PINNED_RESULT_LIMIT", str(4 << 30)))


def pinned_empty(shape, dtype):
    """`np.empty(shape, dtype)` in pinned host memory from the library's pool: device results land in it by DMA, no
    page faults, no staging copy.  An ordinary writable ndarray for the caller; the block goes back to the pool when
    the array (and every view of it) is garbage-collected.  Falls back to np.empty (pageable) when no GPU is visible,
    when the array is larger than PINNED_RESULT_LIMIT, or when the pinned allocation fails -- the drop-in call must
    not fail where the reference's `np.empty` would succeed."""
    import numpy as np
    dt = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
    if n == 0 or n > PINNED_RESULT_LIMIT or device_count() < 1:
        return np.empty(shape, dtype=dt)
    try:
        blk = _PinnedBlock(n)
    except SsqHipError:
        return np.empty(shape, dtype=dt)
    raw = (C.c_char * n).from_address(blk.ptr.value)
    raw._ssq_owner = blk                    # the ctypes array is the ndarray's base: it keeps the block alive
    return np.frombuffer(raw, dtype=dt).reshape(shape)


def require_gpu():
    if device_count() < 1:
        raise SsqHipError("no HIP device visible: ssqueeze_rs_amd computes on an MI355X only "
                          "(there is no CPU fallback)")
