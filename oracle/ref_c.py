"""ctypes wrapper of the oracle's C restatement (oracle/ssq_ref.c).  Test infrastructure only:
tests/ check it against the NumPy oracle; bench.py times it as `cpu_baseline` (kind "port")."""
import ctypes as C
import os

import numpy as np

from .build_ref import LIB, build_ref

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build_ref()
        lib = C.CDLL(LIB)
        lib.ssq_ref_ssq_stft.restype = C.c_int
        lib.ssq_ref_ssq_stft.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_long, C.c_double,
                                         C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ssq_ref_ssq_cwt.restype = C.c_int
        lib.ssq_ref_ssq_cwt.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_double, C.c_int,
                                        C.c_double, C.c_int, C.c_void_p, C.c_void_p]
        lib.ssq_ref_num_threads.restype = C.c_int
        _lib = lib
    return _lib


def num_threads() -> int:
    return load().ssq_ref_num_threads()


def ssq_stft(x, window_sized, n_fft, hop, fs=1.0, padtype="reflect", squeezing="sum", gamma=None,
             mode=0, nthreads=0, want_k=False):
    """mode 0: reference-faithful (linear-scan reassignment); mode 1: optimised CPU."""
    lib = load()
    x = np.ascontiguousarray(x, dtype=np.float64)
    w = np.ascontiguousarray(window_sized, dtype=np.float64)
    assert w.shape[0] == n_fft
    n = x.shape[0]
    n_freqs = n_fft // 2 + 1
    n_frames = (n - 1) // hop + 1
    Tx = np.empty((n_freqs, n_frames), dtype=np.complex128)
    f = np.empty(n_freqs, dtype=np.float64)
    k = np.empty((n_freqs, n_frames), dtype=np.int32) if want_k else None
    rc = lib.ssq_ref_ssq_stft(x.ctypes.data, n, w.ctypes.data, n_fft, hop, float(fs),
                              0 if padtype != "zero" else 1, 1 if squeezing == "lebesgue" else 0,
                              -1.0 if gamma is None else float(gamma), int(mode), int(nthreads),
                              Tx.ctypes.data, f.ctypes.data, None if k is None else k.ctypes.data)
    if rc != 0:
        raise RuntimeError("ssq_ref_ssq_stft failed")
    return (Tx, f, k) if want_k else (Tx, f)


def ssq_cwt(x, scales, wavelet="morlet", dt=1.0, flipud=True, gamma=None, nthreads=0):
    """oracle/ssq_ref.c::ssq_ref_ssq_cwt (ssq_cwt.rs structure: maprange "peak", log ssq_freqs, reflect, "sum")."""
    lib = load()
    x = np.ascontiguousarray(x, dtype=np.float64)
    s = np.ascontiguousarray(scales, dtype=np.float64)
    n, na = x.shape[0], s.shape[0]
    Tx = np.empty((na, n), dtype=np.complex128)
    f = np.empty(na, dtype=np.float64)
    rc = lib.ssq_ref_ssq_cwt(x.ctypes.data, n, s.ctypes.data, na, 1 if wavelet == "morlet" else 0, float(dt),
                             int(bool(flipud)), -1.0 if gamma is None else float(gamma), int(nthreads),
                             Tx.ctypes.data, f.ctypes.data)
    if rc != 0:
        raise RuntimeError("ssq_ref_ssq_cwt failed (%d)" % rc)
    return Tx, f
