"""`ssqueeze_rs_amd._rs` -- drop-in for the reference's PyO3 extension module `ssqueeze._rs`
(registered at rust/src/lib.rs:22-35): same six callables, same keyword names, defaults,
return arity, dtypes, layouts and error behaviour, backed by libssq_hip.so (hand-written
HIP kernels for MI355X) through ctypes.  There is no CPU fallback.

Extensions over the reference (which takes 1-D float64 only):
  * float32 input -> complex64 output (fp32 compute, the MI355X fast path);
  * 2-D `[batch, N]` input -> outputs gain a leading batch axis.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from ._lib import PAD, SQUEEZE, SSQ_F32, SSQ_F64, WAVELET


class PanicException(BaseException):
    """Mirror of `pyo3_runtime.PanicException` (derives from BaseException): raised where the
    reference's Rust code panics (slice/plan length mismatch, index out of bounds, /0)."""


def hello_from_bin() -> str:
    """lib.rs:16-19."""
    return _lib.load().ssq_hello_from_bin().decode()


# --------------------------------------------------------------------------- helpers
def _as_signal(x, name="x"):
    """PyReadonlyArray1<f64> extraction (ssq_stft.rs:76): ndarray, float64, 1-D -- anything
    else is a TypeError.  Extension: float32 and a leading batch axis."""
    if not isinstance(x, np.ndarray):
        raise TypeError(f"argument '{name}': '{type(x).__name__}' object cannot be converted to 'PyArray<T, D>'")
    if x.dtype not in (np.float64, np.float32):
        raise TypeError(f"argument '{name}': type mismatch: from=float64/float32 expected, got {x.dtype}")
    if x.ndim not in (1, 2):
        raise TypeError(f"argument '{name}': dimensionality mismatch: expected 1 (or 2 = [batch, N]), got {x.ndim}")
    batched = x.ndim == 2
    xa = np.ascontiguousarray(x if batched else x[None, :])   # the reference copies too (to_owned, :87)
    return xa, batched, (SSQ_F32 if x.dtype == np.float32 else SSQ_F64)


def _as_f64_vector(a, name):
    if not isinstance(a, np.ndarray):
        raise TypeError(f"argument '{name}': '{type(a).__name__}' object cannot be converted to 'PyArray<T, D>'")
    if a.dtype not in (np.float64, np.float32) or a.ndim != 1:
        raise TypeError(f"argument '{name}': expected a 1-D float64 array, got {a.dtype} ndim={a.ndim}")
    return np.ascontiguousarray(a, dtype=np.float64)


def _usize(v, name):
    if isinstance(v, (bool, np.bool_)) or not isinstance(v, (int, np.integer)):
        raise TypeError(f"argument '{name}': '{type(v).__name__}' object cannot be interpreted as an integer")
    if v < 0:
        raise OverflowError("can't convert negative int to unsigned")
    return int(v)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _cdtype(code):
    return np.complex64 if code == SSQ_F32 else np.complex128


def _call(rc):
    """Map a failing libssq_hip status to the exception the reference would raise."""
    if rc == 0:
        return
    msg = _lib.load().ssq_last_error().decode("utf-8", "replace")
    low = msg.lower()
    if any(s in low for s in ("index out of bounds", "divide by zero", "underflow", "empty input", "panic")):
        raise PanicException(msg)
    if low.startswith("hip") or "hip" in low.split(":")[0]:
        raise _lib.SsqHipError(msg)
    raise ValueError(msg)


# --------------------------------------------------------------------------- stft
def stft(x, n_fft, hop_length, window, padtype, _upstream=False):
    """rust/src/spectral/stft.rs:12-95.  Returns (Sx complex [n_freqs, n_frames], freqs float64
    [n_freqs] in cycles/sample).  All five arguments are required, as in the reference.
    `_upstream=True` (not in the reference): the same call with the numerics of the vendored upstream ssqueezepy
    (modulated frames, its pad split; `ssqueeze_rs_amd.upstream` has upstream's own signatures and the inverses)."""
    if _upstream:
        from . import upstream as _up
        Sx = _up.stft(x, window, n_fft=n_fft, hop_len=hop_length, padtype=padtype)
        nf = Sx.shape[-2]
        return Sx, np.arange(nf, dtype=np.float64) * (0.5 / (nf - 1) if nf > 1 else 0.0)
    lib = _lib.load()
    xa, batched, code = _as_signal(x)
    n_fft = _usize(n_fft, "n_fft")
    hop = _usize(hop_length, "hop_length")
    win = _as_f64_vector(window, "window")
    if not isinstance(padtype, str):
        raise TypeError("argument 'padtype': 'str' expected")
    batch, N = xa.shape
    if N == 0 or n_fft == 0:
        raise PanicException("attempt to subtract with overflow")           # stft.rs:33
    if hop == 0:
        raise PanicException("attempt to divide by zero")                   # stft.rs:33
    if win.shape[0] != n_fft:
        # apply_window truncates (stft_utils.rs:8); rustfft then panics on the length (stft.rs:67)
        raise PanicException(
            f"Provided FFT buffer was too small. Expected len = {n_fft}, got len = {min(n_fft, win.shape[0])}")
    _lib.require_gpu()
    n_freqs = n_fft // 2 + 1
    n_frames = (N - 1) // hop + 1
    Sx = _lib.pinned_empty((batch, n_freqs, n_frames), _cdtype(code))     # results arrive by DMA (no staging copy)
    freqs = np.empty(n_freqs, dtype=np.float64)
    _call(lib.ssq_stft_host(code, _ptr(xa), batch, N, _ptr(win), n_fft, hop, PAD.get(padtype, 0),
                            _ptr(Sx), _ptr(freqs)))
    return (Sx if batched else Sx[0]), freqs


# --------------------------------------------------------------------------- ssq_stft
def ssq_stft(x, window, n_fft=None, win_len=None, hop_len=1, fs=1.0, padtype="reflect",
             squeezing="sum", gamma=None, _debug=False, _upstream=False):
    """rust/src/spectral/ssq_stft.rs:72-313.  Returns (Tx complex [n_freqs, n_frames],
    ssq_freqs float64 [n_freqs]).  `_debug=True` (not in the reference) additionally returns a
    dict with the kernel's Sx, dSx, w and k.  `_upstream=True`: upstream ssqueezepy's numerics (see `stft`)."""
    if _upstream:
        from . import upstream as _up
        Tx, _, f, _ = _up.ssq_stft(x, window, n_fft=n_fft, win_len=win_len, hop_len=hop_len, fs=fs, padtype=padtype,
                                   squeezing=squeezing, gamma=gamma)
        return Tx, f.astype(np.float64)
    lib = _lib.load()
    xa, batched, code = _as_signal(x)
    win = _as_f64_vector(window, "window")
    batch, N = xa.shape
    n_fft = min(N, 512) if n_fft is None else _usize(n_fft, "n_fft")        # :92
    win_len = win.shape[0] if win_len is None else _usize(win_len, "win_len")   # :93
    hop = _usize(hop_len, "hop_len")
    fs = float(fs)
    if win_len > n_fft:                                                      # :96-101
        raise ValueError(f"Window length {win_len} cannot be greater than n_fft {n_fft}")
    if N == 0 or n_fft == 0:
        raise PanicException("attempt to subtract with overflow")           # :183
    if hop == 0:
        raise PanicException("attempt to divide by zero")                   # :183
    n_freqs = n_fft // 2 + 1
    if n_freqs < 2:
        raise PanicException("index out of bounds: the len is 1 but the index is 1")   # :273
    sized = np.empty(n_fft, dtype=np.float64)                                # :104-119
    _call(lib.ssq_size_window(_ptr(win), win.shape[0], n_fft, _ptr(sized)))
    _lib.require_gpu()
    n_frames = (N - 1) // hop + 1
    cd = _cdtype(code)
    Tx = _lib.pinned_empty((batch, n_freqs, n_frames), cd)
    ssq_freqs = np.empty(n_freqs, dtype=np.float64)
    dbg = [_lib.pinned_empty(Tx.shape, cd) for _ in range(3)] if _debug else [None, None, None]
    g = -1.0 if gamma is None else float(gamma)
    _call(lib.ssq_ssq_stft_host(code, _ptr(xa), batch, N, _ptr(sized), n_fft, hop, fs,
                                PAD.get(padtype, 0), SQUEEZE.get(squeezing, 0), g,
                                _ptr(Tx), _ptr(ssq_freqs), _ptr(dbg[0]), _ptr(dbg[1]), _ptr(dbg[2])))
    out = (Tx if batched else Tx[0]), ssq_freqs
    if _debug:
        sel = (lambda a: a) if batched else (lambda a: a[0])
        wk = sel(dbg[2])
        return out + (dict(Sx=sel(dbg[0]), dSx=sel(dbg[1]), w=wk.real.copy(),
                           k=np.rint(wk.imag).astype(np.int64)),)
    return out


# --------------------------------------------------------------------------- cwt family
def _dt_from(fs, t):
    """cwt.rs:66-76 / ssq_cwt.rs:283-293."""
    if t is not None:
        tv = _as_f64_vector(t, "t")
        if tv.shape[0] < 2:
            raise ValueError("Time vector must have at least 2 elements")
        return float(tv[1] - tv[0])
    if fs is not None:
        return 1.0 / float(fs)
    return 1.0


def _scales_or_default(scales, N, nv, simd_variant):
    lib = _lib.load()
    if scales is not None:
        return _as_f64_vector(scales, "scales").copy()
    na = C.c_int64(0)
    _call(lib.ssq_log_scales(N, _usize(nv, "nv"), int(simd_variant), C.byref(na), None))
    s = np.empty(na.value, dtype=np.float64)
    if na.value:
        _call(lib.ssq_log_scales(N, nv, int(simd_variant), C.byref(na), _ptr(s)))
    return s


def _cwt_impl(x, wavelet, scales, fs, t, nv, l1_norm, derivative, padtype, rpadded, simd_variant):
    lib = _lib.load()
    xa, batched, code = _as_signal(x)
    batch, N = xa.shape
    dt = _dt_from(fs, t)
    sc = _scales_or_default(scales, N, nv, simd_variant)
    na = sc.shape[0]
    if N == 0:
        raise PanicException("empty input")
    cd = _cdtype(code)
    P, n1 = C.c_int64(0), C.c_int64(0)
    _call(lib.ssq_cwt_pad_len(N, C.byref(P), C.byref(n1)))
    cols = P.value if rpadded else N
    Wx = _lib.pinned_empty((batch, na, cols), cd)
    dWx = _lib.pinned_empty((batch, na, cols), cd) if derivative else None
    if na > 0:
        _lib.require_gpu()
        _call(lib.ssq_cwt_host(code, _ptr(xa), batch, N, WAVELET.get(wavelet, 0), _ptr(sc), na, dt,
                               int(bool(l1_norm)), PAD.get(padtype, 0), int(bool(rpadded)),
                               _ptr(Wx), _ptr(dWx)))
    if not batched:
        Wx = Wx[0]
        dWx = None if dWx is None else dWx[0]
    return Wx, sc, dWx                                                        # always a 3-tuple (cwt.rs:143)


def cwt(x, wavelet="gmw", scales=None, fs=None, t=None, nv=32, l1_norm=True, derivative=False,
        padtype="reflect", rpadded=False, vectorized=True, patience=0, _upstream=False):
    """rust/src/spectral/cwt.rs:46-144.  Returns (Wx, scales, dWx or None).  `vectorized` selects
    between two code paths with identical numbers in the reference; `patience` is ignored there.
    `_upstream=True`: upstream ssqueezepy's numerics (p2up padding, normalised wavelets) on the same scales."""
    if _upstream:
        from . import upstream as _up
        sc = _scales_or_default(scales, np.asarray(x).shape[-1], nv, False)
        out = _up.cwt(x, wavelet, scales=sc, fs=fs, t=t, l1_norm=l1_norm, derivative=derivative, padtype=padtype,
                      rpadded=rpadded)
        return (out[0], sc, out[2] if derivative else None)
    return _cwt_impl(x, wavelet, scales, fs, t, nv, l1_norm, derivative, padtype, rpadded, False)


def cwt_simd(x, wavelet="gmw", scales=None, fs=None, t=None, nv=32, l1_norm=True, derivative=False,
             padtype="reflect", rpadded=False, vectorized=True, patience=0):
    """rust/src/spectral/cwt_simd.rs:52-150: `cwt` with exp(p*ln2) automatic scales (:474-545)."""
    return _cwt_impl(x, wavelet, scales, fs, t, nv, l1_norm, derivative, padtype, rpadded, True)


def ssq_cwt(x, wavelet="gmw", scales=None, fs=None, t=None, ssq_freqs=None, nv=32,
            padtype="reflect", squeezing="sum", maprange="peak", difftype="trig", gamma=None,
            vectorized=True, flipud=True, _debug=False, _upstream=False):
    """rust/src/spectral/ssq_cwt.rs:244-493.  Returns (Tx complex [n_scales, N], ssq_freqs
    float64 [n_scales]).  `ssq_freqs` is a string ("log"/"linear") as in the reference (:268);
    `difftype` and `vectorized` are accepted and unused there (:296-297).
    `_upstream=True`: upstream ssqueezepy's numerics (clamped bins, ln2/nv, centre-frequency ssq_freqs)."""
    if _upstream:
        from . import upstream as _up
        sc = _scales_or_default(scales, np.asarray(x).shape[-1], nv, False)
        Tx, _, f, _ = _up.ssq_cwt(x, wavelet, scales=sc, fs=fs, t=t, ssq_freqs=ssq_freqs, padtype=padtype,
                                  squeezing=squeezing, maprange=maprange, gamma=gamma, flipud=flipud)
        return Tx, f.astype(np.float64)
    lib = _lib.load()
    xa, batched, code = _as_signal(x)
    batch, N = xa.shape
    dt = _dt_from(fs, t)
    if ssq_freqs is not None and not isinstance(ssq_freqs, str):
        raise TypeError("argument 'ssq_freqs': 'str' expected")
    sc = _scales_or_default(scales, N, nv, False)
    na = sc.shape[0]
    if N == 0:
        raise PanicException("empty input")
    if na == 0:
        raise PanicException("index out of bounds: the len is 0 but the index is 18446744073709551615")  # :459
    _lib.require_gpu()
    cd = _cdtype(code)
    Tx = _lib.pinned_empty((batch, na, N), cd)
    freqs = np.empty(na, dtype=np.float64)
    dbg = [_lib.pinned_empty(Tx.shape, cd) for _ in range(3)] if _debug else [None, None, None]
    g = -1.0 if gamma is None else float(gamma)
    _call(lib.ssq_ssq_cwt_host(code, _ptr(xa), batch, N, WAVELET.get(wavelet, 0), _ptr(sc), na, dt,
                               1 if ssq_freqs == "linear" else 0, 1 if maprange == "maximal" else 0,
                               PAD.get(padtype, 0), SQUEEZE.get(squeezing, 0), int(bool(flipud)), g,
                               _ptr(Tx), _ptr(freqs), _ptr(dbg[0]), _ptr(dbg[1]), _ptr(dbg[2])))
    out = (Tx if batched else Tx[0]), freqs
    if _debug:
        sel = (lambda a: a) if batched else (lambda a: a[0])
        wk = sel(dbg[2])
        return out + (dict(Wx=sel(dbg[0]), dWx=sel(dbg[1]), w=wk.real.copy(),
                           k=np.rint(wk.imag).astype(np.int64), scales=sc),)
    return out


# --------------------------------------------------------------------------- icwt + wavelet helpers
def icwt(Wx, wavelet="gmw", scales=None, nv=None, one_int=True, x_len=None, x_mean=0.0, padtype="reflect",
         rpadded=False, l1_norm=True):
    """rust/src/spectral/cwt.rs:550-718 (advertised by _rs.pyi:61-73).  `Wx` complex128 [n_scales, n_times]
    (extension: complex64); returns float64 [x_len or n_times].  `nv`, `padtype`, `rpadded` are unused there too."""
    lib = _lib.load()
    if not isinstance(Wx, np.ndarray) or Wx.ndim != 2 or Wx.dtype not in (np.complex128, np.complex64):
        raise TypeError("argument 'Wx': expected a 2-D complex128 array")
    if scales is None:
        raise ValueError("Scales must be provided")                              # :571-575
    sc = _as_f64_vector(scales, "scales")
    Wc = np.ascontiguousarray(Wx)
    na, n_times = Wc.shape
    xl = n_times if x_len is None else _usize(x_len, "x_len")
    if xl > n_times or sc.shape[0] < na:
        raise PanicException("index out of bounds")                              # Wx_array[[i, j]] / scales_array[i]
    out = np.empty(xl, dtype=np.float64)
    if xl == 0:
        return out
    _lib.require_gpu()
    _call(lib.ssq_icwt_host(SSQ_F32 if Wc.dtype == np.complex64 else SSQ_F64, _ptr(Wc), na, n_times,
                            WAVELET.get(wavelet, 0) if wavelet in WAVELET else 0, _ptr(sc), sc.shape[0],
                            int(bool(one_int)), xl, float(x_mean), int(bool(l1_norm)), _ptr(out)))
    return out


def _cplx_out(n):
    return np.empty(int(n), dtype=np.complex128)


def morlet(w, mu=6.0, dtype="float64"):
    """rust/src/wavelets/morlet.rs:59-77: Morlet in the frequency domain at the given `w` (complex128, imag = 0)."""
    wv = _as_f64_vector(w, "w")
    out = _cplx_out(wv.shape[0])
    _call(_lib.load().ssq_morlet(_ptr(wv), wv.shape[0], float(mu), _ptr(out)))
    return out


def morlet_freq(n=1024, scale=1.0, mu=6.0, dtype="float64"):
    """rust/src/wavelets/morlet.rs:80-100."""
    out = _cplx_out(_usize(n, "n"))
    _call(_lib.load().ssq_morlet_freq(n, float(scale), float(mu), _ptr(out)))
    return out


def morlet_time(n=1024, scale=1.0, mu=6.0, dtype="float64"):
    """rust/src/wavelets/morlet.rs:103-145."""
    out = _cplx_out(_usize(n, "n"))
    _call(_lib.load().ssq_morlet_time(n, float(scale), float(mu), _ptr(out)))
    return out


def gmw(w, gamma=3.0, beta=60.0, norm="bandpass", order=0, dtype="float64"):
    """rust/src/wavelets/gmw.rs:236-262 (ValueError for gamma <= 0, beta < 0, order < 0)."""
    wv = _as_f64_vector(w, "w")
    out = _cplx_out(wv.shape[0])
    _call(_lib.load().ssq_gmw(_ptr(wv), wv.shape[0], float(gamma), float(beta), str(norm).encode(), int(order), _ptr(out)))
    return out


def gmw_freq(n=1024, scale=1.0, gamma=3.0, beta=60.0, norm="bandpass", order=0, dtype="float64"):
    """rust/src/wavelets/gmw.rs:265-289."""
    out = _cplx_out(_usize(n, "n"))
    _call(_lib.load().ssq_gmw_freq(n, float(scale), float(gamma), float(beta), str(norm).encode(), int(order), _ptr(out)))
    return out


def gmw_time(n=1024, scale=1.0, gamma=3.0, beta=60.0, norm="bandpass", order=0, dtype="float64"):
    """rust/src/wavelets/gmw.rs:292-337."""
    out = _cplx_out(_usize(n, "n"))
    _call(_lib.load().ssq_gmw_time(n, float(scale), float(gamma), float(beta), str(norm).encode(), int(order), _ptr(out)))
    return out


def gmw_center_frequency(gamma=3.0, beta=60.0, kind="peak"):
    """rust/src/wavelets/gmw.rs:340-357 ("peak" | "energy"; anything else -> ValueError)."""
    v = C.c_double(0.0)
    _call(_lib.load().ssq_gmw_center_frequency(float(gamma), float(beta), str(kind).encode(), C.byref(v)))
    return v.value


__all__ = ["hello_from_bin", "stft", "ssq_stft", "cwt", "cwt_simd", "ssq_cwt", "icwt", "morlet", "morlet_freq",
           "morlet_time", "gmw", "gmw_freq", "gmw_time", "gmw_center_frequency", "PanicException"]
