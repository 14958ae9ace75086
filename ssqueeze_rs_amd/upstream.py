"""`ssqueeze_rs_amd.upstream` -- the upstream-parity mode and the inverses on the MI355X engine.

The Rust crate this package replaces was derived from ssqueezepy, vendored at /root/reference/old/ssqueezepy; the two
are numerically DIFFERENT variants (pad split, modulation, diff-window, wavelet normalisation, bin rule, constants --
SURVEY 8(a) lists them).  This module mirrors upstream's callables -- same names, keyword names, defaults, return
arity -- for the supported subset, backed by the *_v / inverse entry points of libssq_hip.so (include/ssq_hip.h).
Host logic upstream does in Python (window sizing, scale-type inference, frequency vectors) is Python here too;
everything per sample runs in HIP kernels.  There is no CPU fallback.

Supported subset (anything else raises ValueError naming the option):
  * `window`: ndarray (upstream's default DPSS / named windows come from scipy.signal, which callers can pass in);
  * `scales`: explicit exponentially spaced ndarray ('log' scaletype; the automatic 'log-piecewise' bounds are not built);
  * wavelets 'gmw' (gamma, beta; bandpass norm, order 0) and 'morlet' (mu) -- names, or (name, {params});
  * difftype 'trig'; squeezing 'sum' / 'lebesgue'; padtype 'reflect' / 'zero'; full inverses (no component curves).
dtype: float64 in -> complex128 (upstream's 'float64'); float32 in -> complex64 (upstream's default 'float32').

Parity: unpinned against upstream itself (it does not import here: numba missing); tests compare with the numba-free
restatement oracle/upstream_oracle.py and pin upstream's own reconstruction thresholds
(old/tests/reconstruction_test.py:111-123, :160-206).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import PAD, SQUEEZE, SSQ_F32, SSQ_F64, WAVELET
from ._rs import _call, _cdtype, _ptr

VARIANT_UPSTREAM, VARIANT_MODULATED, VARIANT_FLIPUD = 1, 2, 4
EPS32, EPS64 = float(np.finfo(np.float32).eps), float(np.finfo(np.float64).eps)


# ------------------------------------------------------------------------------------------------------- helpers ----
def _signal(x):
    if not isinstance(x, np.ndarray) or x.ndim not in (1, 2):
        raise TypeError("`x` must be a 1D or 2D numpy array")
    if x.dtype not in (np.float32, np.float64):
        x = x.astype(np.float64)
    batched = x.ndim == 2
    xa = np.ascontiguousarray(x if batched else x[None, :])
    return xa, batched, (SSQ_F32 if xa.dtype == np.float32 else SSQ_F64)


def _pad_code(padtype):
    if padtype not in PAD:
        raise ValueError(f"padtype {padtype!r}: the MI355X engine builds 'reflect' and 'zero'")
    return PAD[padtype]


def get_window(window, win_len, n_fft=None):
    """old/ssqueezepy/_stft.py:257-309 for an ndarray window: centre zero-pad to n_fft."""
    if not isinstance(window, np.ndarray):
        raise ValueError("`window` must be an ndarray here (named / default DPSS windows: scipy.signal.get_window, "
                         "scipy.signal.windows.dpss(win_len, max(4, win_len//8), sym=False))")
    window = np.asarray(window, dtype=np.float64)
    if n_fft is None:
        return window
    if win_len > n_fft:
        raise ValueError("Can't have `win_len > n_fft` ({} > {})".format(win_len, n_fft))
    pl = (n_fft - win_len) // 2
    pr = n_fft - win_len - pl
    if len(window) < (win_len + pl + pr):
        window = np.pad(window, [pl, pr])
    return np.ascontiguousarray(window)


def _wavelet(wavelet):
    name, kw = (wavelet, {}) if isinstance(wavelet, str) else wavelet
    if name == "gmw":
        if kw.get("norm", "bandpass") != "bandpass" or kw.get("order", 0) != 0 or kw.get("centered_scale", False):
            raise ValueError("gmw: only norm='bandpass', order=0, centered_scale=False are built")
        return WAVELET["gmw"], float(kw.get("gamma", 3.0)), float(kw.get("beta", 60.0))
    if name == "morlet":
        return WAVELET["morlet"], float(kw.get("mu", 13.4)), 0.0
    raise ValueError(f"wavelet {name!r}: the MI355X engine builds 'gmw' and 'morlet'")


def _scales(scales):
    if not isinstance(scales, np.ndarray):
        raise ValueError("`scales` must be an explicit ndarray (the automatic scale bounds of 'log-piecewise' / 'log' are "
                         "not built)")
    s = np.ascontiguousarray(scales, dtype=np.float64).reshape(-1)
    if len(s) < 2 or np.mean(np.abs(np.diff(np.log(s), 2))) >= 4e-15 * 1e3:     # utils/cwt_utils.py:264-298
        raise ValueError("`scales` must be exponentially spaced (scaletype 'log')")
    return s, int(np.round(1 / np.diff(np.log2(s))[0]))


def _dt(fs, t, N):
    """utils/cwt_utils.py:698-720 (_process_fs_and_t)."""
    if t is not None:
        if len(t) != N:
            raise ValueError("`t` must be of same length as `x`")
        return float((t[-1] - t[0]) / (N - 1))
    return 1.0 / float(fs) if fs is not None else 1.0


def adm_ssq(wavelet) -> float:
    """utils/cwt_utils.py:28-47."""
    code, p0, p1 = _wavelet(wavelet)
    out = C.c_double(0)
    _call(_lib.load().ssq_upstream_adm(code, p0, p1, 0, C.byref(out)))
    return out.value


def adm_cwt(wavelet) -> float:
    """utils/cwt_utils.py:50-63."""
    code, p0, p1 = _wavelet(wavelet)
    out = C.c_double(0)
    _call(_lib.load().ssq_upstream_adm(code, p0, p1, 1, C.byref(out)))
    return out.value


def p2up(n):
    """utils/common.py:32-51."""
    up, n1, n2 = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    _call(_lib.load().ssq_upstream_p2up(int(n), C.byref(up), C.byref(n1), C.byref(n2)))
    return up.value, n1.value, n2.value


# --------------------------------------------------------------------------------------------------- STFT family ----
def stft(x, window=None, n_fft=None, win_len=None, hop_len=1, fs=None, t=None, padtype="reflect", modulated=True,
         derivative=False, dtype=None):
    """ssqueezepy.stft (old/ssqueezepy/_stft.py:13-193) -> Sx, or (Sx, dSx) with derivative=True."""
    lib = _lib.load()
    xa, batched, code = _signal(x)
    batch, N = xa.shape
    fs = 1.0 / _dt(fs, t, N)
    n_fft = n_fft or min(N // hop_len, 512)
    if win_len is None:
        win_len = len(window) if isinstance(window, np.ndarray) else n_fft
    win = get_window(window, win_len, n_fft)
    _lib.require_gpu()
    variant = VARIANT_UPSTREAM | (VARIANT_MODULATED if modulated else 0)
    shape = (batch, n_fft // 2 + 1, (N - 1) // hop_len + 1)
    Sx = _lib.pinned_empty(shape, _cdtype(code))
    dSx = _lib.pinned_empty(shape, _cdtype(code)) if derivative else None
    _call(lib.ssq_stft_host_v(code, _ptr(xa), batch, N, _ptr(win), n_fft, hop_len, fs, _pad_code(padtype), variant,
                              _ptr(Sx), _ptr(dSx)))
    if not batched:
        Sx, dSx = Sx[0], (dSx[0] if derivative else None)
    return (Sx, dSx) if derivative else Sx


def istft(Sx, window=None, n_fft=None, win_len=None, hop_len=1, N=None, modulated=True, win_exp=1):
    """ssqueezepy.istft (old/ssqueezepy/_stft.py:196-254)."""
    lib = _lib.load()
    if not isinstance(Sx, np.ndarray) or Sx.ndim != 2 or Sx.dtype not in (np.complex64, np.complex128):
        raise TypeError("`Sx` must be a 2D complex64 / complex128 array")
    n_fft = n_fft or (Sx.shape[0] - 1) * 2
    win_len = win_len or n_fft
    N = N or hop_len * Sx.shape[1]
    if Sx.shape[0] != n_fft // 2 + 1:
        raise ValueError("`Sx` has %d rows, n_fft=%d needs %d" % (Sx.shape[0], n_fft, n_fft // 2 + 1))
    win = get_window(window, win_len, n_fft)
    code = SSQ_F32 if Sx.dtype == np.complex64 else SSQ_F64
    _lib.require_gpu()
    Sc = np.ascontiguousarray(Sx)
    x = np.empty(N, dtype=np.float32 if code == SSQ_F32 else np.float64)
    _call(lib.ssq_istft_host(code, _ptr(Sc), Sc.shape[1], _ptr(win), n_fft, hop_len, N, int(bool(modulated)),
                             int(win_exp), _ptr(x)))
    return x


def ssq_stft(x, window=None, n_fft=None, win_len=None, hop_len=1, fs=None, t=None, modulated=True, ssq_freqs=None,
             padtype="reflect", squeezing="sum", gamma=None, preserve_transform=None, dtype=None, astensor=True,
             flipud=False, get_w=False, get_dWx=False):
    """ssqueezepy.ssq_stft (old/ssqueezepy/_ssq_stft.py:12-137) -> (Tx, Sx, ssq_freqs, Sfs[, w][, dSx])."""
    lib = _lib.load()
    if ssq_freqs is not None:
        raise ValueError("`ssq_freqs` other than None (= Sfs, linear) is not built")
    if squeezing not in SQUEEZE:
        raise ValueError(f"squeezing {squeezing!r}: 'sum' and 'lebesgue' are built")
    xa, batched, code = _signal(x)
    batch, N = xa.shape
    fs = 1.0 / _dt(fs, t, N)
    n_fft = n_fft or min(N // hop_len, 512)
    if win_len is None:
        win_len = len(window) if isinstance(window, np.ndarray) else n_fft
    win = get_window(window, win_len, n_fft)
    _lib.require_gpu()
    variant = VARIANT_UPSTREAM | (VARIANT_MODULATED if modulated else 0) | (VARIANT_FLIPUD if flipud else 0)
    n_freqs, n_frames = n_fft // 2 + 1, (N - 1) // hop_len + 1
    shape = (batch, n_freqs, n_frames)
    cdt = _cdtype(code)
    Tx, Sx = _lib.pinned_empty(shape, cdt), _lib.pinned_empty(shape, cdt)
    dSx = _lib.pinned_empty(shape, cdt) if get_dWx else None
    wk = _lib.pinned_empty(shape, cdt) if get_w else None
    f = np.empty(n_freqs, dtype=np.float64)
    _call(lib.ssq_ssq_stft_host_v(code, _ptr(xa), batch, N, _ptr(win), n_fft, hop_len, fs, _pad_code(padtype),
                                  SQUEEZE[squeezing], -1.0 if gamma is None else float(gamma), variant, _ptr(Tx),
                                  _ptr(f), _ptr(Sx), _ptr(dSx), _ptr(wk)))
    rdt = np.float32 if code == SSQ_F32 else np.float64
    Sfs = np.linspace(0, .5 * fs, n_freqs, dtype=rdt)                      # _ssq_stft.py:248-257
    out = [Tx if batched else Tx[0], Sx if batched else Sx[0], f.astype(rdt), Sfs]
    if get_w:
        w = wk.real.copy()
        out.append(w if batched else w[0])
    if get_dWx:
        out.append(dSx if batched else dSx[0])
    return tuple(out)


def issq_stft(Tx, window=None, cc=None, cw=None, n_fft=None, win_len=None, hop_len=1, modulated=True):
    """ssqueezepy.issq_stft (old/ssqueezepy/_ssq_stft.py:139-198), full inverse."""
    if not modulated:
        raise ValueError("inversion with `modulated == False` is unsupported.")
    if hop_len != 1:
        raise ValueError("inversion with `hop_len != 1` is unsupported.")
    if cc is not None or cw is not None:
        raise ValueError("component inversion (cc, cw) is not built: full inverse only")
    n_fft = n_fft or (Tx.shape[0] - 1) * 2
    win_len = win_len or n_fft
    win = get_window(window, win_len, n_fft)
    return _issq(Tx, 2.0 / float(win[len(win) // 2]))


def _issq(Tx, scale, row_scale=None):
    lib = _lib.load()
    if not isinstance(Tx, np.ndarray) or Tx.ndim != 2 or Tx.dtype not in (np.complex64, np.complex128):
        raise TypeError("`Tx` must be a 2D complex64 / complex128 array")
    code = SSQ_F32 if Tx.dtype == np.complex64 else SSQ_F64
    _lib.require_gpu()
    Tc = np.ascontiguousarray(Tx)
    x = np.empty(Tc.shape[1], dtype=np.float32 if code == SSQ_F32 else np.float64)
    rs = None if row_scale is None else np.ascontiguousarray(row_scale, dtype=np.float64)
    _call(lib.ssq_issq_host(code, _ptr(Tc), Tc.shape[0], Tc.shape[1], float(scale), _ptr(rs), _ptr(x)))
    return x


# ---------------------------------------------------------------------------------------------------- CWT family ----
def cwt(x, wavelet="gmw", scales="log-piecewise", fs=None, t=None, nv=32, l1_norm=True, derivative=False,
        padtype="reflect", rpadded=False, vectorized=True, astensor=True, cache_wavelet=None, order=0, average=None,
        nan_checks=None, patience=0):
    """ssqueezepy.cwt (old/ssqueezepy/_cwt.py:12-318) -> (Wx, scales[, dWx]).  `scales` must be an explicit array (the
    default string asks for upstream's automatic bounds, which are not built); `vectorized`, `astensor`, `cache_wavelet`,
    `nan_checks`, `patience` select code paths with identical numbers upstream and are accepted and unused."""
    if order != 0 or average is not None:
        raise ValueError("higher-order GMWs (`order`, `average`) are not built")
    if isinstance(scales, np.ndarray):
        nv = None                                            # _cwt.py:226-227
    lib = _lib.load()
    xa, batched, code = _signal(x)
    batch, N = xa.shape
    dt = _dt(fs, t, N)
    wcode, p0, p1 = _wavelet(wavelet)
    s, _nv = _scales(scales)
    if nv is not None and nv != _nv:
        raise Exception("`nv` used in `scales` differs from `nv` passed (%s != %s)" % (_nv, nv))   # cwt_utils.py:229-231
    _lib.require_gpu()
    cols = p2up(N)[0] if rpadded else N
    shape = (batch, len(s), cols)
    Wx = _lib.pinned_empty(shape, _cdtype(code))
    dWx = _lib.pinned_empty(shape, _cdtype(code)) if derivative else None
    _call(lib.ssq_cwt_host_v(code, _ptr(xa), batch, N, wcode, p0, p1, _ptr(s), len(s), dt, int(bool(l1_norm)),
                             _pad_code(padtype), int(bool(rpadded)), VARIANT_UPSTREAM, _ptr(Wx), _ptr(dWx)))
    sc = s.astype(np.float32 if code == SSQ_F32 else np.float64)
    if not batched:
        Wx, dWx = Wx[0], (dWx[0] if derivative else None)
    return (Wx, sc, dWx) if derivative else (Wx, sc)


def _ssq_freqs(s, N, wcode, p0, p1, dt, maprange, scaletype):
    """ssqueezing.py:218-290 (ascending)."""
    na = len(s)
    if maprange == "maximal":
        fm, fM = 1 / (dt * N), 1 / (2 * dt)
    elif maprange == "peak":
        lib = _lib.load()
        Np = p2up(N)[0]
        wc = C.c_double(0)
        _call(lib.ssq_upstream_center_frequency(wcode, p0, p1, float(s[-1]), Np, C.byref(wc)))
        fm = wc.value / (2 * np.pi) / dt
        _call(lib.ssq_upstream_center_frequency(wcode, p0, p1, float(s[0]), Np, C.byref(wc)))
        fM = wc.value / (2 * np.pi) / dt
    else:
        raise ValueError(f"maprange {maprange!r}: 'peak' and 'maximal' are built")
    if scaletype == "log":
        return fm * np.power(fM / fm, np.arange(na) / (na - 1))
    if scaletype == "linear":
        return np.linspace(fm, fM, na)
    raise ValueError(f"ssq_freqs {scaletype!r}: 'log' and 'linear' are built")


def ssq_cwt(x, wavelet="gmw", scales="log-piecewise", nv=None, fs=None, t=None, ssq_freqs=None, padtype="reflect",
            squeezing="sum", maprange="peak", difftype="trig", difforder=None, gamma=None, vectorized=True,
            preserve_transform=None, astensor=True, order=0, nan_checks=None, patience=0, flipud=True,
            cache_wavelet=None, get_w=False, get_dWx=False):
    """ssqueezepy.ssq_cwt (old/ssqueezepy/_ssq_cwt.py:12-311) -> (Tx, Wx, ssq_freqs, scales[, w][, dWx])."""
    lib = _lib.load()
    if difftype != "trig" or order != 0:
        raise ValueError("only difftype='trig', order=0 are built")
    if squeezing not in SQUEEZE:
        raise ValueError(f"squeezing {squeezing!r}: 'sum' and 'lebesgue' are built")
    xa, batched, code = _signal(x)
    batch, N = xa.shape
    dt = _dt(fs, t, N)
    wcode, p0, p1 = _wavelet(wavelet)
    s, _nv = _scales(scales)
    if nv is not None and nv != _nv:
        raise Exception("`nv` used in `scales` differs from `nv` passed (%s != %s)" % (_nv, nv))
    scaletype = ssq_freqs if isinstance(ssq_freqs, str) else "log"
    if ssq_freqs is not None and not isinstance(ssq_freqs, str):
        raise ValueError("`ssq_freqs` arrays are not built: None, 'log' or 'linear'")
    f_asc = np.ascontiguousarray(_ssq_freqs(s, N, wcode, p0, p1, dt, maprange, scaletype), dtype=np.float64)
    _lib.require_gpu()
    shape = (batch, len(s), N)
    cdt = _cdtype(code)
    Tx, Wx = _lib.pinned_empty(shape, cdt), _lib.pinned_empty(shape, cdt)
    dWx = _lib.pinned_empty(shape, cdt) if get_dWx else None
    wk = _lib.pinned_empty(shape, cdt) if get_w else None
    variant = VARIANT_UPSTREAM | (VARIANT_FLIPUD if flipud else 0)
    _call(lib.ssq_ssq_cwt_host_v(code, _ptr(xa), batch, N, wcode, p0, p1, _ptr(s), len(s), dt, _nv, _ptr(f_asc),
                                 0 if scaletype == "log" else 1, _pad_code(padtype), SQUEEZE[squeezing],
                                 -1.0 if gamma is None else float(gamma), variant, _ptr(Tx), _ptr(Wx), _ptr(dWx),
                                 _ptr(wk)))
    rdt = np.float32 if code == SSQ_F32 else np.float64
    out = [Tx if batched else Tx[0], Wx if batched else Wx[0], f_asc[::-1].astype(rdt), s.astype(rdt)]   # ssqueezing.py:199-205
    if get_w:
        w = wk.real.copy()
        out.append(w if batched else w[0])
    if get_dWx:
        out.append(dWx if batched else dWx[0])
    return tuple(out)


def issq_cwt(Tx, wavelet="gmw", cc=None, cw=None):
    """ssqueezepy.issq_cwt (old/ssqueezepy/_ssq_cwt.py:313-378), full inverse: (2 / Css) sum_rows Re Tx."""
    if cc is not None or cw is not None:
        raise ValueError("component inversion (cc, cw) is not built: full inverse only")
    return _issq(Tx, 2.0 / adm_ssq(wavelet))


def icwt(Wx, wavelet="gmw", scales="log-piecewise", nv=None, one_int=True, x_len=None, x_mean=0, padtype="reflect",
         rpadded=False, l1_norm=True):
    """ssqueezepy.icwt (old/ssqueezepy/_cwt.py:321-452), one-integral form on exponential scales:
    (2 / Cpsi) ln(2^(1/nv)) sum_a Re Wx[a] / (1 or sqrt(a))  + x_mean."""
    if not one_int:
        raise ValueError("only the one-integral inverse (one_int=True) is built")
    s, _nv = _scales(scales)
    if Wx.shape[0] != len(s):
        raise AssertionError("%s != %s" % (len(s), Wx.shape[0]))
    x = _issq(Wx, (2.0 / adm_ssq(wavelet)) * np.log(2 ** (1 / _nv)), None if l1_norm else 1.0 / np.sqrt(s))
    return x + x_mean
