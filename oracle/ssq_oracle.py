"""CPU oracle for the ssqueeze `_rs` hot path (stft / ssq_stft / cwt / cwt_simd / ssq_cwt) and the §8(f)
rows next to it (icwt, the wavelet helper functions).

TEST INFRASTRUCTURE ONLY.  Nothing under ``ssqueeze_rs_amd/`` may import this
module: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` use it, and only as the checker.

This is a NumPy (float64 / complex128) restatement of the algorithm of the
reference Rust crate, function by function, *including its quirks* (SURVEY.md
appendix).  Every function cites the reference ``file:line`` it follows
(paths relative to the reference checkout root).

PARITY UNPINNED: the reference cannot be built here (Rust, no toolchain, no
lockfile) and its own tests assert no numeric values, so
there is no golden vector from the reference to pin this restatement against.
What *is* pinned: the output shapes / dtypes the reference scripts state
(`tests/cwt_test.py:49-60`, `tests/ssq_cwt_test.py:49-57`), and the FFT, which
is the only third-party arithmetic (crate `rustfft` ^6.2, unnormalised
forward/inverse DFT) and is restated with ``numpy.fft`` (pocketfft), itself
cross-checked against a direct O(n^2) DFT in ``tests/test_oracle.py``.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np

EPS64 = 2.2204460492503131e-16
DEFAULT_GAMMA = 10.0 * EPS64          # ssq_stft.rs:258-261, ssq_cwt.rs:438-441
TWO_PI_LITERAL = 6.283185307179586    # ssq_stft.rs:32


class RustPanic(BaseException):
    """Stands for a Rust panic (surfaces in Python as pyo3_runtime.PanicException,
    which derives from BaseException)."""


# ----------------------------------------------------------------------------
# helpers shared by the STFT family   (rust/src/spectral/stft_utils.rs)
# ----------------------------------------------------------------------------
def stft_pad(x: np.ndarray, n_fft: int, padtype: str) -> np.ndarray:
    """stft_utils.rs:19-49 (pad_reflect) / :52-65 (pad_zeros).

    Total pad n_fft-1, left = (n_fft-1)//2, right = the rest.  Mirror indices
    falling outside [0, n) leave zeros (`if mirror_idx < n`; the right side's
    `n - 2 - i` wraps in release builds and then fails the same test).
    Unknown padtype -> reflect (stft.rs:28, ssq_stft.rs:127).
    """
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    pad = n_fft - 1
    pl = pad // 2
    pr = pad - pl
    p = np.zeros(n + pad, dtype=np.float64)
    p[pl:pl + n] = x
    if padtype != "zero":
        i = np.arange(pl)
        m = pl - i
        ok = m < n
        p[i[ok]] = x[m[ok]]
        i = np.arange(pr)
        m = n - 2 - i
        ok = (m >= 0) & (m < n)
        p[n + pl + i[ok]] = x[m[ok]]
    return p


def stft_frames(n_padded: int, n_fft: int, hop: int) -> int:
    """stft.rs:33 / ssq_stft.rs:183 (usize arithmetic; underflow or /0 panics)."""
    if hop <= 0:
        raise RustPanic("attempt to divide by zero")
    if n_padded < n_fft:
        raise RustPanic("attempt to subtract with overflow")
    return (n_padded - n_fft) // hop + 1


def size_window(window: np.ndarray, n_fft: int) -> np.ndarray:
    """ssq_stft.rs:104-119: centre zero-pad (left=(n_fft-L)//2) or centre crop."""
    w = np.asarray(window, dtype=np.float64)
    L = w.shape[0]
    if L < n_fft:
        pl = (n_fft - L) // 2
        out = np.zeros(n_fft, dtype=np.float64)
        out[pl:pl + L] = w
        return out
    if L > n_fft:
        s = (L - n_fft) // 2
        return w[s:s + n_fft].copy()
    return w.copy()


def diff_window(win: np.ndarray) -> np.ndarray:
    """ssq_stft.rs:131-179: g' = Re(IFFT(FFT(g) * i*xi)) / n with
    xi_k = 2*pi*k~/n, k~ = k (k <= n/2) else k-n.  The Nyquist term is kept."""
    n = win.shape[0]
    freqs = np.zeros(n, dtype=np.float64)
    k = np.arange(n)
    freqs[: n // 2 + 1] = k[: n // 2 + 1]
    freqs[n // 2 + 1:] = k[n // 2 + 1:] - float(n)
    freqs = freqs * (2.0 * math.pi / float(n))          # :146-148
    W = np.fft.fft(win.astype(np.complex128))
    W = (-W.imag * freqs) + 1j * (W.real * freqs)       # :163-167
    w = np.fft.ifft(W, norm="forward")                  # unnormalised inverse
    return w.real * (1.0 / float(n))                    # :173-176


def rust_linspace(a: float, b: float, n: int) -> np.ndarray:
    """ndarray::Array1::linspace: a + step*i, step=(b-a)/(n-1) (no endpoint fix-up)."""
    if n <= 0:
        return np.zeros(0)
    step = (b - a) / float(n - 1) if n > 1 else 0.0
    return a + step * np.arange(n, dtype=np.float64)


def _frame_matrix(padded: np.ndarray, n_fft: int, hop: int, n_frames: int) -> np.ndarray:
    idx = (np.arange(n_frames) * hop)[:, None] + np.arange(n_fft)[None, :]
    return padded[idx]                                   # [n_frames, n_fft]


# ----------------------------------------------------------------------------
# _rs.stft          (rust/src/spectral/stft.rs:12-95)
# ----------------------------------------------------------------------------
def stft(x, n_fft: int, hop_length: int, window, padtype: str):
    x = np.asarray(x, dtype=np.float64)
    window = np.asarray(window, dtype=np.float64)
    if x.shape[0] == 0:
        raise RustPanic("empty input")
    padded = stft_pad(x, n_fft, padtype)                 # :25-29
    n_frames = stft_frames(padded.shape[0], n_fft, hop_length)   # :32-33
    n_freqs = n_fft // 2 + 1                             # :34
    if window.shape[0] != n_fft:
        # apply_window truncates to the shorter (stft_utils.rs:8) and rustfft then
        # panics on the buffer/plan length mismatch (stft.rs:67).
        raise RustPanic("FFT buffer length does not match plan length")
    fr = _frame_matrix(padded, n_fft, hop_length, n_frames) * window[None, :]   # :58
    S = np.fft.fft(fr, axis=1)[:, :n_freqs]              # :67-74
    Sx = np.ascontiguousarray(S.T)                       # [n_freqs, n_frames] :81-85
    freqs = rust_linspace(0.0, 0.5, n_freqs)             # :40
    return Sx, freqs


# ----------------------------------------------------------------------------
# _rs.ssq_stft      (rust/src/spectral/ssq_stft.rs)
# ----------------------------------------------------------------------------
def phase_stft(Sx, dSx, Sfs, gamma):
    """ssq_stft.rs:11-39."""
    a, b = dSx.real, dSx.imag
    c, d = Sx.real, Sx.imag
    with np.errstate(all="ignore"):
        pd = (b * c - a * d) / ((c * c + d * d) * TWO_PI_LITERAL)
        w = np.abs(Sfs[:, None] - pd)
        w = np.where(np.hypot(c, d) < gamma, np.inf, w)  # Complex::norm() = hypot
    return w


def stft_ssq_freqs(n_freqs: int, fs: float) -> np.ndarray:
    """ssq_stft.rs:42-54: (i*0.5*fs)/(n_freqs-1) -- rounds differently from Sfs."""
    i = np.arange(n_freqs, dtype=np.float64)
    with np.errstate(all="ignore"):
        return (i * 0.5 * fs) / (float(n_freqs) - 1.0)


def nearest_bin_first_min(w: np.ndarray, freqs: np.ndarray) -> np.ndarray:
    """ssq_stft.rs:280-289: k = first argmin_idx |w - freqs[idx]| with strict `<`
    (NaN distance never wins -> k stays 0; +inf never reaches here).

    Exact but O(1) per element: a candidate from w/dw, then the reference's own
    distance expression on candidates-1..+1 picks the first minimum.  Verified
    against the brute-force scan in tests (`nearest_bin_bruteforce`)."""
    n = freqs.shape[0]
    dw = freqs[1] - freqs[0]
    with np.errstate(all="ignore"):
        est = np.where(np.isfinite(w), w / dw, 0.0)
    est = np.clip(np.nan_to_num(est, nan=0.0), 0, n - 1)
    c = np.rint(est).astype(np.int64)
    best_k = np.zeros(w.shape, dtype=np.int64)
    best_d = np.full(w.shape, np.inf)
    # scan candidates in ascending index so that ties keep the lower index
    for off in (-2, -1, 0, 1, 2):
        kk = np.clip(c + off, 0, n - 1)
        with np.errstate(all="ignore"):
            dist = np.abs(w - freqs[kk])
        better = dist < best_d
        # a candidate equal to an earlier (lower or same) index must not win on ties:
        best_k = np.where(better, kk, best_k)
        best_d = np.where(better, dist, best_d)
    best_k = np.where(np.isnan(w), 0, best_k)
    # Beyond the last bin fl(w - f_k) can round to the same value for several k
    # (huge w): the scan's strict `<` then keeps the *first* such k.  Resolve those
    # by the literal scan, chunked.
    far = np.isfinite(w) & (w > freqs[n - 1])
    if far.any():
        wf = w[far]
        out = np.empty(wf.shape[0], dtype=np.int64)
        step = max(1, (1 << 22) // n)
        for s in range(0, wf.shape[0], step):
            d = np.abs(wf[s:s + step, None] - freqs[None, :])
            out[s:s + step] = np.argmin(d, axis=1)       # first occurrence of the min
        best_k[far] = out
    return best_k


def nearest_bin_bruteforce(w: np.ndarray, freqs: np.ndarray) -> np.ndarray:
    """Literal restatement of the scan at ssq_stft.rs:280-289 (small inputs only)."""
    flat = w.reshape(-1)
    out = np.zeros(flat.shape[0], dtype=np.int64)
    for t, wv in enumerate(flat):
        k = 0
        min_dist = math.inf
        for idx, f in enumerate(freqs):
            dist = abs(wv - f)
            if dist < min_dist:
                min_dist = dist
                k = idx
        out[t] = k
    return out.reshape(w.shape)


def ssq_stft(x, window, n_fft: Optional[int] = None, win_len: Optional[int] = None,
             hop_len: int = 1, fs: float = 1.0, padtype: str = "reflect",
             squeezing: str = "sum", gamma: Optional[float] = None,
             return_intermediates: bool = False):
    """ssq_stft.rs:72-313."""
    x = np.asarray(x, dtype=np.float64)
    window = np.asarray(window, dtype=np.float64)
    n = x.shape[0]
    n_fft = min(n, 512) if n_fft is None else int(n_fft)                 # :92
    win_len = window.shape[0] if win_len is None else int(win_len)       # :93
    if win_len > n_fft:                                                  # :96-101
        raise ValueError(
            f"Window length {win_len} cannot be greater than n_fft {n_fft}")
    if n == 0 or n_fft == 0:
        raise RustPanic("empty input")
    win = size_window(window, n_fft)                                     # :104-119
    padded = stft_pad(x, n_fft, padtype)                                 # :124-128
    dwin = diff_window(win)                                              # :131-179
    n_frames = stft_frames(padded.shape[0], n_fft, hop_len)              # :182-183
    n_freqs = n_fft // 2 + 1                                             # :184
    if n_freqs < 2:
        raise RustPanic("index out of bounds: ssq_freqs[1]")             # :273
    fr = _frame_matrix(padded, n_fft, hop_len, n_frames)
    Sx = np.ascontiguousarray(np.fft.fft(fr * win[None, :], axis=1)[:, :n_freqs].T)       # :202,226
    dSx = np.ascontiguousarray(
        np.fft.fft((fr * dwin[None, :]) * fs, axis=1)[:, :n_freqs].T)                     # :208,227
    Sfs = rust_linspace(0.0, 0.5 * fs, n_freqs)                          # :255
    g = DEFAULT_GAMMA if gamma is None else float(gamma)                 # :258-261
    w = phase_stft(Sx, dSx, Sfs, g)                                      # :264
    ssq_freqs = stft_ssq_freqs(n_freqs, fs)                              # :267
    dw = ssq_freqs[1] - ssq_freqs[0]                                     # :273
    k = nearest_bin_first_min(w, ssq_freqs)                              # :280-289
    Tx = np.zeros((n_freqs, n_frames), dtype=np.complex128)
    cols = np.arange(n_frames)
    lebesgue = squeezing == "lebesgue"
    for i in range(n_freqs):                                             # rows ascending = reference order
        m = ~np.isinf(w[i])                                              # :278
        if not m.any():
            continue
        if lebesgue:
            wt = np.full(n_frames, 1.0 / float(n_freqs), dtype=np.complex128)   # :294
        else:
            wt = Sx[i]                                                   # :293,:295
        contrib = (wt.real * dw) + 1j * (wt.imag * dw)                   # Complex*f64
        Tx[k[i, m], cols[m]] += contrib[m]                               # :298
    if return_intermediates:
        return Tx, ssq_freqs, dict(Sx=Sx, dSx=dSx, w=w, k=k, Sfs=Sfs, dw=dw, window=win,
                                   diff_window=dwin)
    return Tx, ssq_freqs


# ----------------------------------------------------------------------------
# CWT family helpers
# ----------------------------------------------------------------------------
def next_power_of_2(n: int) -> int:
    """utils/array.rs:9-11: 1 << ceil(log2(n)) (f64 log2; n=0 -> -inf -> cast 0 -> 1)."""
    if n <= 0:
        return 1
    return 1 << int(math.ceil(math.log2(float(n))))


def cwt_pad(x: np.ndarray, pad_len: int, padtype: str) -> np.ndarray:
    """utils/array.rs:52-82 (pad_reflect) / :85-98 (pad_zero)."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    if pad_len < n:
        raise RustPanic("attempt to subtract with overflow")
    pad = pad_len - n
    pl = pad // 2
    pr = pad - pl
    p = np.zeros(pad_len, dtype=np.float64)
    p[pl:pl + n] = x
    if padtype != "zero":
        i = np.arange(pl)
        m = pl - i
        ok = m < n
        p[i[ok]] = x[m[ok]]
        i = np.arange(pr)
        m = n - 2 - i
        ok = (m >= 0) & (m < n)
        p[n + pl + i[ok]] = x[m[ok]]
    return p


def xifn(scale: float, n: int) -> np.ndarray:
    """wavelets/base.rs:18-33."""
    h = scale * (2.0 * math.pi) / float(n)
    xi = np.zeros(n, dtype=np.float64)
    i = np.arange(n)
    xi[: n // 2 + 1] = i[: n // 2 + 1].astype(np.float64) * h
    xi[n // 2 + 1:] = (i[n // 2 + 1:] - n).astype(np.float64) * h
    return xi


def wavelet_fourier(xi: np.ndarray, scale: float, wavelet: str) -> np.ndarray:
    """cwt.rs:492-547 (dup cwt_simd.rs:548-613).  Real-valued (imag = 0)."""
    w = scale * xi
    psih = np.zeros(xi.shape[0], dtype=np.float64)
    if wavelet == "morlet":
        mu = 6.0
        norm = math.pow(math.pi, -0.25) * math.sqrt(2.0)     # :501 (SQRT_2 constant)
        k_exp = math.exp(-0.5 * mu * mu)                     # :502
        m = w >= 0.0                                         # :512
        wm = w[m]
        t1 = np.exp(-0.5 * ((wm - mu) * (wm - mu)))          # :515 powi(2)
        t2 = k_exp * np.exp(-0.5 * (wm * wm))                # :516
        psih[m] = norm * (t1 - t2)                           # :518
    else:                                                    # "gmw" | _  :522
        gamma, beta = 3.0, 60.0
        m = w > 0.0                                          # :536
        wm = w[m]
        with np.errstate(all="ignore"):
            term = np.exp(beta * np.log(wm) - np.power(wm, gamma))   # :538-539
        psih[m] = 2.0 * term                                 # :540
    return psih


def log_scales(N: int, nv: int, simd_variant: bool = False) -> np.ndarray:
    """cwt.rs:461-489 / ssq_cwt.rs:300-326; cwt_simd.rs:474-545 uses exp(p*ln2)
    instead of 2^p when there are >= 16 scales."""
    log_min = math.log2(2.0)
    log_max = math.log2(float(N) * 0.5) if N > 0 else -math.inf
    num_octaves = log_max - log_min
    v = math.ceil(num_octaves * float(nv))
    num = int(v) if (math.isfinite(v) and v > 0) else 0      # `as usize` saturates
    sf = (log_max - log_min) / float(num - 1) if num > 1 else 0.0
    out = np.zeros(num, dtype=np.float64)
    for i in range(num):
        p = log_min + float(i) * sf
        if simd_variant and num >= 16:
            out[i] = math.exp(p * math.log(2.0))
        else:
            out[i] = math.pow(2.0, p)
    return out


def _dt_from(fs, t) -> float:
    """cwt.rs:66-76 / ssq_cwt.rs:283-293."""
    if t is not None:
        t = np.asarray(t, dtype=np.float64)
        if t.shape[0] < 2:
            raise ValueError("Time vector must have at least 2 elements")
        return float(t[1] - t[0])
    if fs is not None:
        return 1.0 / float(fs)
    return 1.0


def _cwt_core(x, scales, wavelet, dt, padtype, want_d: bool):
    """Shared by cwt.rs:85-105,169-326 and ssq_cwt.rs:329-431: returns padded
    (Wx, dWx|None) already multiplied by 1/P, plus (P, n1)."""
    N = x.shape[0]
    P = next_power_of_2(N + N // 2)                          # cwt.rs:87
    padded = cwt_pad(x, P, padtype)                          # :88-92
    xh = np.fft.fft(padded.astype(np.complex128))            # :147-162
    n1 = (P - N) // 2                                        # :98
    xi = xifn(1.0, P)                                        # :189
    na = scales.shape[0]
    Wx = np.zeros((na, P), dtype=np.complex128)
    dWx = np.zeros((na, P), dtype=np.complex128) if want_d else None
    norm = 1.0 / float(P)                                    # :251
    for i in range(na):
        psih = wavelet_fourier(xi, float(scales[i]), wavelet)
        r = np.fft.ifft(xh * psih, norm="forward")           # unnormalised inverse
        Wx[i] = (r.real * norm) + 1j * (r.imag * norm)
        if want_d:
            dpsih = (psih + 0j) * (1j * (xi / dt))           # :205-208 Complex*Complex(0, w/dt)
            r = np.fft.ifft(xh * dpsih, norm="forward")
            dWx[i] = (r.real * norm) + 1j * (r.imag * norm)
    return Wx, dWx, P, n1


def cwt(x, wavelet: str = "gmw", scales=None, fs=None, t=None, nv: int = 32,
        l1_norm: bool = True, derivative: bool = False, padtype: str = "reflect",
        rpadded: bool = False, vectorized: bool = True, patience: int = 0,
        _simd_variant: bool = False):
    """cwt.rs:46-144 (always a 3-tuple (Wx, scales, dWx|None))."""
    x = np.asarray(x, dtype=np.float64)
    N = x.shape[0]
    dt = _dt_from(fs, t)
    if scales is None:
        scales_a = log_scales(N, nv, simd_variant=_simd_variant)         # :79-82
    else:
        scales_a = np.asarray(scales, dtype=np.float64).copy()
    Wx, dWx, P, n1 = _cwt_core(x, scales_a, wavelet, dt, padtype, derivative)
    if not l1_norm:                                                      # :253-262
        sf = np.sqrt(scales_a)[:, None]
        Wx = (Wx.real * sf) + 1j * (Wx.imag * sf)
        if dWx is not None:
            dWx = (dWx.real * sf) + 1j * (dWx.imag * sf)
    if not rpadded:                                                      # :108-129
        Wx = np.ascontiguousarray(Wx[:, n1:n1 + N])
        if dWx is not None:
            dWx = np.ascontiguousarray(dWx[:, n1:n1 + N])
    return Wx, scales_a, dWx


def cwt_simd(*args, **kw):
    """cwt_simd.rs:52-150: same numbers as `cwt` except automatic scale generation."""
    return cwt(*args, _simd_variant=True, **kw)


def phase_cwt(Wx, dWx, gamma):
    """ssq_cwt.rs:15-47."""
    a, b = dWx.real, dWx.imag
    c, d = Wx.real, Wx.imag
    with np.errstate(all="ignore"):
        pd = (b * c - a * d) / ((c * c + d * d) * 2.0 * math.pi)
        w = np.abs(pd)
        w = np.where(np.hypot(c, d) < gamma, np.inf, w)
    return w


def cwt_ssq_freqs(n: int, fmin: float, fmax: float, dist: str) -> np.ndarray:
    """ssq_cwt.rs:50-113 ("linear" or log2-spaced; unknown -> log)."""
    out = np.zeros(n, dtype=np.float64)
    if dist == "linear":
        step = (fmax - fmin) / float(n - 1) if n > 1 else 0.0
        for i in range(n):
            out[i] = fmin + float(i) * step
    else:
        lmin, lmax = math.log2(fmin), math.log2(fmax)
        sf = (lmax - lmin) / float(n - 1) if n > 1 else 0.0
        for i in range(n):
            out[i] = math.pow(2.0, lmin + float(i) * sf)
    return out


def rust_round(v: np.ndarray) -> np.ndarray:
    """f64::round: half away from zero (exact; no +0.5 trick)."""
    r = np.trunc(v)
    with np.errstate(invalid="ignore"):
        adj = np.where(np.abs(v - r) >= 0.5, np.sign(v), 0.0)
    return r + adj


def cwt_bins(w: np.ndarray, ssq_freqs: np.ndarray):
    """ssq_cwt.rs:135-196: returns (bin, valid); `as isize` saturates, NaN -> 0."""
    n = ssq_freqs.shape[0]
    is_log = bool(n > 1 and (ssq_freqs[1] / ssq_freqs[0] > 1.1))          # :135-139
    with np.errstate(all="ignore"):
        if is_log:
            lmin = math.log2(ssq_freqs[0])
            lstep = (math.log2(ssq_freqs[n - 1]) - lmin) / float(n - 1) if n > 1 else 1.0
            v = (np.log2(w) - lmin) / lstep                               # :175-176
        else:
            lin_min = float(ssq_freqs[0])
            lstep = (float(ssq_freqs[n - 1]) - lin_min) / float(n - 1) if n > 1 else 1.0
            v = (w - lin_min) / lstep                                     # :187
        r = rust_round(v)
    skip = np.isinf(w) | np.isnan(w)                                      # :167
    r = np.where(np.isnan(r), 0.0, r)                                     # NaN as isize = 0
    r = np.clip(r, -9.0e18, 9.0e18)                                       # saturating cast
    b = r.astype(np.int64)
    valid = (~skip) & (b >= 0) & (b < n)                                  # :177,:188
    return b, valid, is_log


def ssq_cwt(x, wavelet: str = "gmw", scales=None, fs=None, t=None,
            ssq_freqs: Optional[str] = None, nv: int = 32, padtype: str = "reflect",
            squeezing: str = "sum", maprange: str = "peak", difftype: str = "trig",
            gamma: Optional[float] = None, vectorized: bool = True, flipud: bool = True,
            return_intermediates: bool = False):
    """ssq_cwt.rs:244-493."""
    x = np.asarray(x, dtype=np.float64)
    N = x.shape[0]
    dt = _dt_from(fs, t)
    if scales is None:
        scales_a = log_scales(N, nv)                                      # :300-326
    else:
        scales_a = np.asarray(scales, dtype=np.float64).copy()
    na = scales_a.shape[0]
    if na == 0:
        raise RustPanic("index out of bounds: scales[len-1]")             # :459
    Wxp, dWxp, P, n1 = _cwt_core(x, scales_a, wavelet, dt, padtype, True)  # :329-431
    Wx = np.ascontiguousarray(Wxp[:, n1:n1 + N])                          # :434-435
    dWx = np.ascontiguousarray(dWxp[:, n1:n1 + N])
    g = DEFAULT_GAMMA if gamma is None else float(gamma)
    w = phase_cwt(Wx, dWx, g)                                             # :444
    dist = "log" if ssq_freqs is None else ssq_freqs                      # :447
    with np.errstate(all="ignore"):
        if maprange == "maximal":                                         # :450-455
            dT = float(N) * dt
            fmin, fmax = 1.0 / dT, 0.5 / dt
        else:                                                             # :456-460
            fmin, fmax = 1.0 / float(scales_a[-1]), 1.0 / float(scales_a[0])
    freqs = cwt_ssq_freqs(na, fmin, fmax, dist)                           # :464-469
    b, valid, is_log = cwt_bins(w, freqs)
    kk = (na - 1 - b) if flipud else b                                    # :180-184
    Tx = np.zeros((na, N), dtype=np.complex128)
    cols = np.arange(N)
    lebesgue = squeezing == "lebesgue"
    for i in range(na):                                                   # scales ascending = reference order
        m = valid[i]
        if not m.any():
            continue
        if lebesgue:
            Tx[kk[i, m], cols[m]] += 1.0 / float(na)                      # :201-204
        else:
            Tx[kk[i, m], cols[m]] += Wx[i, m]                             # :200,:205,:208
    if return_intermediates:
        return Tx, freqs, dict(Wx=Wx, dWx=dWx, w=w, bin=b, valid=valid, k=kk,
                               is_log=is_log, scales=scales_a, P=P, n1=n1)
    return Tx, freqs


def hello_from_bin() -> str:
    """lib.rs:16-19."""
    return "Hello from ssqueeze!"


# ----------------------------------------------------------------------------
# icwt and the wavelet helper functions (implemented in the reference but not
# registered in lib.rs:25-32; advertised by src/ssqueeze/_rs.pyi:61-132)
# ----------------------------------------------------------------------------
def icwt(Wx, wavelet: str = "gmw", scales=None, nv=None, one_int: bool = True, x_len=None, x_mean: float = 0.0,
         padtype: str = "reflect", rpadded: bool = False, l1_norm: bool = True) -> np.ndarray:
    """cwt.rs:550-718.  `nv`, `padtype`, `rpadded` are accepted and unused there."""
    Wx = np.asarray(Wx, dtype=np.complex128)
    n_scales, n_times = Wx.shape
    if scales is None:
        raise ValueError("Scales must be provided")                       # :571-575
    sc = np.asarray(scales, dtype=np.float64)
    adm = {"morlet": 0.776, "gmw": 1.0}.get(wavelet, 1.0)                 # :578-582
    x_length = n_times if x_len is None else int(x_len)                   # :586
    if x_length > n_times or sc.shape[0] < n_scales:
        raise RustPanic("index out of bounds")                            # Wx_array[[i, j]] / scales_array[i]
    if n_scales > 1 and sc[1] > sc[0]:                                    # :593-597, :704-708
        dj = math.log(sc[1] / sc[0])
    else:
        dj = 0.1
    final_norm = (2.0 / adm) * dj
    x = np.zeros(x_length, dtype=np.float64)
    if one_int:                                                           # :588-627
        for i in range(n_scales):                                         # per column: scales ascending (:620-622)
            nf = 1.0 if l1_norm else 1.0 / math.sqrt(sc[i])               # :605-609
            x += Wx[i, :x_length].real * nf
        return x * final_norm + x_mean                                    # :623
    xi = xifn(1.0, x_length)                                              # :633
    for i in range(n_scales):                                             # :636-693 (summed ascending, :696-700)
        psih = wavelet_fourier(xi, float(sc[i]), wavelet)
        tmp = np.fft.fft(Wx[i, :x_length]) * np.conj(psih + 0j)           # :659-666
        r = np.fft.ifft(tmp, norm="forward")                              # unnormalised inverse (:675)
        sn = 1.0 / sc[i] if l1_norm else 1.0 / (math.sqrt(sc[i]) ** 2)    # :679-683
        x += r.real * (1.0 / float(x_length)) * sn                        # :678, :685-687
    return x * final_norm + x_mean                                        # :710-713


def morlet(w, mu: float = 6.0, dtype: str = "float64") -> np.ndarray:
    """wavelets/morlet.rs:22-41 via :59-77.  No `w >= 0` gate here (the hot path's inline Morlet has one and a
    different normalisation, cwt.rs:497-520)."""
    w = np.asarray(w, dtype=np.float64)
    cs = (1.0 + math.exp(-(mu * mu)) - 2.0 * math.exp(-3.0 / 4.0 * (mu * mu))) ** (-0.5)     # :24
    ks = math.exp(-0.5 * (mu * mu))                                                          # :25
    factor = math.sqrt(2.0) * cs * math.pow(math.pi, 0.25)                                   # :33
    out = factor * (np.exp(-0.5 * ((w - mu) * (w - mu))) - ks * np.exp(-0.5 * (w * w)))      # :37-40
    return out.astype(np.complex128)


def morlet_freq(n: int = 1024, scale: float = 1.0, mu: float = 6.0, dtype: str = "float64") -> np.ndarray:
    """wavelets/morlet.rs:80-100."""
    return morlet(xifn(scale, n), mu, dtype)


def _time_from_freq(psih: np.ndarray) -> np.ndarray:
    """wavelets/morlet.rs:114-141 == gmw.rs:306-333: (-1)^i spectral reversal, halve the Nyquist bin when n is even,
    unnormalised inverse FFT, times 1/n."""
    n = psih.shape[0]
    p = psih * np.where(np.arange(n) % 2 == 0, 1.0, -1.0)
    if n % 2 == 0 and n > 0:
        p[n // 2] /= 2.0
    r = np.fft.ifft(p, norm="forward")
    return (r.real * (1.0 / n)) + 1j * (r.imag * (1.0 / n))


def morlet_time(n: int = 1024, scale: float = 1.0, mu: float = 6.0, dtype: str = "float64") -> np.ndarray:
    """wavelets/morlet.rs:103-145."""
    return _time_from_freq(morlet_freq(n, scale, mu, dtype))


def gamma_function(x: float) -> float:
    """wavelets/gmw.rs:172-201 (Lanczos, g = 7, 8 coefficients)."""
    if x < 0.5:
        return math.pi / (math.sin(math.pi * x) * gamma_function(1.0 - x))
    p = [676.5203681218851, -1259.1392167224028, 771.32342877765313, -176.61502916214059, 12.507343278686905,
         -0.13857109526572012, 9.9843695780195716e-6, 1.5056327351493116e-7]
    x = x - 1.0
    y = 0.99999999999980993
    for i in range(len(p)):
        y += p[i] / (x + float(i) + 1.0)
    t = x + float(len(p)) - 0.5
    return math.sqrt(2.0 * math.pi) * math.pow(t, x + 0.5) * math.exp(-t) * y


def _factorial(n: int) -> float:
    """gmw.rs:204-209."""
    r = 1.0
    for i in range(1, n + 1):
        r *= float(i)
    return r


def _binomial(n: int, k: int) -> float:
    """gmw.rs:212-232."""
    if k < 0 or k > n:
        return 0.0
    if k == 0 or k == n:
        return 1.0
    if n <= 20:
        return _factorial(n) / (_factorial(k) * _factorial(n - k))
    c = 0.0
    for i in range(1, k + 1):
        c += math.log(float(n - k + i)) - math.log(float(i))
    return math.exp(c)


def _trunc_i32(v: float) -> int:
    return int(v)            # `c as i32`: truncation toward zero


def gmw(w, gamma: float = 3.0, beta: float = 60.0, norm: str = "bandpass", order: int = 0,
        dtype: str = "float64") -> np.ndarray:
    """wavelets/gmw.rs:236-262: validation (:246-254), then GMW::psih."""
    if gamma <= 0.0:
        raise ValueError("gamma must be positive")
    if beta < 0.0:
        raise ValueError("beta must be non-negative")
    if order < 0:
        raise ValueError("order must be non-negative")
    return _gmw_psih(w, gamma, beta, norm, order)


def _gmw_psih(w, gamma: float, beta: float, norm: str, order: int) -> np.ndarray:
    """wavelets/gmw.rs:72-159 (norm is lower-cased, :25; anything but "bandpass" is the L2 branch)."""
    w = np.asarray(w, dtype=np.float64)
    bandpass = norm.lower() == "bandpass"
    wc = math.pow(beta / gamma, 1.0 / gamma)                              # :33-35
    r = (2.0 * beta + 1.0) / gamma                                        # :38-40
    out = np.zeros(w.shape[0], dtype=np.float64)
    pos = w > 0.0                                                         # :84, :100, :131
    wp = w[pos]
    with np.errstate(all="ignore"):
        if order == 0:
            if bandpass:
                nc = 2.0 / math.exp(beta * math.log(wc) - math.pow(wc, gamma)) if wc > 0 else float("nan")   # :46-50
                out[pos] = nc * np.exp(beta * np.log(wp) - np.power(wp, gamma))                              # :89-92
            else:
                nc = math.sqrt(2.0 * math.pi * gamma * math.pow(2.0, r) / gamma_function(r))                 # :52-54
                out[pos] = nc * np.power(wp, beta) * np.exp(-np.power(wp, gamma))                            # :105-108
        else:
            c = r - 1.0
            k = int(order)
            if bandpass:
                coeff = 2.0 * math.sqrt(gamma_function(r) * gamma_function(k + 1.0) / gamma_function(k + r))  # :119-123
            else:
                coeff = math.sqrt(2.0 * math.pi * gamma * math.pow(2.0, r) * gamma_function(k + 1.0) /
                                  gamma_function(k + r))                                                      # :125-127
            xx = 2.0 * np.power(wp, gamma)                                # :137
            lag = np.zeros_like(wp)
            ci = _trunc_i32(c)
            for m in range(k + 1):                                        # :59-68
                b = _binomial(k + ci + 1, ci + m + 1) * _binomial(k, m)
                lag = lag + b * (((-1.0) ** m) * np.power(xx, m) / _factorial(m))
            if bandpass:
                e = np.exp(-beta * math.log(wc) + math.pow(wc, gamma) + beta * np.log(wp) - np.power(wp, gamma))   # :141-145
                out[pos] = coeff * lag * e
            else:
                out[pos] = coeff * lag * np.power(wp, beta) * np.exp(-np.power(wp, gamma))                    # :149-153
    return out.astype(np.complex128)


def gmw_freq(n: int = 1024, scale: float = 1.0, gamma: float = 3.0, beta: float = 60.0, norm: str = "bandpass",
             order: int = 0, dtype: str = "float64") -> np.ndarray:
    """wavelets/gmw.rs:265-289 (no parameter validation on this entry)."""
    return _gmw_psih(xifn(scale, n), gamma, beta, norm, order)


def gmw_time(n: int = 1024, scale: float = 1.0, gamma: float = 3.0, beta: float = 60.0, norm: str = "bandpass",
             order: int = 0, dtype: str = "float64") -> np.ndarray:
    """wavelets/gmw.rs:292-337."""
    return _time_from_freq(gmw_freq(n, scale, gamma, beta, norm, order, dtype))


def gmw_center_frequency(gamma: float = 3.0, beta: float = 60.0, kind: str = "peak") -> float:
    """wavelets/gmw.rs:340-357."""
    if kind == "peak":
        return math.pow(beta / gamma, 1.0 / gamma)
    if kind == "energy":
        return (1.0 / math.pow(2.0, 1.0 / gamma)) * (gamma_function((2.0 * beta + 2.0) / gamma) /
                                                      gamma_function((2.0 * beta + 1.0) / gamma))
    raise ValueError(f"Unknown center frequency kind: {kind}")


# ----------------------------------------------------------------------------
# synthetic workloads -- shared by tests and bench
# ----------------------------------------------------------------------------
from ssqueeze_rs_amd.synth import synth_signal  # noqa: E402,F401  (the workload generator lives with the package; re-exported for the tests)


# ----------------------------------------------------------------------------
# fp32-mode semantics of the HIP kernels (an extension: the reference is fp64 only)
# ----------------------------------------------------------------------------
def stft_bins_f32_model(w32: np.ndarray, dw: float, n_freqs: int) -> np.ndarray:
    """Bin index the fp32 kernels define for a given fp32 `w` (stft_kernels.h phase_bin):
    u = fma(w, fl32(1/dw), -0.5); k = ceil(u) clamped to [0, n_freqs-1]; NaN -> 0.
    Exact half-bin ties go to the lower bin, like the reference scan's strict `<`."""
    w32 = np.asarray(w32, dtype=np.float32)
    inv = np.float32(1.0 / dw)
    with np.errstate(all="ignore"):
        u = (w32.astype(np.float64) * np.float64(inv) - 0.5).astype(np.float32)   # one rounding = fma
        k = np.where(u >= np.float32(n_freqs - 1), n_freqs - 1, np.ceil(u))
    k = np.where(np.isnan(w32), 0, k)
    return np.nan_to_num(k, nan=0.0, posinf=n_freqs - 1).astype(np.int64)


def accumulate_tx(Sx: np.ndarray, k: np.ndarray, keep: np.ndarray, dw: float, n_out: int,
                  lebesgue: bool = False) -> np.ndarray:
    """Tx[k[i,j], j] += Sx[i,j]*dw over kept bins, rows ascending (ssq_stft.rs:276-301)."""
    n_rows, n_cols = Sx.shape
    Tx = np.zeros((n_out, n_cols), dtype=np.complex128)
    cols = np.arange(n_cols)
    for i in range(n_rows):
        m = keep[i]
        if not m.any():
            continue
        wt = np.full(n_cols, 1.0 / float(n_rows), dtype=np.complex128) if lebesgue else Sx[i].astype(np.complex128)
        Tx[k[i, m], cols[m]] += (wt.real[m] * dw) + 1j * (wt.imag[m] * dw)
    return Tx
