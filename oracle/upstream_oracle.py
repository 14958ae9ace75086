"""NumPy restatement of the vendored UPSTREAM variant (ssqueezepy, /root/reference/old/ssqueezepy) -- SURVEY 8(f)-4.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing else); the product never loads it.

**Parity unpinned.**  Upstream does not import here (`numba` missing: an ordinary ModuleNotFoundError at
old/ssqueezepy/algos.py:5), so this file is a numba-free restatement read off the source, pinned only by the facts
upstream's own tests state: the reconstruction thresholds of old/tests/reconstruction_test.py:111-123 (cwt / icwt /
issq_cwt, mad_rms < 0.02 on echirp(1024)) and :160-206 (stft -> istft MAE < 1e-14; ssq_stft -> issq_stft MAE < 1e-1),
which tests/test_upstream_oracle.py checks on this restatement and tests/test_gpu_upstream.py on the HIP path.

What differs from the Rust path (oracle/ssq_oracle.py), each cited below: pad split (`common.py:115-120`), modulated
frames (`stft_utils.py:70-83`), Nyquist term of the diff-window zeroed (`_stft.py:297-298`), `Sfs = np.linspace`,
keep rule `|Sx| > gamma`, bins `min(round(max((w - v0)/dv, 0)), n-1)` with round-half-even and clamping instead of a
first-minimum scan / dropping (`algos.py:957-984`, `:899-910`), CWT padding `p2up` (`common.py:32-51`), normalised
wavelets with the Nyquist bin halved (`wavelets.py:62-95`, `_gmw.py:204-210`, `wavelets.py:497-523`), `ssq_freqs` from
the wavelet's peak centre frequency (`ssqueezing.py:218-244`), the constant `ln 2 / nv` (`ssqueezing.py:122-128`).
Supported subset: explicit `scales` arrays that are exponentially spaced ('log' scaletype); wavelets 'gmw'
(gamma, beta; bandpass norm, order 0) and 'morlet' (mu); difftype 'trig'; squeezing 'sum' / 'lebesgue'.
"""
from __future__ import annotations

import math

import numpy as np

EPS64 = float(np.finfo(np.float64).eps)
EPS32 = float(np.finfo(np.float32).eps)
PI = math.pi


# ---------------------------------------------------------------------------------------------- shared helpers ----
def p2up(n: int):
    """utils/common.py:32-51: next power of two by ROUNDING log2, left pad the larger half."""
    up = int(2 ** (1 + np.round(np.log2(n))))
    n2 = int((up - n) // 2)
    n1 = int(up - n - n2)
    return up, n1, n2


def padsignal(x, padtype="reflect", padlength=None):
    """utils/common.py:54-158 ('reflect' and 'zero'); returns (xp, n_up, n1, n2)."""
    x = np.asarray(x)
    N = x.shape[-1]
    if padlength is None:
        n_up, n1, n2 = p2up(N)
    else:
        n_up = int(padlength)
        if abs(padlength - N) % 2 == 0:                     # :111-116  even: left = right, odd: left = right + 1
            n1 = n2 = (n_up - N) // 2
        else:
            n2 = (n_up - N) // 2
            n1 = n2 + 1
    if padtype == "zero":
        xp = np.pad(x, (n1, n2))
    elif padtype == "reflect":
        xp = np.pad(x, (n1, n2), mode="reflect")
    else:
        raise ValueError(f"padtype {padtype!r} is outside the supported subset ('reflect', 'zero')")
    return xp, n_up, n1, n2


def xifn(scale: float, N: int) -> np.ndarray:
    """wavelets.py:473-483."""
    xi = np.zeros(N)
    h = scale * (2 * PI) / N
    for i in range(N // 2 + 1):
        xi[i] = i * h
    for i in range(N // 2 + 1, N):
        xi[i] = (i - N) * h
    return xi


# --------------------------------------------------------------------------------------------------- wavelets ----
def morsefreq(gamma: float, beta: float) -> float:
    """_gmw.py:611-657 (first output, beta > 0): the peak frequency (beta/gamma)^(1/gamma)."""
    return float(np.exp((1.0 / gamma) * (np.log(beta) - np.log(gamma))))


def gmw_l1(w, gamma=3.0, beta=60.0):
    """_gmw.py:187-210 (bandpass norm, order 0): 2 exp(-beta ln wc + wc^gamma + beta ln w - w^gamma), 0 for w < 0."""
    w = np.array(w, dtype=np.float64, copy=True)
    wc = morsefreq(gamma, beta)
    wcl = np.log(wc)
    nonneg = (w >= 0)
    w = w * nonneg
    with np.errstate(divide="ignore", invalid="ignore"):
        out = 2 * np.exp(-beta * wcl + wc ** gamma + beta * np.log(w) - w ** gamma) * nonneg
    return out


def morlet(w, mu=13.4):
    """wavelets.py:497-523: sqrt(2) cs pi^(1/4) (exp(-(w - mu)^2/2) - ks exp(-w^2/2)) -- NOT cut at w < 0."""
    w = np.asarray(w, dtype=np.float64)
    cs = (1 + np.exp(-mu ** 2) - 2 * np.exp(-3 / 4 * mu ** 2)) ** (-.5)
    ks = np.exp(-.5 * mu ** 2)
    return np.sqrt(2) * cs * PI ** .25 * (np.exp(-.5 * (w - mu) ** 2) - ks * np.exp(-.5 * w ** 2))


def wavelet_fn(wavelet):
    """('gmw', {'gamma':, 'beta':}) / ('morlet', {'mu':}) / plain names -> psih(w) (wavelets.py:409-470 subset)."""
    name, kw = (wavelet, {}) if isinstance(wavelet, str) else wavelet
    if name == "gmw":
        g, b = float(kw.get("gamma", 3.0)), float(kw.get("beta", 60.0))
        return lambda w: gmw_l1(w, g, b)
    if name == "morlet":
        mu = float(kw.get("mu", 13.4))
        return lambda w: morlet(w, mu)
    raise ValueError(f"wavelet {name!r} is outside the supported subset ('gmw', 'morlet')")


def psih_at_scale(fn, scale: float, N: int) -> np.ndarray:
    """Wavelet.__call__(scale=, nohalf=False) (wavelets.py:62-95): psih(scale * xi) with the Nyquist bin halved."""
    psih = np.array(fn(scale * xifn(1.0, N)), dtype=np.float64)
    if N % 2 == 0:
        psih[N // 2] /= 2
    return psih


def center_frequency_peak(fn, scale: float, N: int) -> float:
    """wavelets.center_frequency(kind='peak') (wavelets.py:691-716): w[argmax |psih(scale w)|^2] on the centred grid."""
    w = _aifftshift(xifn(1.0, N))
    psih = fn(scale * w)
    return float(w[np.argmax(np.abs(psih) ** 2)])


def _aifftshift(xh):
    """wavelets.py:950-962: moves the left N//2 + 1 bins to the right (even N), ifftshift for odd N."""
    N = len(xh)
    if N % 2 != 0:
        return np.fft.ifftshift(xh)
    out = np.zeros(N, dtype=xh.dtype)
    out[N // 2 - 1:] = xh[:N // 2 + 1]
    out[:N // 2 - 1] = xh[N // 2 + 1:]
    return out


def _min_neglect_idx(arr, th=1e-12):
    """algos.py:616-622."""
    for i, x in enumerate(arr):
        if x < th:
            return i
    return i


def integrate_analytic(int_fn):
    """utils/cwt_utils.py:583-627 (trapezoid on the stated grids; the non-convergent warning branch is not needed for
    the supported wavelets)."""
    def est(mxlim, n):
        t = np.linspace(mxlim, .1, n, endpoint=False)[::-1].copy()
        arr = int_fn(t)
        mi = int(np.argmax(arr))
        return arr, t, _min_neglect_idx(np.abs(arr[mi:]), th=1e-15) + mi

    t0 = np.logspace(-15, -1, 1000)
    int_nz = np.trapezoid(int_fn(t0), t0) if hasattr(np, "trapezoid") else np.trapz(int_fn(t0), t0)
    for m, mxlim in zip([1, 1, 4, 8], [1, 20, 80, 160]):
        arr, t, mni = est(mxlim, 10000 * m)
        if (len(t) - mni > 1000 * m) and np.sum(np.abs(arr)) > 1e-5:
            break
    arr, t = arr[:mni], t[:mni]
    body = np.trapezoid(arr, t) if hasattr(np, "trapezoid") else np.trapz(arr, t)
    return body + int_nz


def adm_ssq(wavelet) -> float:
    """utils/cwt_utils.py:28-47: integral of conj(psih(w)) / w over (0, inf)."""
    fn = wavelet_fn(wavelet)
    return float(np.real(integrate_analytic(lambda w: np.conj(fn(w)) / w)))


def adm_cwt(wavelet) -> float:
    """utils/cwt_utils.py:50-63."""
    fn = wavelet_fn(wavelet)
    return float(np.real(integrate_analytic(lambda w: np.conj(fn(w)) * fn(w) / w)))


# ------------------------------------------------------------------------------------------------- STFT family ----
def get_window(window, win_len, n_fft=None, derivative=False):
    """_stft.py:257-309 for an ndarray window: centre-pad to n_fft; diff-window by frequency-domain
    differentiation with the Nyquist term ZEROED for even length (:293-299)."""
    window = np.asarray(window, dtype=np.float64)
    if n_fft is None:
        pl = pr = 0
    else:
        if win_len > n_fft:
            raise ValueError("Can't have `win_len > n_fft` ({} > {})".format(win_len, n_fft))
        pl = (n_fft - win_len) // 2
        pr = n_fft - win_len - pl
    if len(window) < (win_len + pl + pr):
        window = np.pad(window, [pl, pr])
    if not derivative:
        return window
    wf = np.fft.fft(window)
    Nw = len(window)
    xi = xifn(1, Nw)
    if Nw % 2 == 0:
        xi[Nw // 2] = 0
    return window, np.fft.ifft(wf * 1j * xi).real


def buffer(x, seg_len, n_overlap, modulated=False):
    """utils/stft_utils.py:20-83."""
    hop = seg_len - n_overlap
    n_segs = (len(x) - seg_len) // hop + 1
    s20 = int(np.ceil(seg_len / 2))
    s21 = s20 - 1 if (seg_len % 2 == 1) else s20
    out = np.zeros((seg_len, n_segs), dtype=x.dtype)
    for i in range(n_segs):
        if not modulated:
            out[:, i] = x[hop * i: hop * i + seg_len]
        else:
            s0 = hop * i
            e0 = s0 + s21
            out[:s20, i] = x[e0:e0 + s20]
            out[s20:, i] = x[s0:e0]
    return out


def stft(x, window, n_fft=None, win_len=None, hop_len=1, fs=1.0, padtype="reflect", modulated=True,
         derivative=False):
    """_stft.py:13-193 (ndarray window).  Returns Sx or (Sx, dSx), each [n_fft//2 + 1, (N - 1)//hop_len + 1]."""
    x = np.asarray(x, dtype=np.float64)
    N = len(x)
    n_fft = n_fft or min(N // hop_len, 512)
    if win_len is None:
        win_len = len(window)
    window, diff_window = get_window(window, win_len, n_fft, derivative=True)
    xp, *_ = padsignal(x, padtype, padlength=N + n_fft - 1)                  # :170-171
    Sx = buffer(xp, n_fft, n_fft - hop_len, modulated)
    dSx = Sx.copy() if derivative else None
    if modulated:                                                            # :132-135
        window = np.fft.ifftshift(window)
        diff_window = np.fft.ifftshift(diff_window) * fs
    elif derivative:
        # upstream multiplies by fs only inside the `modulated` branch (:134-135); reproduced as written
        pass
    Sx = np.fft.rfft(Sx * window.reshape(-1, 1), axis=0)
    if derivative:
        dSx = np.fft.rfft(dSx * diff_window.reshape(-1, 1), axis=0)
        return Sx, dSx
    return Sx


def istft(Sx, window, n_fft=None, win_len=None, hop_len=1, N=None, modulated=True, win_exp=1):
    """_stft.py:196-254."""
    n_fft = n_fft or (Sx.shape[0] - 1) * 2
    win_len = win_len or n_fft
    N = N or hop_len * Sx.shape[1]
    window = get_window(window, win_len, n_fft=n_fft)
    xbuf = np.fft.irfft(Sx, n=n_fft, axis=0).real
    if modulated:
        xbuf = np.fft.fftshift(xbuf, axes=0)
    wpow = 1 if win_exp == 0 else (window if win_exp == 1 else window ** win_exp)      # stft_utils.py:141-165
    x = np.zeros(N + n_fft - 1)
    for i in range(xbuf.shape[1]):
        n = i * hop_len
        x[n:n + n_fft] += xbuf[:, i] * wpow
    wn = np.zeros(N + n_fft - 1)                                                        # :169-191
    wp = window ** (win_exp + 1)
    for i in range((len(wn) - n_fft) // hop_len + 1):
        n = i * hop_len
        wn[n:n + n_fft] += wp
    th = np.finfo(x.dtype).tiny
    if wn.min() < th:
        nz = wn > th
        x[nz] /= wn[nz]
    else:
        x /= wn
    return x[n_fft // 2: -((n_fft - 1) // 2)]


def phase_stft(Sx, dSx, Sfs, gamma):
    """algos.py:794-803."""
    with np.errstate(all="ignore"):
        A, B, C, D = dSx.real, dSx.imag, Sx.real, Sx.imag
        w = np.abs(Sfs[:, None] - (B * C - A * D) / ((C ** 2 + D ** 2) * 6.283185307179586))
    return np.where(np.abs(Sx) < gamma, np.inf, w)


def _bins_lin(w, vmin, dv, omax):
    """algos.py:231-239 / :957-968: int(min(round(max((w - vmin)/dv, 0)), omax)) -- Python round = half to even."""
    with np.errstate(all="ignore"):
        return np.minimum(np.rint(np.maximum((w - vmin) / dv, 0)), omax).astype(np.int64)


def _bins_log(w, vlmin, dvl, omax):
    """algos.py:173-180 / :899-910."""
    with np.errstate(all="ignore"):
        return np.minimum(np.rint(np.maximum((np.log2(w) - vlmin) / dvl, 0)), omax).astype(np.int64)


def ssq_stft(x, window, n_fft=None, win_len=None, hop_len=1, fs=1.0, padtype="reflect", modulated=True,
             squeezing="sum", gamma=None, flipud=False, return_intermediates=False):
    """_ssq_stft.py:12-137 -> (Tx, Sx, ssq_freqs, Sfs); the fused loop algos.py:957-968."""
    Sx, dSx = stft(x, window, n_fft=n_fft, win_len=win_len, hop_len=hop_len, fs=fs, padtype=padtype,
                   modulated=modulated, derivative=True)
    n_rows = Sx.shape[0]
    Sfs = np.linspace(0, .5 * fs, n_rows)                                   # _ssq_stft.py:248-257
    if gamma is None:
        gamma = 10 * EPS64
    const = Sfs[1] - Sfs[0]                                                 # ssqueezing.py:129-130
    vmin, dv, omax = float(Sfs[0]), float(Sfs[1] - Sfs[0]), n_rows - 1
    with np.errstate(all="ignore"):
        A, B, C, D = dSx.real, dSx.imag, Sx.real, Sx.imag
        w = np.abs(Sfs[:, None] - (B * C - A * D) / ((C ** 2 + D ** 2) * 6.283185307179586))
    keep = np.abs(Sx) > gamma                                               # algos.py:960
    k = _bins_lin(w, vmin, dv, omax)
    if flipud:
        k = omax - k
    Wv = (np.ones(Sx.shape, dtype=Sx.dtype) / len(Sx)) if squeezing == "lebesgue" else Sx   # ssqueezing.py:183-184
    Tx = np.zeros(Sx.shape, dtype=np.complex128)
    cols = np.arange(Sx.shape[1])
    for i in range(n_rows):                                                 # rows ascending per column (:957-968)
        m = keep[i]
        np.add.at(Tx, (k[i, m], cols[m]), Wv[i, m] * const)
    ssq_freqs = Sfs[::-1] if flipud else Sfs                                # ssqueezing.py:199-205
    if return_intermediates:
        return Tx, Sx, ssq_freqs, Sfs, dict(dSx=dSx, w=np.where(keep, w, np.inf), k=np.where(keep, k, -1), const=const)
    return Tx, Sx, ssq_freqs, Sfs


def issq_stft(Tx, window, n_fft=None, win_len=None, hop_len=1, modulated=True):
    """_ssq_stft.py:139-198 (full inverse)."""
    if not modulated:
        raise ValueError("inversion with `modulated == False` is unsupported.")
    if hop_len != 1:
        raise ValueError("inversion with `hop_len != 1` is unsupported.")
    n_fft = n_fft or (Tx.shape[0] - 1) * 2
    win_len = win_len or n_fft
    window = get_window(window, win_len, n_fft=n_fft)
    x = Tx.real.sum(axis=0)
    return x * (2 / window[len(window) // 2])


# -------------------------------------------------------------------------------------------------- CWT family ----
def infer_nv(scales) -> int:
    """utils/cwt_utils.py:264-298 for an exponentially spaced array ('log' scaletype)."""
    s = np.asarray(scales, dtype=np.float64).reshape(-1)
    if np.mean(np.abs(np.diff(np.log(s), 2))) >= 4e-15 * 1e3:
        raise ValueError("`scales` must be exponentially spaced (the supported subset: scaletype 'log')")
    return int(np.round(1 / np.diff(np.log2(s))[0]))


def cwt(x, wavelet="gmw", scales=None, fs=1.0, l1_norm=True, derivative=False, padtype="reflect", rpadded=False):
    """_cwt.py:12-318 with an explicit scales array -> (Wx, scales[, dWx])."""
    x = np.asarray(x, dtype=np.float64)
    N = len(x)
    dt = 1.0 / fs
    fn = wavelet_fn(wavelet)
    xp, n_up, n1, _ = padsignal(x, padtype)                                  # :277-278
    xh = np.fft.fft(xp)
    scales = np.asarray(scales, dtype=np.float64).reshape(-1)
    xi = xifn(1.0, n_up)
    Wx = np.zeros((len(scales), n_up), dtype=np.complex128)
    dWx = np.zeros_like(Wx) if derivative else None
    for i, a in enumerate(scales):                                           # :177-197
        psih = psih_at_scale(fn, float(a), n_up)
        Wx[i] = np.fft.ifft(psih * xh)
        if derivative:
            dWx[i] = np.fft.ifft((1j * xi / dt) * psih * xh)
    if not rpadded:
        Wx = Wx[:, n1:n1 + N]
        if derivative:
            dWx = dWx[:, n1:n1 + N]
    if not l1_norm:                                                          # :305-308
        Wx = Wx * np.sqrt(scales)[:, None]
        if derivative:
            dWx = dWx * np.sqrt(scales)[:, None]
    return (Wx, scales, dWx) if derivative else (Wx, scales)


def cwt_ssq_freqs(scales, N, wavelet, dt=1.0, maprange="peak", scaletype="log"):
    """ssqueezing.py:218-290: [fm, fM] from the peak centre frequency at the last / first scale (padded length), or
    'maximal'; exponential ('log') or linear spacing."""
    fn = wavelet_fn(wavelet)
    na = len(scales)
    if maprange == "maximal":
        fm, fM = 1 / (dt * N), 1 / (2 * dt)
    else:
        Np = p2up(N)[0]
        fm = center_frequency_peak(fn, float(scales[-1]), Np) / (2 * PI) / dt
        fM = center_frequency_peak(fn, float(scales[0]), Np) / (2 * PI) / dt
    if scaletype == "log":
        return fm * np.power(fM / fm, np.arange(na) / (na - 1))
    return np.linspace(fm, fM, na)


def ssq_cwt(x, wavelet="gmw", scales=None, fs=1.0, ssq_freqs=None, padtype="reflect", squeezing="sum",
            maprange="peak", gamma=None, flipud=True, return_intermediates=False):
    """_ssq_cwt.py:12-311 (difftype 'trig', explicit exponential scales) -> (Tx, Wx, ssq_freqs, scales)."""
    x = np.asarray(x, dtype=np.float64)
    N = len(x)
    dt = 1.0 / fs
    scales = np.asarray(scales, dtype=np.float64).reshape(-1)
    nv = infer_nv(scales)
    Wx, _, dWx = cwt(x, wavelet, scales=scales, fs=fs, l1_norm=True, derivative=True, padtype=padtype)
    if gamma is None:
        gamma = 10 * EPS64
    scaletype = ssq_freqs if isinstance(ssq_freqs, str) else "log"
    freqs = cwt_ssq_freqs(scales, N, wavelet, dt, maprange, scaletype)
    const = np.log(2) / nv                                                   # ssqueezing.py:122-124
    na = len(scales)
    with np.errstate(all="ignore"):
        A, B, C, D = dWx.real, dWx.imag, Wx.real, Wx.imag
        w = np.abs((B * C - A * D) / ((C ** 2 + D ** 2) * 6.283185307179586))
    keep = np.abs(Wx) > gamma                                                # algos.py:902
    if scaletype == "log":
        vlmin = float(np.log2(freqs[0]))
        dvl = float(np.log2(freqs[1]) - np.log2(freqs[0]))                   # algos.py:356-363
        k = _bins_log(w, vlmin, dvl, na - 1)
    else:
        k = _bins_lin(w, float(freqs[0]), float(freqs[1] - freqs[0]), na - 1)
    if flipud:
        k = na - 1 - k
    Wv = (np.ones(Wx.shape, dtype=Wx.dtype) / len(Wx)) if squeezing == "lebesgue" else Wx
    Tx = np.zeros(Wx.shape, dtype=np.complex128)
    cols = np.arange(N)
    for i in range(na):
        m = keep[i]
        np.add.at(Tx, (k[i, m], cols[m]), Wv[i, m] * const)
    out_freqs = freqs[::-1] if flipud else freqs                             # ssqueezing.py:199-205 (cwt and not flipud
    if not flipud:                                                           #  -> reversed as well)
        out_freqs = freqs[::-1]
    if return_intermediates:
        return Tx, Wx, out_freqs, scales, dict(dWx=dWx, w=np.where(keep, w, np.inf), k=np.where(keep, k, -1),
                                               const=const, freqs_ascending=freqs)
    return Tx, Wx, out_freqs, scales


def issq_cwt(Tx, wavelet="gmw"):
    """_ssq_cwt.py:313-378 (full inverse): (2 / Css) sum over rows of Re Tx."""
    return Tx.real.sum(axis=0) * (2 / adm_ssq(wavelet))


def icwt(Wx, wavelet="gmw", scales=None, l1_norm=True, x_mean=0.0):
    """_cwt.py:321-493, one-integral, exponential scales: (2 / Cpsi) (ln 2 / nv) sum_a Re Wx [/ sqrt(a) if L2]."""
    scales = np.asarray(scales, dtype=np.float64).reshape(-1)
    nv = infer_nv(scales)
    norm = 1.0 if l1_norm else np.sqrt(scales)[:, None]                      # :483-492 (scaletype 'log')
    x = (Wx.real / norm).sum(axis=0)
    Cpsi = adm_ssq(wavelet)                                                  # :430-431 (one_int)
    x = x * (2 / Cpsi) * np.log(2 ** (1 / nv))                               # :432-433
    return x + x_mean
