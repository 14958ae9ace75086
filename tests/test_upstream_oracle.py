"""The upstream-variant restatement (oracle/upstream_oracle.py) against the facts upstream's OWN tests state
(/root/reference/old/tests/reconstruction_test.py) -- the only pins that exist: upstream does not import here
(numba missing), so parity of SURVEY 8(f)-4 is otherwise unpinned."""
import numpy as np
import pytest

from oracle import upstream_oracle as u


def _t(a, b, n):
    return np.linspace(a, b, n, endpoint=False)


def echirp(N):                                   # reconstruction_test.py:33-35
    t = _t(0, 10, N)
    return np.cos(2 * np.pi * 3 * np.exp(t / 3)), t


def mad_rms(x, xrec):                            # reconstruction_test.py:26-29
    return np.mean(np.abs(x - xrec)) / np.sqrt(np.mean(x ** 2))


def _dpss(n):
    from scipy.signal.windows import dpss        # upstream's default window (_stft.py:283-285)
    return dpss(n, max(4, n // 8), sym=False)


def test_stft_istft_reconstructs_to_1e14():
    """reconstruction_test.py:160-180: every even/odd combination, MAE < 1e-14."""
    rng = np.random.default_rng(0)
    for N in (128, 129):
        x = rng.standard_normal(N)
        for n_fft in (120, 121):
            win = _dpss(n_fft)
            for hop in (1, 2, 3):
                for mod in (True, False):
                    Sx = u.stft(x, win, n_fft=n_fft, hop_len=hop, modulated=mod)
                    assert Sx.shape == (n_fft // 2 + 1, (N - 1) // hop + 1)
                    xr = u.istft(Sx, win, n_fft=n_fft, hop_len=hop, N=N, modulated=mod)
                    assert len(xr) == N
                    assert np.abs(x - xr).mean() < 1e-14, (N, n_fft, hop, mod)


def test_ssq_stft_issq_stft_reconstructs_to_1e1():
    """reconstruction_test.py:183-206: MAE < 1e-1 (window scaling 1 and .5)."""
    rng = np.random.default_rng(1)
    for N in (128, 129):
        x = rng.standard_normal(N)
        for n_fft in (120, 121):
            for scaling in (1.0, 0.5):
                win = _dpss(n_fft) * scaling
                Tx, Sx, f, Sfs = u.ssq_stft(x, win, n_fft=n_fft)
                assert Tx.shape == Sx.shape == (n_fft // 2 + 1, N)
                xr = u.issq_stft(Tx, win, n_fft=n_fft)
                assert np.abs(x - xr).mean() < 1e-1, (N, n_fft, scaling)


def test_modulated_stft_is_the_plain_one_times_a_phase_ramp():
    """What the HIP path relies on: the modulated frame is the frame rotated by n_fft//2 (stft_utils.py:70-83)."""
    rng = np.random.default_rng(2)
    x = rng.standard_normal(300)
    for n_fft in (64, 61):
        win = np.hanning(n_fft)
        a = u.stft(x, win, n_fft=n_fft, hop_len=4, modulated=True)
        b = u.stft(x, win, n_fft=n_fft, hop_len=4, modulated=False)
        k = np.arange(n_fft // 2 + 1)[:, None]
        assert np.abs(a - b * np.exp(2j * np.pi * k * (n_fft // 2) / n_fft)).max() < 1e-12 * np.abs(b).max()


@pytest.mark.parametrize("wavelet", ["gmw", ("morlet", {"mu": 13.4})])
def test_cwt_icwt_issq_cwt_reconstruct_echirp(wavelet):
    """reconstruction_test.py:111-123 (there with the automatic 'log-piecewise' scales; here with an explicit
    exponential grid over the same range, the supported subset): mad_rms < 0.02 for icwt and issq_cwt."""
    x, ts = echirp(1024)
    fs = 1 / (ts[1] - ts[0])
    nv = 32
    wc = 20 ** (1 / 3) if wavelet == "gmw" else 13.4          # peak of psih: the finest scale puts it at Nyquist
    j0 = int(np.ceil(np.log2(wc / np.pi) * nv))
    scales = 2 ** (np.arange(j0, j0 + 9 * nv) / nv)
    Tx, Wx, f, sc = u.ssq_cwt(x, wavelet, scales=scales, fs=fs)
    assert Tx.shape == Wx.shape == (len(scales), 1024)
    assert mad_rms(x, u.icwt(Wx, wavelet, scales=scales)) < .02
    assert mad_rms(x, u.issq_cwt(Tx, wavelet)) < .02


def test_admissibility_constants():
    """Closed forms to check the quadrature against: for the L1 GMW  int psih/w dw = 2 e^{wc^g} wc^{-b} Gamma(b/g)/g."""
    from math import gamma as G
    g, b = 3.0, 60.0
    wc = (b / g) ** (1 / g)
    exact = 2 * np.exp(wc ** g - b * np.log(wc)) * G(b / g) / g
    assert abs(u.adm_ssq("gmw") - exact) < 1e-6 * exact
    assert u.p2up(1000) == (2048, 524, 524) and u.p2up(1024) == (2048, 512, 512) and u.p2up(1500)[0] == 4096


def test_admissibility_stays_away_from_zero_like_upstream_requires():
    """old/tests/adm_coef_test.py:16-38: adm_cwt and adm_ssq of the Morlet wavelet exceed 1e-3 for every mu in
    linspace(4, 30, 200) (an unstable quadrature would give ~0) -- on the restatement AND on the library's host code
    (`ssq_upstream_adm` needs no GPU)."""
    import ctypes as C
    from ssqueeze_rs_amd import _lib
    lib = _lib.load()
    out = C.c_double(0)
    for mu in np.linspace(4, 30, 200)[::10]:
        w = ("morlet", {"mu": float(mu)})
        a, b = u.adm_cwt(w), u.adm_ssq(w)
        assert a > 1e-3 and b > 1e-3
        for which, ref in ((1, a), (0, b)):
            assert lib.ssq_upstream_adm(_lib.WAVELET["morlet"], float(mu), 0.0, which, C.byref(out)) == 0
            assert abs(out.value - ref) <= 1e-10 * ref


def test_gmw_bandpass_time_domain_l1_norm_is_two():
    """old/tests/gmw_test.py:59-81 (norm='bandpass'): the time-domain wavelet built as `compute_gmw(..., time=True)` does
    (_gmw.py:134-184: positive-frequency half, Nyquist halved, ifft of psih * (-1)^n) has L1 norm 2 to 1e-3, for
    (gamma, beta) in {(3, 60), (4, 80)}, scales 2 and 3, N = 512 and 513; and the frequency-domain peak is 2."""
    for g, b in ((3, 60), (4, 80)):
        for scale in (2, 3):
            for N in (512, 513):
                w = u.xifn(scale, N)
                X = np.zeros(N)
                X[:N // 2 + 1] = u.gmw_l1(w[:N // 2 + 1], g, b)
                X[np.isinf(X) | np.isnan(X)] = 0.0
                Xr = X.copy()
                if N % 2 == 0:
                    Xr[N // 2] /= 2
                psi = np.fft.ifft(Xr * (-1) ** np.arange(N))
                assert abs(np.sum(np.abs(psi)) - 2) < 1e-3, (g, b, scale, N)
        wc = (b / g) ** (1 / g)
        ww = np.linspace(0.5 * wc, 1.5 * wc, 20001)
        assert abs(u.gmw_l1(ww, g, b).max() - 2) < 1e-6
        assert abs(u.center_frequency_peak(lambda x: u.gmw_l1(x, g, b), 1.0, 1 << 16) - wc) <= 2 * np.pi / (1 << 16)


def test_modulated_buffer_is_the_ifftshifted_plain_one_exactly():
    """old/tests/fft_test.py:383-415 (test_buffer): `buffer(..., modulated=True)` equals `ifftshift(buffer(...,
    modulated=False), axes=0)` with mean absolute difference exactly 0, for even and odd segment lengths and overlaps."""
    rng = np.random.default_rng(4)
    N = 128
    x = rng.standard_normal(N)
    for seg_len in (N // 2, N // 2 - 1):
        for n_overlap in (N // 2 - 1, N // 2 - 2, N // 2 - 3):
            if seg_len == n_overlap:
                continue
            a = u.buffer(x, seg_len, n_overlap, True)
            b = np.fft.ifftshift(u.buffer(x, seg_len, n_overlap, False), axes=0)
            assert np.abs(a - b).mean() == 0, (seg_len, n_overlap)
