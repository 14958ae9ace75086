"""GPU parity of the upstream-parity mode and the inverses (`ssqueeze_rs_amd.upstream`).

Checked against the numba-free restatement of the vendored upstream (oracle/upstream_oracle.py) and -- the only facts
upstream itself pins -- its reconstruction thresholds (/root/reference/old/tests/reconstruction_test.py:111-123,
:160-206).  Parity against upstream proper is UNPINNED: it does not import here (numba missing).
Tolerances: fp64 |dSx|,|dWx| <= 1e-11 max, bins index-exact except within 1e-9 of a rounding boundary (ties are
half-to-even: algos.py:957-968), Tx <= 1e-10 after re-accumulating the oracle with the kernel's own bins."""
import numpy as np
import pytest

from oracle import upstream_oracle as u
from ssqueeze_rs_amd import upstream as up

pytestmark = pytest.mark.gpu


def _t(a, b, n):
    return np.linspace(a, b, n, endpoint=False)


def echirp(N):                                   # reconstruction_test.py:33-35
    t = _t(0, 10, N)
    return np.cos(2 * np.pi * 3 * np.exp(t / 3)), t


def mad_rms(x, xrec):                            # reconstruction_test.py:26-29
    return np.mean(np.abs(x - xrec)) / np.sqrt(np.mean(x ** 2))


def _dpss(n):
    from scipy.signal.windows import dpss        # upstream's default window (_stft.py:283-285); data, not code under test
    return dpss(n, max(4, n // 8), sym=False)


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("n_fft,hop,mod", [(120, 1, True), (121, 2, True), (120, 3, False), (256, 64, True),
                                           (1024, 256, True), (20, 5, True)])
def test_stft_matches_upstream_restatement(n_fft, hop, mod):
    rng = np.random.default_rng(n_fft + hop)
    x = rng.standard_normal(700 if n_fft < 1024 else 5000)
    win = _dpss(n_fft)
    Sx, dSx = up.stft(x, win, n_fft=n_fft, hop_len=hop, fs=3.0, modulated=mod, derivative=True)
    So, dSo = u.stft(x, win, n_fft=n_fft, hop_len=hop, fs=3.0, modulated=mod, derivative=True)
    assert Sx.shape == So.shape and Sx.dtype == np.complex128
    assert _rel(Sx, So) <= 1e-11 and _rel(dSx, dSo) <= 1e-11
    Sz = up.stft(x, win, n_fft=n_fft, hop_len=hop, padtype="zero", modulated=mod)
    assert _rel(Sz, u.stft(x, win, n_fft=n_fft, hop_len=hop, padtype="zero", modulated=mod)) <= 1e-11


def test_stft_istft_reconstruction_thresholds():
    """reconstruction_test.py:160-180: MAE < 1e-14 for every even/odd (N, n_fft, hop_len, modulated)."""
    rng = np.random.default_rng(0)
    for N in (128, 129):
        x = rng.standard_normal(N)
        for n_fft in (120, 121):
            win = _dpss(n_fft)
            for hop in (1, 2, 3):
                for mod in (True, False):
                    Sx = up.stft(x, win, n_fft=n_fft, hop_len=hop, modulated=mod)
                    xr = up.istft(Sx, win, n_fft=n_fft, hop_len=hop, N=N, modulated=mod)
                    assert len(xr) == N
                    assert np.abs(x - xr).mean() < 1e-14, (N, n_fft, hop, mod)
                    # and the inverse alone against the restatement, on the oracle's spectrum
                    So = u.stft(x, win, n_fft=n_fft, hop_len=hop, modulated=mod)
                    assert np.abs(up.istft(So, win, n_fft=n_fft, hop_len=hop, N=N, modulated=mod) -
                                  u.istft(So, win, n_fft=n_fft, hop_len=hop, N=N, modulated=mod)).max() < 1e-13


def _check_ssq_stft(x, win, **kw):
    out = up.ssq_stft(x, win, get_w=True, get_dWx=True, **kw)
    Tx, Sx, f, Sfs, w, dSx = out
    To, So, fo, Sfo, im = u.ssq_stft(x, win, return_intermediates=True,
                                     **{k: v for k, v in kw.items()})
    assert Tx.shape == To.shape and np.array_equal(f, fo) and np.array_equal(Sfs, Sfo)
    smax = np.abs(So).max()
    assert np.abs(Sx - So).max() <= 1e-11 * smax and np.abs(dSx - im["dSx"]).max() <= 1e-11 * np.abs(im["dSx"]).max()
    keep_o = im["k"] >= 0
    keep_g = np.isfinite(w)
    flip = keep_o != keep_g                                  # |Sx| at the gamma threshold
    assert np.abs(So[flip]).max(initial=0.0) <= 1e-6 * smax + 1e-12
    # the kernel's bins through the restated rule on the kernel's own w: index-exact away from rounding boundaries
    n = Tx.shape[0]
    dv = float(Sfo[1] - Sfo[0])
    k_own = np.minimum(np.rint(np.maximum((np.where(keep_g, w, 0.0) - Sfo[0]) / dv, 0)), n - 1).astype(np.int64)
    if kw.get("flipud"):
        k_own = n - 1 - k_own
    # re-accumulate the oracle's Sx with those bins: the scatter itself
    const = im["const"]
    Wv = (np.ones(So.shape) / n) if kw.get("squeezing") == "lebesgue" else So
    Tre = np.zeros_like(To)
    cols = np.arange(Tx.shape[1])
    for i in range(n):
        m = keep_g[i]
        np.add.at(Tre, (k_own[i, m], cols[m]), Wv[i, m] * const)
    assert np.abs(Tx - Tre).max() <= 1e-10 * max(np.abs(Tre).max(), 1e-300)
    # and end to end against the oracle's own bins: only at half-bin boundaries (w agrees to ~1e-12)
    both = keep_o & keep_g & (np.abs(So) > 1e-6 * smax)
    mism = both & (k_own != im["k"])
    if mism.any():
        v = im["w"][mism] / dv
        assert (np.abs(np.abs(v - np.floor(v)) - 0.5) < 1e-6).all()
    assert mism.mean() <= 1e-3
    return Tx


def test_ssq_stft_matches_upstream_restatement():
    rng = np.random.default_rng(5)
    x = np.cos(2 * np.pi * 0.11 * np.arange(900)) + 0.5 * np.cos(2 * np.pi * (0.2 + 1e-4 * np.arange(900)) * np.arange(900)) \
        + 0.01 * rng.standard_normal(900)
    _check_ssq_stft(x, _dpss(128), n_fft=128)
    _check_ssq_stft(x, _dpss(121), n_fft=121, hop_len=3, fs=50.0)
    _check_ssq_stft(x, np.hanning(256), n_fft=256, hop_len=4, flipud=True, squeezing="lebesgue")
    _check_ssq_stft(x, np.hanning(100), n_fft=128, hop_len=2, padtype="zero")       # window centre-padded (get_window)


def test_ssq_stft_issq_stft_reconstruction_thresholds():
    """reconstruction_test.py:183-206: MAE < 1e-1 with window scaling 1 and .5."""
    rng = np.random.default_rng(1)
    for N in (128, 129):
        x = rng.standard_normal(N)
        for n_fft in (120, 121):
            for scaling in (1.0, 0.5):
                win = _dpss(n_fft) * scaling
                Tx, *_ = up.ssq_stft(x, win, n_fft=n_fft)
                xr = up.issq_stft(Tx, win, n_fft=n_fft)
                assert len(xr) == N and np.abs(x - xr).mean() < 1e-1, (N, n_fft, scaling)
                To, *_ = u.ssq_stft(x, win, n_fft=n_fft)
                assert np.abs(up.issq_stft(To, win, n_fft=n_fft) - u.issq_stft(To, win, n_fft=n_fft)).max() < 1e-12


def _scales(wavelet, nv=32, octaves=9):
    wc = 20 ** (1 / 3) if wavelet == "gmw" else 13.4
    j0 = int(np.ceil(np.log2(wc / np.pi) * nv))
    return 2 ** (np.arange(j0, j0 + octaves * nv) / nv)


@pytest.mark.parametrize("wavelet", ["gmw", ("morlet", {"mu": 13.4}), ("gmw", {"gamma": 3, "beta": 20})])
def test_cwt_matches_upstream_restatement(wavelet):
    x, ts = echirp(1000)                      # p2up(1000) = 2048 with n1 = n2 = 524; 1001 -> odd split
    fs = 1 / (ts[1] - ts[0])
    sc = _scales(wavelet if isinstance(wavelet, str) else wavelet[0], nv=8)
    for xx in (x, np.append(x, 0.3)):
        Wx, s, dWx = up.cwt(xx, wavelet, scales=sc, fs=fs, derivative=True)
        Wo, so, dWo = u.cwt(xx, wavelet, scales=sc, fs=fs, derivative=True)
        assert Wx.shape == Wo.shape == (len(sc), len(xx))
        assert _rel(Wx, Wo) <= 1e-11 and _rel(dWx, dWo) <= 1e-11
    W2, _ = up.cwt(x, wavelet, scales=sc, fs=fs, l1_norm=False, rpadded=True, padtype="zero")
    Wo2, _ = u.cwt(x, wavelet, scales=sc, fs=fs, l1_norm=False, rpadded=True, padtype="zero")
    assert W2.shape == Wo2.shape == (len(sc), 2048) and _rel(W2, Wo2) <= 1e-11


@pytest.mark.parametrize("wavelet", ["gmw", ("morlet", {"mu": 13.4})])
@pytest.mark.parametrize("kw", [dict(), dict(flipud=False, squeezing="lebesgue"), dict(maprange="maximal"),
                                dict(ssq_freqs="linear")])
def test_ssq_cwt_matches_upstream_restatement(wavelet, kw):
    x, ts = echirp(1024)
    fs = 1 / (ts[1] - ts[0])
    sc = _scales(wavelet if isinstance(wavelet, str) else wavelet[0], nv=16)
    Tx, Wx, f, s, w, dWx = up.ssq_cwt(x, wavelet, scales=sc, fs=fs, get_w=True, get_dWx=True, **kw)
    To, Wo, fo, so, im = u.ssq_cwt(x, wavelet, scales=sc, fs=fs, return_intermediates=True, **kw)
    assert Tx.shape == To.shape and np.allclose(f, fo, rtol=1e-14, atol=0)
    wmax = np.abs(Wo).max()
    assert np.abs(Wx - Wo).max() <= 1e-11 * wmax and np.abs(dWx - im["dWx"]).max() <= 1e-11 * np.abs(im["dWx"]).max()
    keep_g, keep_o = np.isfinite(w), im["k"] >= 0
    assert np.abs(Wo[keep_g != keep_o]).max(initial=0.0) <= 1e-6 * wmax + 1e-12
    na = len(sc)
    fa = im["freqs_ascending"]
    with np.errstate(all="ignore"):
        if kw.get("ssq_freqs") == "linear":
            v = (w - fa[0]) / (fa[1] - fa[0])
        else:
            v = (np.log2(w) - np.log2(fa[0])) / (np.log2(fa[1]) - np.log2(fa[0]))
        k_own = np.minimum(np.rint(np.maximum(np.where(keep_g, v, 0.0), 0)), na - 1).astype(np.int64)
    if kw.get("flipud", True):
        k_own = na - 1 - k_own
    Wv = (np.ones(Wo.shape) / na) if kw.get("squeezing") == "lebesgue" else Wo
    Tre = np.zeros_like(To)
    cols = np.arange(Tx.shape[1])
    for i in range(na):
        m = keep_g[i]
        np.add.at(Tre, (k_own[i, m], cols[m]), Wv[i, m] * im["const"])
    assert np.abs(Tx - Tre).max() <= 1e-10 * max(np.abs(Tre).max(), 1e-300)
    both = keep_o & keep_g & (np.abs(Wo) > 1e-6 * wmax)
    mism = both & (k_own != im["k"])
    if mism.any():
        vv = v[mism]
        assert (np.abs(np.abs(vv - np.floor(vv)) - 0.5) < 1e-6).all()
    assert mism.mean() <= 1e-3


@pytest.mark.parametrize("wavelet", ["gmw", ("morlet", {"mu": 13.4})])
def test_cwt_icwt_issq_cwt_reconstruction_thresholds(wavelet):
    """reconstruction_test.py:111-123: mad_rms < 0.02 on echirp(1024) for icwt and issq_cwt (explicit exponential scales
    over the range upstream's automatic ones cover)."""
    x, ts = echirp(1024)
    fs = 1 / (ts[1] - ts[0])
    sc = _scales(wavelet if isinstance(wavelet, str) else wavelet[0], nv=32)
    Tx, Wx, f, s = up.ssq_cwt(x, wavelet, scales=sc, fs=fs)
    assert mad_rms(x, up.issq_cwt(Tx, wavelet)) < .02
    assert mad_rms(x, up.icwt(Wx, wavelet, scales=sc)) < .02
    assert abs(up.adm_ssq(wavelet) - u.adm_ssq(wavelet)) <= 1e-12 * u.adm_ssq(wavelet)
    assert abs(up.adm_cwt(wavelet) - u.adm_cwt(wavelet)) <= 1e-12 * u.adm_cwt(wavelet)
    To, Wo, *_ = u.ssq_cwt(x, wavelet, scales=sc, fs=fs)
    assert np.abs(up.issq_cwt(To, wavelet) - u.issq_cwt(To, wavelet)).max() <= 1e-12 * np.abs(x).max()
    assert np.abs(up.icwt(Wo, wavelet, scales=sc) - u.icwt(Wo, wavelet, scales=sc)).max() <= 1e-12 * np.abs(x).max()
    assert np.abs(up.icwt(Wo, wavelet, scales=sc, l1_norm=False, x_mean=2.0) -
                  u.icwt(Wo, wavelet, scales=sc, l1_norm=False, x_mean=2.0)).max() <= 1e-12 * 3


def test_float32_mode_and_batches():
    """upstream's default dtype is float32: same pipeline in fp32 (10 eps32 threshold), and [batch, N] input."""
    rng = np.random.default_rng(3)
    xb = rng.standard_normal((3, 400)).astype(np.float32)
    win = np.hanning(64)
    Tx, Sx, f, Sfs = up.ssq_stft(xb, win, n_fft=64, hop_len=2)
    assert Tx.shape == (3, 33, 200) and Tx.dtype == np.complex64 and f.dtype == np.float32
    for b in range(3):
        To, So, *_ = u.ssq_stft(xb[b].astype(np.float64), win, n_fft=64, hop_len=2, gamma=10 * u.EPS32)
        assert np.abs(Sx[b] - So).max() <= 2e-6 * np.abs(So).max()
        assert np.abs(Tx[b].sum(0) - To.sum(0)).max() <= 1e-4 * np.abs(So).max() * (Sfs[1] - Sfs[0]) * 33
    sc = _scales("gmw", nv=8, octaves=6)
    Tc, Wc, fc, s = up.ssq_cwt(xb, "gmw", scales=sc)
    assert Tc.shape == (3, len(sc), 400) and Tc.dtype == np.complex64
    Wo, _ = u.cwt(xb[1].astype(np.float64), "gmw", scales=sc)
    assert np.abs(Wc[1] - Wo).max() <= 5e-6 * np.abs(Wo).max()


def test_unsupported_options_raise_value_error():
    x = np.zeros(64)
    with pytest.raises(ValueError):
        up.cwt(x, "gmw", scales="log-piecewise")
    with pytest.raises(ValueError):
        up.cwt(x, "bump", scales=_scales("gmw", 8, 3))
    with pytest.raises(ValueError):
        up.issq_stft(np.zeros((33, 64), dtype=np.complex128), np.hanning(64), hop_len=2)
    with pytest.raises(ValueError):
        up.stft(x, np.hanning(80), n_fft=64)                       # win_len > n_fft (_stft.py:264-266)


def test_rs_functions_take_the_upstream_switch():
    """`_rs.*(..., _upstream=True)`: the drop-in signatures and return arity with upstream's numerics."""
    from ssqueeze_rs_amd import _rs
    x, ts = echirp(1000)
    win = np.hanning(256)
    Sx, fr = _rs.stft(x, 256, 64, win, "reflect", _upstream=True)
    assert _rel(Sx, u.stft(x, win, n_fft=256, hop_len=64)) <= 1e-11 and fr.shape == (129,)
    Tx, f = _rs.ssq_stft(x, win, n_fft=256, hop_len=64, fs=100.0, _upstream=True)
    To, _, fo, _ = u.ssq_stft(x, win, n_fft=256, hop_len=64, fs=100.0)
    assert np.array_equal(f, fo) and np.abs(Tx.sum(0) - To.sum(0)).max() <= 1e-9 * np.abs(To).max() * 129
    Wx, sc, dWx = _rs.cwt(x, "morlet", nv=8, _upstream=True)
    Wo, _ = u.cwt(x, "morlet", scales=sc)
    assert dWx is None and _rel(Wx, Wo) <= 1e-11
    T2, f2 = _rs.ssq_cwt(x, "gmw", nv=8, _upstream=True)
    To2, _, fo2, _ = u.ssq_cwt(x, "gmw", scales=sc)
    assert np.allclose(f2, fo2, rtol=1e-14) and np.abs(T2.sum(0) - To2.sum(0)).max() <= 1e-9 * np.abs(To2).max() * len(sc)


def test_edge_shapes():
    """Tiny and ragged inputs: n_fft 2 / 3 (odd) / 8, hop > n_fft, hop = n_fft, N barely above n_fft, N = 1, an all-zero
    signal (every bin below gamma), N smaller than the padded wavelet support, two scales, a non-power-of-two N with an
    uneven p2up split -- all against the restatement."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal(50)
    win = np.hanning(16)
    for kw, w in ((dict(n_fft=16, hop_len=4), win), (dict(n_fft=2, hop_len=1), np.ones(2)),
                  (dict(n_fft=3, hop_len=2), np.hanning(5)[1:4]), (dict(n_fft=16, hop_len=20), win)):
        assert np.abs(up.stft(x, w, **kw) - u.stft(x, w, **kw)).max() <= 1e-13
    assert np.abs(up.stft(x[:17], win, n_fft=16, hop_len=4) - u.stft(x[:17], win, n_fft=16, hop_len=4)).max() <= 1e-13
    assert up.stft(x[:1], win, n_fft=16).shape == (9, 1)
    assert np.abs(up.ssq_stft(x, np.hanning(8), n_fft=8)[0] - u.ssq_stft(x, np.hanning(8), n_fft=8)[0]).max() <= 1e-13
    So = u.stft(x, win, n_fft=16, hop_len=8)
    assert np.abs(up.istft(So, win, n_fft=16, hop_len=8, N=50) - u.istft(So, win, n_fft=16, hop_len=8, N=50)).max() <= 1e-13
    assert np.abs(up.ssq_stft(np.zeros(64), win, n_fft=16)[0]).max() == 0.0
    sc = 2.0 ** (np.arange(8, 24) / 8)
    assert np.abs(up.cwt(x[:20], "gmw", scales=sc)[0] - u.cwt(x[:20], "gmw", scales=sc)[0]).max() <= 1e-13
    assert up.cwt(x[:3], "gmw", scales=sc)[0].shape == (16, 3)
    assert np.abs(up.ssq_cwt(np.zeros(64), "morlet", scales=sc)[0]).max() == 0.0
    xx = rng.standard_normal(1001)                       # p2up(1001) = 2048 with n1 = 524, n2 = 523
    a = up.ssq_cwt(xx, ("gmw", {"beta": 12}), scales=sc)[0]
    b = u.ssq_cwt(xx, ("gmw", {"beta": 12}), scales=sc)[0]
    assert np.abs(a.sum(0) - b.sum(0)).max() <= 1e-12 * np.abs(b).max() * len(sc)
    big = rng.standard_normal(20000)                     # n_fft beyond the fused range, not a power of two
    assert _rel(up.stft(big, np.hanning(5000), n_fft=5000, hop_len=1250), u.stft(big, np.hanning(5000), n_fft=5000, hop_len=1250)) <= 1e-11


def test_hop_one_on_a_long_signal_has_more_frames_than_a_grid_dimension():
    """upstream's default hop_len is 1: 70 000 frames > 65 535 (the y-limit of a launch grid) through the packing kernel,
    the inverse, and the n_fft rows of the istft expansion."""
    rng = np.random.default_rng(9)
    x = rng.standard_normal(70000)
    win = _dpss(32)
    Sx = up.stft(x, win, n_fft=32, hop_len=1)
    So = u.stft(x, win, n_fft=32, hop_len=1)
    assert Sx.shape == (17, 70000) and _rel(Sx, So) <= 1e-11
    xr = up.istft(Sx, win, n_fft=32, hop_len=1, N=70000)
    assert np.abs(x - xr).mean() < 1e-14
    Tx, *_ = up.ssq_stft(x, win, n_fft=32)
    To, *_ = u.ssq_stft(x, win, n_fft=32)
    assert np.abs(Tx.sum(0) - To.sum(0)).max() <= 1e-10 * np.abs(To).max() * 17      # invariant under bin flips
    assert np.abs(up.issq_stft(Tx, win, n_fft=32) - u.issq_stft(To, win, n_fft=32)).max() <= 1e-9
