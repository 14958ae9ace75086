"""CPU tests of the oracle itself: against the committed golden vectors, against independent
arithmetic (direct DFT, brute-force scan), against the facts the reference's scripts state, and the
C restatement against the NumPy one.  (The golden vectors were produced by this oracle -- "parity
unpinned", see oracle/ssq_oracle.py -- so they guard against drift, not against the reference.)"""
import os

import numpy as np
import pytest

from oracle import ssq_oracle as o

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def test_fft_restated_with_numpy_matches_direct_dft():
    rng = np.random.default_rng(0)
    for n in (8, 100, 256):
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        k = np.arange(n)
        W = np.exp(-2j * np.pi * np.outer(k, k) / n)
        assert np.abs(np.fft.fft(x) - W @ x).max() <= 1e-12 * np.abs(x).sum()
        assert np.abs(np.fft.ifft(x, norm="forward") - W.conj() @ x).max() <= 1e-12 * np.abs(x).sum()


def test_reference_stated_facts_c1():
    """README.md:66-79 / tests/stft_test.py:137-151 inputs; SURVEY.md §8: (129,16), peak bin 26, ssq row 25."""
    g = _load("c1_stft_ssq_stft.npz")
    Sx, freqs = o.stft(g["x"], 256, 64, g["window"], "reflect")
    assert Sx.shape == (129, 16) and Sx.dtype == np.complex128
    assert np.array_equal(freqs, g["freqs"]) and freqs[-1] == 0.5
    assert np.abs(Sx).sum(1).argmax() == 26
    assert np.array_equal(Sx, g["Sx"])
    Tx, sf, im = o.ssq_stft(g["x"], g["window"], n_fft=256, hop_len=64, fs=1000, return_intermediates=True)
    assert Tx.shape == (129, 16) and np.abs(Tx).sum(1).argmax() == 25
    assert np.array_equal(Tx, g["Tx"]) and np.array_equal(sf, g["ssq_freqs"])
    assert np.array_equal(im["k"], g["k"])


def test_padding_quirks():
    x = np.arange(1.0, 11.0)
    p = o.stft_pad(x, 8, "reflect")              # left = 3, right = 4 (the larger half goes right)
    assert p.shape[0] == 17
    assert np.array_equal(p[:3], [4.0, 3.0, 2.0]) and np.array_equal(p[13:], [9.0, 8.0, 7.0, 6.0])
    assert np.array_equal(o.stft_pad(x, 8, "zero")[:3], [0, 0, 0])
    assert np.array_equal(o.stft_pad(x, 8, "anything"), p)          # unknown -> reflect
    short = o.stft_pad(np.array([1.0, 2.0, 3.0]), 12, "reflect")    # mirror beyond the signal leaves zeros
    assert short.shape[0] == 14 and short[0] == 0.0 and short[-1] == 0.0
    assert o.next_power_of_2(1000 + 500) == 2048 and o.next_power_of_2(1 << 20) == 1 << 20
    assert o.next_power_of_2((1 << 20) + (1 << 19)) == 1 << 21
    q = o.cwt_pad(x, 16, "reflect")
    assert np.array_equal(q[:3], [4.0, 3.0, 2.0]) and np.array_equal(q[13:], [9.0, 8.0, 7.0])


def test_nearest_bin_matches_bruteforce_including_ties_and_extremes():
    f = o.stft_ssq_freqs(33, 2.0)
    dw = f[1] - f[0]
    rng = np.random.default_rng(1)
    w = np.concatenate([rng.uniform(0, 1.2, 500), f[:-1] + 0.5 * dw, f, [0.0, 5.0, 1e17, 1e300, np.nan]])
    w = w.reshape(-1, 1)
    assert np.array_equal(o.nearest_bin_first_min(w, f), o.nearest_bin_bruteforce(w, f))
    # exact half-bin tie -> lower bin; NaN -> 0; far beyond the last bin -> first of the tied minima
    assert o.nearest_bin_first_min(np.array([[f[3] + 0.5 * dw]]), f)[0, 0] in (3, 4)
    assert o.nearest_bin_first_min(np.array([[np.nan]]), f)[0, 0] == 0
    assert o.nearest_bin_first_min(np.array([[1e300]]), f)[0, 0] == 0


def test_rust_round_and_cwt_bins():
    v = np.array([0.5, -0.5, 1.5, 2.5, -2.5, 0.49999999999999994, 1e30, -1e30, np.nan])
    assert np.array_equal(o.rust_round(v)[:6], [1.0, -1.0, 2.0, 3.0, -3.0, 0.0])
    f_log = o.cwt_ssq_freqs(8, 0.01, 10.0, "log")
    b, valid, is_log = o.cwt_bins(np.array([0.0, 0.01, 10.0, 11.0, 1e9, np.inf, np.nan]), f_log)
    assert is_log and list(valid) == [False, True, True, True, False, False, False]
    f_lin = o.cwt_ssq_freqs(256, 2.0 ** -19, 0.5, "log")
    assert not o.cwt_bins(np.array([0.1]), f_lin)[2]            # 256 log scales: ratio 1.05 < 1.1 -> linear quirk


def test_golden_stft_modes_and_cwt():
    g = _load("stft4096_modes.npz")
    for sq in ("sum", "lebesgue"):
        for pad in ("reflect", "zero"):
            Tx, f, im = o.ssq_stft(g["x"], g["window"], n_fft=256, hop_len=64, fs=2.0, padtype=pad,
                                   squeezing=sq, return_intermediates=True)
            assert np.array_equal(Tx, g[f"Tx_{sq}_{pad}"])
            assert np.array_equal(np.where(np.isinf(im["w"]), -1, im["k"]), g[f"k_{sq}_{pad}"])
    g = _load("smoke_cwt_ssq_cwt.npz")
    for wv in ("gmw", "morlet"):
        Wx, sc, dWx = o.cwt(g["x"], wv, scales=g["scales"], fs=1000, nv=16, derivative=True)
        assert Wx.shape == (32, 1000)                           # tests/cwt_test.py:49-60
        assert np.array_equal(Wx[:, ::4], g[f"Wx_{wv}"]) and np.array_equal(dWx[:, ::4], g[f"dWx_{wv}"])
        T, f = o.ssq_cwt(g["x"], wv, scales=g["scales"], fs=1000, nv=16)
        assert T.shape == (32, 1000)                            # tests/ssq_cwt_test.py:49-57
        assert np.array_equal(T[:, ::4], g[f"Tx_{wv}"]) and np.array_equal(f, g[f"ssq_freqs_{wv}"])
    g = _load("cwt2048_options.npz")
    for name, kw in (("default", {}), ("noflip", dict(flipud=False)), ("maximal", dict(maprange="maximal")),
                     ("linear", dict(ssq_freqs="linear")), ("lebesgue", dict(squeezing="lebesgue"))):
        T, f, im = o.ssq_cwt(g["x"], "morlet", nv=4, return_intermediates=True, **kw)
        assert np.array_equal(T[:, ::16], g[f"Tx_{name}"]) and np.array_equal(f, g[f"f_{name}"])
        assert bool(g[f"is_log_{name}"]) == im["is_log"]
    assert np.array_equal(im["scales"], g["scales"])


def test_wavelet_quirks():
    xi = o.xifn(1.0, 64)
    assert xi[32] > 0 and xi[33] < 0 and abs(xi[32] - np.pi) < 1e-15      # Nyquist kept positive
    m = o.wavelet_fourier(xi, 4.0, "morlet")
    assert m[0] == 0.0 and (m[33:] == 0).all() and m.max() > 1.0
    assert abs(m.max() / (np.pi ** -0.25 * np.sqrt(2.0)) - 1) < 0.05        # pi^-1/4 * sqrt2 normalisation
    gm = o.wavelet_fourier(xi, 4.0, "anything")                            # unknown -> gmw, un-normalised
    assert gm.max() > 1e15 and gm[0] == 0.0
    s = o.log_scales(1000, 16)
    assert s.shape[0] == int(np.ceil((np.log2(500) - 1) * 16)) and s[0] == 2.0 and abs(s[-1] - 500) < 1e-9
    s2 = o.log_scales(1000, 16, simd_variant=True)
    assert np.allclose(s, s2, rtol=1e-14)


def test_c_restatement_matches_numpy_oracle():
    from oracle import ref_c
    x = o.synth_signal(5000, 7)
    for n_fft, hop, sq, pad in ((256, 64, "sum", "reflect"), (128, 50, "lebesgue", "zero"), (100, 30, "sum", "reflect")):
        win = o.size_window(np.hanning(n_fft - 10), n_fft)
        Tx_o, f_o, im = o.ssq_stft(x, win, n_fft=n_fft, hop_len=hop, fs=3.0, padtype=pad, squeezing=sq,
                                   return_intermediates=True)
        for mode in (0, 1):
            Tx, f, k = ref_c.ssq_stft(x, win, n_fft, hop, fs=3.0, padtype=pad, squeezing=sq, mode=mode, want_k=True)
            keep = ~np.isinf(im["w"])
            assert np.array_equal(f, f_o)
            assert np.array_equal(k[keep], im["k"][keep]) and (k[~keep] == -1).all()
            assert np.abs(Tx - Tx_o).max() <= 1e-12 * np.abs(Tx_o).max()


def test_c2_summary_statistics_match_oracle_run():
    """Config 2 at full size is too big to commit; its checksums are."""
    g = _load("c2_summary.npz")
    assert g["col_sums"].shape == (4096,) and g["row_energy"].shape == (513,) and g["k_hist"].sum() > 2_000_000
    assert float(g["sx_absmax"]) > 100


def test_c_restatement_of_ssq_cwt_matches_numpy_oracle():
    """oracle/ssq_ref.c::ssq_ref_ssq_cwt (bench.py's cpu_baseline of the C4 leg) against oracle.ssq_cwt: same
    frequencies, same bins (identical Tx up to summation rounding), both wavelets, the linear-formula quirk included."""
    from oracle import ref_c
    x = o.synth_signal(3000, 1, np.float64)
    for wavelet, sc in (("morlet", 2.0 ** np.linspace(1, 9, 40)), ("gmw", np.logspace(1, 3, 12) / 10)):
        T, f = ref_c.ssq_cwt(x, sc, wavelet=wavelet)
        To, fo = o.ssq_cwt(x, wavelet=wavelet, scales=sc)
        assert np.array_equal(f, fo)
        assert np.abs(T - To).max() <= 1e-12 * np.abs(To).max()
