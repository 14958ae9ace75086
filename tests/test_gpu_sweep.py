"""Seeded random sweeps over the argument space of the path (GPU vs the NumPy oracle).

The parametrised tests in test_gpu_stft.py / test_gpu_cwt.py pin named configurations; these sweeps walk the
corners between them: ragged signal lengths (shorter than one frame, not a multiple of the hop or of a tile),
hops from 1 to beyond n_fft, windows shorter than n_fft, both paddings and squeezings, odd sampling rates, batches,
both dtypes -- with the same tolerances as the named tests (they reuse their checkers).
The 16-wave fp32 n_fft = 1024 kernel (bench path) gets its own sweep: its interior / edge tile split depends on
N, hop and the tile size.
"""
import numpy as np
import pytest

from oracle import ssq_oracle as o
from ssqueeze_rs_amd import _rs
from tests.test_gpu_stft import _check_ssq_f32, _check_ssq_f64, _sig

pytestmark = pytest.mark.gpu


def _cases_f64(n):
    rng = np.random.default_rng(20250101)
    out = []
    for i in range(n):
        n_fft = int(rng.choice([64, 128, 256, 512, 1024, 2048, 4096, 96, 250, 1000])) if i % 3 else \
            int(rng.integers(8, 700))
        hop = int(rng.integers(1, max(2, n_fft // 2))) if i % 4 else int(rng.integers(n_fft, 2 * n_fft + 1))
        N = int(rng.integers(max(8, n_fft // 3), 6 * n_fft + 3000))
        if hop < 8 and N > 3000:
            N = int(rng.integers(200, 3000))          # keep the oracle's serial scan quick
        wl = n_fft if i % 2 else int(rng.integers(max(2, n_fft // 2), n_fft + 1))
        out.append((N, n_fft, hop, wl, ["reflect", "zero"][i % 2], ["sum", "lebesgue"][(i // 2) % 2],
                    float(rng.choice([1.0, 2.0, 1000.0, 24414.0625, 0.37]))))
    return out


@pytest.mark.parametrize("case", _cases_f64(24), ids=lambda c: "N%d-nfft%d-hop%d-w%d-%s-%s" % c[:6])
def test_ssq_stft_f64_sweep(case):
    N, n_fft, hop, wl, pad, sq, fs = case
    if pad == "reflect" and N < (n_fft + 1) // 2 + 1:
        pad = "zero"                                   # reflect needs N > pad_left (stft_utils.rs:19-49)
    x = _sig(N, N % 97)
    _check_ssq_f64(x, np.hanning(wl), n_fft, hop, fs, pad, sq)


def _cases_tx1024(n):
    rng = np.random.default_rng(777)
    hops = [1, 7, 37, 100, 255, 256, 257, 512, 1000, 1024, 1500, 2048, 3000]
    out = []
    for i in range(n):
        hop = hops[i % len(hops)]
        # few frames (all tiles are edge tiles), one tile, several tiles with a ragged tail
        frames = int(rng.choice([1, 2, 15, 16, 17, 31, 33, 50, 129]))
        N = max(600, (frames - 1) * hop + int(rng.integers(1, hop + 1)))
        if hop < 37:
            N = min(N, 3000)
        out.append((N, hop, ["reflect", "zero"][i % 2], ["sum", "lebesgue"][(i // 2) % 2]))
    return out


@pytest.mark.parametrize("case", _cases_tx1024(20), ids=lambda c: "N%d-hop%d-%s-%s" % c)
def test_ssq_stft_f32_n1024_kernel_sweep(case):
    N, hop, pad, sq = case
    x = _sig(N, N % 89, np.float32)
    _check_ssq_f32(x, np.hanning(1024), 1024, hop, 1.0, pad, sq)


def test_ssq_stft_f32_n1024_batch_rows_are_independent():
    """A batch through the plan equals its signals one by one, bitwise (fixed-point tile: order-exact)."""
    xb = np.stack([_sig(50000, 300 + b, np.float32) for b in range(5)])
    win = np.hanning(1024)
    Tb, f = _rs.ssq_stft(xb, win, n_fft=1024, hop_len=256)
    assert Tb.shape == (5, 513, (50000 - 1) // 256 + 1)
    for b in range(5):
        T1, f1 = _rs.ssq_stft(xb[b], win, n_fft=1024, hop_len=256)
        assert np.array_equal(Tb[b], T1) and np.array_equal(f, f1)


def _cases_cwt(n):
    rng = np.random.default_rng(4242)
    out = []
    for i in range(n):
        N = int(rng.choice([16, 33, 100, 777, 2730, 2731, 4097, 9000, 21845, 21846, 30000]))
        out.append((N, int(rng.choice([1, 2, 4, 7])), ["morlet", "gmw"][i % 2],
                    [np.float64, np.float32][(i // 2) % 2], ["reflect", "zero"][(i // 4) % 2],
                    [None, 1.0, 250.0][i % 3]))
    return out


@pytest.mark.parametrize("case", _cases_cwt(16),
                         ids=lambda c: "N%d-nv%d-%s-%s-%s" % (c[0], c[1], c[2], np.dtype(c[3]).name, c[4]))
def test_cwt_sweep(case):
    N, nv, wavelet, dtype, pad, fs = case
    x = _sig(N, N % 83, dtype)
    Wx, sc, dWx = _rs.cwt(x, wavelet=wavelet, nv=nv, fs=fs, padtype=pad, derivative=True, l1_norm=(N % 2 == 0))
    Wx_o, sc_o, dWx_o = o.cwt(x.astype(np.float64), wavelet, nv=nv, fs=fs, padtype=pad, derivative=True,
                              l1_norm=(N % 2 == 0))
    assert Wx.shape == Wx_o.shape == (len(sc_o), N) and np.array_equal(sc, sc_o)
    tol = 1e-11 if dtype == np.float64 else 2e-5
    assert np.abs(Wx - Wx_o).max() <= tol * np.abs(Wx_o).max()
    assert np.abs(dWx - dWx_o).max() <= tol * np.abs(dWx_o).max()


@pytest.mark.parametrize("case", _cases_cwt(8)[::2], ids=lambda c: "N%d-nv%d-%s" % (c[0], c[1], c[2]))
def test_ssq_cwt_sweep_f64(case):
    N, nv, wavelet, _, pad, fs = case
    x = _sig(N, N % 79)
    Tx, f, dbg = _rs.ssq_cwt(x, wavelet=wavelet, nv=nv, fs=fs, padtype=pad, _debug=True)
    Tx_o, f_o, im = o.ssq_cwt(x, wavelet, nv=nv, fs=fs, padtype=pad, return_intermediates=True)
    assert Tx.shape == Tx_o.shape and np.array_equal(f, f_o)
    wmax = np.abs(im["Wx"]).max()
    assert np.abs(dbg["Wx"] - im["Wx"]).max() <= 1e-11 * wmax
    keep = dbg["k"] >= 0
    both = keep & im["valid"]
    assert (keep == im["valid"]).mean() >= 0.995
    if both.any():                                     # tiny inputs: every bin may be out of range (dropped)
        assert (dbg["k"][both] == im["k"][both]).mean() >= 0.999
    # the scatter itself, from the kernel's own bins
    Tx_re = np.zeros_like(Tx_o)
    rows, cols = np.nonzero(keep)
    np.add.at(Tx_re, (dbg["k"][rows, cols].astype(np.int64), cols), im["Wx"][rows, cols])
    assert np.abs(Tx - Tx_re).max() <= 1e-9 * max(wmax, 1e-300)
