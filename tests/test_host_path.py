"""The drop-in call's host path: cached plans keyed by configuration, cached device scratch,
two-stream pipeline, results in the library's pinned pool.  Nothing here may change a single bit of the results."""
import gc

import numpy as np
import pytest

from oracle import ssq_oracle as o
from ssqueeze_rs_amd import _lib, _rs

pytestmark = pytest.mark.gpu


def _stats():
    import ctypes as C
    a, b = C.c_int64(0), C.c_int64(0)
    _lib.check(_lib.load().ssq_host_cache_stats(C.byref(a), C.byref(b)))
    return a.value, b.value


def test_pinned_results_are_ordinary_arrays_and_return_to_the_pool():
    gc.collect()
    live0, idle0 = _stats()
    x = o.synth_signal(50000, 1, np.float32)
    Tx, f = _rs.ssq_stft(x, np.hanning(512), n_fft=512, hop_len=128)
    live1, _ = _stats()
    assert live1 > live0
    assert Tx.flags.writeable and Tx.flags.c_contiguous and Tx.dtype == np.complex64
    view = Tx[5:9, ::2]
    ref = Tx.copy()
    del Tx
    gc.collect()
    assert np.array_equal(view, ref[5:9, ::2])            # a view keeps the block alive
    assert _stats()[0] == live1
    view[:] = 0                                           # and it is writable memory
    del view
    gc.collect()
    live2, idle2 = _stats()
    assert live2 == live0 and idle2 >= idle0              # back in the pool


def test_plan_cache_is_keyed_by_every_parameter_and_by_the_window_contents():
    x = o.synth_signal(30000, 2)
    w1, w2 = np.hanning(256), np.hamming(256)
    a1, _ = _rs.ssq_stft(x, w1, n_fft=256, hop_len=64, fs=2.0)
    b1, _ = _rs.ssq_stft(x, w2, n_fft=256, hop_len=64, fs=2.0)            # same shape, other window
    c1, _ = _rs.ssq_stft(x, w1, n_fft=256, hop_len=64, fs=2.0, squeezing="lebesgue")
    d1, _ = _rs.ssq_stft(x, w1, n_fft=256, hop_len=64, fs=3.0)
    e1, _ = _rs.ssq_stft(x, w1, n_fft=256, hop_len=64, fs=2.0, padtype="zero")
    g1, _ = _rs.ssq_stft(x, w1, n_fft=256, hop_len=64, fs=2.0, gamma=1e-2)
    a2, _ = _rs.ssq_stft(x, w1, n_fft=256, hop_len=64, fs=2.0)            # served by the cached plan
    assert np.array_equal(a1, a2)
    for other in (b1, c1, d1, e1, g1):
        assert not np.array_equal(a1, other)
    _lib.check(_lib.load().ssq_host_cache_clear())
    a3, _ = _rs.ssq_stft(x, w1, n_fft=256, hop_len=64, fs=2.0)
    assert np.array_equal(a1, a3)
    # more plans than the cache holds, then the first again
    for n in (64, 128, 512, 1024, 2048, 100, 200, 300, 400):
        _rs.stft(x, n, n // 4, np.hanning(n), "reflect")
    assert np.array_equal(a1, _rs.ssq_stft(x, w1, n_fft=256, hop_len=64, fs=2.0)[0])


def test_pipelined_batch_equals_single_calls_every_group_size():
    for N, B, n_fft, hop in ((3000, 37, 128, 32), (200000, 5, 1024, 256), (1 << 20, 3, 1024, 256)):
        xb = np.stack([o.synth_signal(N, 10 + b, np.float32) for b in range(B)])
        win = np.hanning(n_fft)
        Tb, _ = _rs.ssq_stft(xb, win, n_fft=n_fft, hop_len=hop)
        Sb, _ = _rs.stft(xb, n_fft, hop, win, "reflect")
        for b in (0, B // 2, B - 1):
            assert np.array_equal(Tb[b], _rs.ssq_stft(xb[b], win, n_fft=n_fft, hop_len=hop)[0])
            assert np.array_equal(Sb[b], _rs.stft(xb[b], n_fft, hop, win, "reflect")[0])


def test_cwt_host_path_cache_and_overlapped_download():
    xb = np.stack([o.synth_signal(6000, 20 + b) for b in range(5)])
    T1, f1 = _rs.ssq_cwt(xb, wavelet="morlet", nv=4)
    T2, f2 = _rs.ssq_cwt(xb, wavelet="morlet", nv=4)                      # cached plan + scratch
    assert np.array_equal(T1, T2) and np.array_equal(f1, f2)
    for b in range(5):
        assert np.array_equal(T1[b], _rs.ssq_cwt(xb[b], wavelet="morlet", nv=4)[0])
    W, sc, dW = _rs.cwt(xb, wavelet="gmw", nv=4, derivative=True)
    for b in (0, 4):
        Wb, _, dWb = _rs.cwt(xb[b], wavelet="gmw", nv=4, derivative=True)
        assert np.array_equal(W[b], Wb) and np.array_equal(dW[b], dWb)
    T3, _ = _rs.ssq_cwt(xb, wavelet="morlet", scales=sc[::2].copy())     # other scales: another plan
    assert T3.shape == (5, sc[::2].shape[0], 6000)
