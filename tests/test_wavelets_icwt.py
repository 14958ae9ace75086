"""SURVEY §8 (f-2) icwt and (f-3) the wavelet helper functions against the oracle restatements
(oracle/ssq_oracle.py: cwt.rs:550-718, wavelets/morlet.rs:59-145, wavelets/gmw.rs:226-357).
The helpers are host-side fp64 code in the library (no GPU needed); icwt computes on the GPU."""
import numpy as np
import pytest

from oracle import ssq_oracle as o
from ssqueeze_rs_amd import _rs


def _close(a, b, tol=1e-13):
    s = max(np.abs(b).max(), 1e-300)
    return np.abs(a - b).max() <= tol * s


# ------------------------------------------------------------------------------ wavelet helpers (CPU) ----
def test_morlet_family_matches_oracle():
    w = np.linspace(-3.0, 25.0, 997)
    for mu in (6.0, 5.0, 13.4):
        a, b = _rs.morlet(w, mu=mu), o.morlet(w, mu)
        assert a.dtype == np.complex128 and a.shape == w.shape and _close(a, b)
    for n, scale in ((1024, 1.0), (1000, 3.5), (7, 0.5), (1, 1.0)):
        assert _close(_rs.morlet_freq(n, scale), o.morlet_freq(n, scale))
        assert _close(_rs.morlet_time(n, scale), o.morlet_time(n, scale), 1e-12)
    assert _rs.morlet_freq(0).shape == (0,)
    assert _rs.morlet_freq().shape == (1024,)
    # the helper's normalisation differs from the hot path's inline Morlet (cwt.rs:497-520): SURVEY a-9
    assert abs(_rs.morlet(np.array([6.0]))[0].real - o.wavelet_fourier(np.array([6.0]), 1.0, "morlet")[0]) > 0.1


def test_gmw_family_matches_oracle():
    w = np.concatenate([[-1.0, 0.0], np.linspace(1e-3, 6.0, 500)])
    for kw in (dict(), dict(gamma=2.0, beta=10.0), dict(norm="energy"), dict(norm="BandPass"), dict(order=1),
               dict(order=3, gamma=3.0, beta=20.0), dict(order=2, norm="energy", beta=8.0)):
        a, b = _rs.gmw(w, **kw), o.gmw(w, **kw)
        assert a.dtype == np.complex128 and _close(a, b, 1e-12), kw
        assert a[0] == 0 and a[1] == 0                              # w <= 0 -> 0 (gmw.rs:84)
    assert abs(_rs.gmw(np.array([o.gmw_center_frequency()]))[0].real - 2.0) < 1e-12     # bandpass: peak value 2
    for n, scale in ((1024, 1.0), (777, 20.0), (16, 2.0)):
        assert _close(_rs.gmw_freq(n, scale), o.gmw_freq(n, scale), 1e-12)
        assert _close(_rs.gmw_time(n, scale, order=1), o.gmw_time(n, scale, order=1), 1e-11)
    for bad in (dict(gamma=0.0), dict(beta=-1.0), dict(order=-1)):
        with pytest.raises(ValueError):
            _rs.gmw(w, **bad)
        with pytest.raises(ValueError):
            o.gmw(w, **bad)
    assert _rs.gmw_center_frequency() == o.gmw_center_frequency() == (60.0 / 3.0) ** (1.0 / 3.0)
    assert abs(_rs.gmw_center_frequency(2.0, 7.0, "energy") - o.gmw_center_frequency(2.0, 7.0, "energy")) < 1e-13
    with pytest.raises(ValueError):
        _rs.gmw_center_frequency(kind="median")
    assert abs(o.gamma_function(5.0) - 24.0) < 1e-12 and abs(o.gamma_function(0.3) - 2.99156898768759) < 1e-10


def test_oracle_icwt_reconstructs():
    x = o.synth_signal(2048, 0)
    sc = o.log_scales(2048, 8)
    W, _, _ = o.cwt(x, "morlet", scales=sc)
    assert np.corrcoef(o.icwt(W, "morlet", scales=sc), x)[0, 1] > 0.98
    with pytest.raises(ValueError):
        o.icwt(W, "morlet")
    with pytest.raises(ValueError):
        _rs.icwt(W, "morlet")
    with pytest.raises(_rs.PanicException):
        _rs.icwt(W, "morlet", scales=sc, x_len=4096)


# -------------------------------------------------------------------------------------- icwt (GPU) ----
@pytest.mark.gpu
@pytest.mark.parametrize("N,nv", [(2048, 8), (3000, 4), (1 << 16, 6)])
def test_icwt_matches_oracle(N, nv):
    x = o.synth_signal(N, 90)
    sc = o.log_scales(N, nv)
    for wavelet in ("morlet", "gmw"):
        W, _, _ = o.cwt(x, wavelet, scales=sc)
        for kw in (dict(), dict(l1_norm=False), dict(x_mean=0.25), dict(x_len=N - 37)):
            a = _rs.icwt(W, wavelet, scales=sc, **kw)
            b = o.icwt(W, wavelet, scales=sc, **kw)
            assert a.dtype == np.float64 and a.shape == b.shape
            assert np.array_equal(a, b) or _close(a, b, 1e-15), (wavelet, kw)     # same summation order
        if N <= 3000 or wavelet == "morlet":
            for kw in (dict(), dict(l1_norm=False, x_mean=-1.0), dict(x_len=N - 37)):   # any length: Bluestein
                a = _rs.icwt(W, wavelet, scales=sc, one_int=False, **kw)
                b = o.icwt(W, wavelet, scales=sc, one_int=False, **kw)
                assert _close(a - kw.get("x_mean", 0.0), b - kw.get("x_mean", 0.0), 1e-11), (wavelet, kw)
    # complex64 input (extension) and a non-contiguous view
    W32 = W.astype(np.complex64)
    assert _close(_rs.icwt(W32, "gmw", scales=sc), o.icwt(W32.astype(np.complex128), "gmw", scales=sc), 1e-12)
    assert _close(_rs.icwt(W[:, ::2], "gmw", scales=sc), o.icwt(W[:, ::2], "gmw", scales=sc), 1e-15)
    # GPU round trip: cwt -> icwt recovers the signal shape
    Wg, scg, _ = _rs.cwt(x, wavelet="morlet", scales=sc)
    assert np.corrcoef(_rs.icwt(Wg, "morlet", scales=scg), x)[0, 1] > 0.97
