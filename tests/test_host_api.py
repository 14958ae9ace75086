"""CPU tests of the Python mirror `ssqueeze_rs_amd._rs`: surface, defaults and error behaviour of
the reference's PyO3 module (rust/src/lib.rs:22-35, src/ssqueeze/_rs.pyi) -- everything that is
decided before the GPU is touched."""
import inspect

import numpy as np
import pytest

from ssqueeze_rs_amd import _rs


def test_module_surface_matches_lib_rs():
    for name in ("hello_from_bin", "stft", "ssq_stft", "cwt", "cwt_simd", "ssq_cwt"):   # lib.rs:25-32
        assert callable(getattr(_rs, name))
    assert _rs.hello_from_bin() == "Hello from ssqueeze!"
    import ssqueeze_rs_amd
    assert ssqueeze_rs_amd._rs is _rs


def _defaults(fn):
    return {k: v.default for k, v in inspect.signature(fn).parameters.items()
            if v.default is not inspect.Parameter.empty and not k.startswith("_")}


def test_signatures_and_defaults_match_pyo3_signatures():
    # (private switches such as `_debug` / `_upstream` start with an underscore and are not part of the mirrored surface)
    assert [k for k in inspect.signature(_rs.stft).parameters if not k.startswith("_")] == \
        ["x", "n_fft", "hop_length", "window", "padtype"]
    assert _defaults(_rs.stft) == {}                                             # stft.rs:12-19: all required
    assert _defaults(_rs.ssq_stft) == dict(n_fft=None, win_len=None, hop_len=1, fs=1.0, padtype="reflect",
                                           squeezing="sum", gamma=None)          # ssq_stft.rs:73
    cwt_d = dict(wavelet="gmw", scales=None, fs=None, t=None, nv=32, l1_norm=True, derivative=False,
                 padtype="reflect", rpadded=False, vectorized=True, patience=0)  # cwt.rs:32-45
    assert _defaults(_rs.cwt) == cwt_d and _defaults(_rs.cwt_simd) == cwt_d
    assert _defaults(_rs.ssq_cwt) == dict(wavelet="gmw", scales=None, fs=None, t=None, ssq_freqs=None, nv=32,
                                          padtype="reflect", squeezing="sum", maprange="peak", difftype="trig",
                                          gamma=None, vectorized=True, flipud=True)   # ssq_cwt.rs:245-260
    assert list(inspect.signature(_rs.ssq_stft).parameters)[:2] == ["x", "window"]
    # the functions the stubs advertise beyond lib.rs (src/ssqueeze/_rs.pyi:61-132; cwt.rs:551, morlet.rs:60,:81,:104,
    # gmw.rs:237,:266,:293,:341)
    assert _defaults(_rs.icwt) == dict(wavelet="gmw", scales=None, nv=None, one_int=True, x_len=None, x_mean=0.0,
                                       padtype="reflect", rpadded=False, l1_norm=True)
    assert _defaults(_rs.morlet) == dict(mu=6.0, dtype="float64")
    assert _defaults(_rs.morlet_freq) == _defaults(_rs.morlet_time) == dict(n=1024, scale=1.0, mu=6.0, dtype="float64")
    assert _defaults(_rs.gmw) == dict(gamma=3.0, beta=60.0, norm="bandpass", order=0, dtype="float64")
    assert _defaults(_rs.gmw_freq) == _defaults(_rs.gmw_time) == dict(n=1024, scale=1.0, gamma=3.0, beta=60.0,
                                                                      norm="bandpass", order=0, dtype="float64")
    assert _defaults(_rs.gmw_center_frequency) == dict(gamma=3.0, beta=60.0, kind="peak")


def test_argument_errors_raised_before_any_gpu_work():
    x = np.sin(np.arange(1000.0))
    win = np.hanning(256)
    with pytest.raises(TypeError):
        _rs.stft(list(x), 256, 64, win, "reflect")               # not an ndarray
    with pytest.raises(TypeError):
        _rs.stft(x.astype(np.int64), 256, 64, win, "reflect")    # not float
    with pytest.raises(TypeError):
        _rs.stft(x.reshape(10, 10, 10), 256, 64, win, "reflect")
    with pytest.raises(TypeError):
        _rs.stft(x, 256.0, 64, win, "reflect")                   # usize argument
    with pytest.raises(OverflowError):
        _rs.stft(x, -256, 64, win, "reflect")
    with pytest.raises(_rs.PanicException):
        _rs.stft(x, 256, 64, np.hanning(100), "reflect")         # stft.rs:67 rustfft length panic
    with pytest.raises(_rs.PanicException):
        _rs.stft(x, 256, 0, win, "reflect")                      # /0
    with pytest.raises(_rs.PanicException):
        _rs.stft(x[:0], 256, 64, win, "reflect")                 # usize underflow
    with pytest.raises(ValueError, match="Window length 300 cannot be greater than n_fft 256"):
        _rs.ssq_stft(x, np.hanning(300), n_fft=256)              # ssq_stft.rs:96-101
    with pytest.raises(ValueError):
        _rs.ssq_stft(x, np.hanning(600))                         # default n_fft = min(N, 512)
    with pytest.raises(_rs.PanicException):
        _rs.ssq_stft(x, np.ones(1), n_fft=1)                     # ssq_freqs[1] out of bounds (:273)
    with pytest.raises(ValueError, match="Time vector must have at least 2 elements"):
        _rs.cwt(x, t=np.array([0.0]))                            # cwt.rs:68-70
    with pytest.raises(ValueError):
        _rs.ssq_cwt(x, t=np.array([0.0]))                        # ssq_cwt.rs:285-287
    with pytest.raises(TypeError):
        _rs.ssq_cwt(x, ssq_freqs=np.linspace(1, 2, 8))           # a str in the reference (:268)
    with pytest.raises(_rs.PanicException):
        _rs.ssq_cwt(x, scales=np.zeros(0))                       # scales[len-1] (:459)
    assert not issubclass(_rs.PanicException, Exception)         # like pyo3_runtime.PanicException


def test_reference_import_line_runs_unchanged():
    """/root/reference README.md:74-79 and src/ssqueeze/__init__.py:2-3: `from ssqueeze import _rs`."""
    import importlib
    ssq = importlib.import_module("ssqueeze")
    assert ssq._rs is _rs and ssq.__all__ == ["_rs"]
    from ssqueeze import _rs as rs2
    from ssqueeze._rs import ssq_stft
    assert rs2 is _rs and ssq_stft is _rs.ssq_stft
    assert rs2.hello_from_bin() == "Hello from ssqueeze!"


def test_pinned_empty_falls_back_to_pageable_memory(monkeypatch):
    """a failing page-locked allocation (or an array above the limit) must still give a usable ndarray."""
    from ssqueeze_rs_amd import _lib

    class Boom:
        def __init__(self, nbytes):
            raise _lib.SsqHipError("hipHostMalloc failed (test)")

    monkeypatch.setattr(_lib, "device_count", lambda: 1)
    monkeypatch.setattr(_lib, "_PinnedBlock", Boom)
    a = _lib.pinned_empty((3, 5), np.complex64)
    assert a.shape == (3, 5) and a.dtype == np.complex64 and a.flags.writeable
    a[:] = 1
    calls = []

    class Count:
        def __init__(self, nbytes):
            calls.append(nbytes)
            raise _lib.SsqHipError("x")

    monkeypatch.setattr(_lib, "_PinnedBlock", Count)
    monkeypatch.setattr(_lib, "PINNED_RESULT_LIMIT", 16)
    b = _lib.pinned_empty((4, 4), np.float64)            # 128 B > limit: the pool is not even asked
    assert b.shape == (4, 4) and calls == []


def test_upstream_mirror_signatures_match_ssqueezepy():
    """`ssqueeze_rs_amd.upstream` mirrors the vendored upstream's callables: parameter names, order and
    defaults as written at old/ssqueezepy/_stft.py:13-14, :184-185, _ssq_stft.py:13-16, :139-140, _cwt.py:12-15, :321-322,
    _ssq_cwt.py:12-17, :313.  Decided before the GPU is touched: options outside the built subset raise ValueError."""
    from ssqueeze_rs_amd import upstream as up

    def sig(fn):
        return [(k, v.default) for k, v in inspect.signature(fn).parameters.items()][1:]

    assert sig(up.stft) == [("window", None), ("n_fft", None), ("win_len", None), ("hop_len", 1), ("fs", None),
                            ("t", None), ("padtype", "reflect"), ("modulated", True), ("derivative", False),
                            ("dtype", None)]
    assert sig(up.istft) == [("window", None), ("n_fft", None), ("win_len", None), ("hop_len", 1), ("N", None),
                             ("modulated", True), ("win_exp", 1)]
    assert sig(up.ssq_stft) == [("window", None), ("n_fft", None), ("win_len", None), ("hop_len", 1), ("fs", None),
                                ("t", None), ("modulated", True), ("ssq_freqs", None), ("padtype", "reflect"),
                                ("squeezing", "sum"), ("gamma", None), ("preserve_transform", None), ("dtype", None),
                                ("astensor", True), ("flipud", False), ("get_w", False), ("get_dWx", False)]
    assert sig(up.issq_stft) == [("window", None), ("cc", None), ("cw", None), ("n_fft", None), ("win_len", None),
                                 ("hop_len", 1), ("modulated", True)]
    assert sig(up.cwt) == [("wavelet", "gmw"), ("scales", "log-piecewise"), ("fs", None), ("t", None), ("nv", 32),
                           ("l1_norm", True), ("derivative", False), ("padtype", "reflect"), ("rpadded", False),
                           ("vectorized", True), ("astensor", True), ("cache_wavelet", None), ("order", 0),
                           ("average", None), ("nan_checks", None), ("patience", 0)]
    assert sig(up.icwt) == [("wavelet", "gmw"), ("scales", "log-piecewise"), ("nv", None), ("one_int", True),
                            ("x_len", None), ("x_mean", 0), ("padtype", "reflect"), ("rpadded", False), ("l1_norm", True)]
    assert sig(up.ssq_cwt) == [("wavelet", "gmw"), ("scales", "log-piecewise"), ("nv", None), ("fs", None), ("t", None),
                               ("ssq_freqs", None), ("padtype", "reflect"), ("squeezing", "sum"), ("maprange", "peak"),
                               ("difftype", "trig"), ("difforder", None), ("gamma", None), ("vectorized", True),
                               ("preserve_transform", None), ("astensor", True), ("order", 0), ("nan_checks", None),
                               ("patience", 0), ("flipud", True), ("cache_wavelet", None), ("get_w", False),
                               ("get_dWx", False)]
    assert sig(up.issq_cwt) == [("wavelet", "gmw"), ("cc", None), ("cw", None)]
    x = np.zeros(64)
    for call in (lambda: up.cwt(x), lambda: up.ssq_cwt(x), lambda: up.cwt(x, "bump", scales=2.0 ** np.arange(1, 5.0)),
                 lambda: up.stft(x, "hann"), lambda: up.ssq_cwt(x, scales=np.array([1.0, 2.0, 5.0])),
                 lambda: up.issq_stft(np.zeros((33, 64), complex), np.hanning(64), hop_len=2),
                 lambda: up.cwt(x, scales=2.0 ** np.arange(1, 5.0), order=1)):
        with pytest.raises(ValueError):
            call()
