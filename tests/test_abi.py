"""CPU tests of the drop-in boundary: libssq_hip.so loads, exports every symbol include/ssq_hip.h
declares, and its host-side (fp64, no GPU) entry points agree with the oracle.  No compute calls."""
import ctypes as C

import numpy as np
import pytest

from oracle import ssq_oracle as o
from ssqueeze_rs_amd import _lib


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _lib.header_symbols()
    assert len(declared) >= 40
    assert set(declared) == set(_lib._SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ssq_hello_from_bin() == b"Hello from ssqueeze!"          # lib.rs:16-19


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def test_shape_helpers():
    lib = _lib.load()
    nf, nfr = C.c_int64(), C.c_int64()
    assert lib.ssq_stft_shape(1000, 256, 64, C.byref(nf), C.byref(nfr)) == 0
    assert (nf.value, nfr.value) == (129, 16)
    assert lib.ssq_stft_shape(1 << 20, 1024, 256, C.byref(nf), C.byref(nfr)) == 0
    assert (nf.value, nfr.value) == (513, 4096)
    assert lib.ssq_stft_shape(0, 256, 64, C.byref(nf), C.byref(nfr)) != 0
    assert lib.ssq_stft_shape(10, 256, 0, C.byref(nf), C.byref(nfr)) != 0
    assert b"divide by zero" in lib.ssq_last_error()
    P, n1 = C.c_int64(), C.c_int64()
    for N in (1, 5, 1000, 2730, 2731, 1 << 20, 1 << 22):
        assert lib.ssq_cwt_pad_len(N, C.byref(P), C.byref(n1)) == 0
        assert P.value == o.next_power_of_2(N + N // 2) and n1.value == (P.value - N) // 2


def test_window_helpers_match_oracle():
    lib = _lib.load()
    for L, n in ((200, 256), (300, 256), (256, 256), (7, 8), (100, 100), (63, 100)):
        w = np.hanning(L) + 0.01
        out = np.empty(n)
        assert lib.ssq_size_window(_vp(w), L, n, _vp(out)) == 0
        assert np.array_equal(out, o.size_window(w, n))
        d = np.empty(n)
        assert lib.ssq_diff_window(_vp(out), n, _vp(d)) == 0
        ref = o.diff_window(out)
        assert np.abs(d - ref).max() <= 1e-13 * max(np.abs(ref).max(), 1e-300)


def test_scales_and_freqs_match_oracle():
    lib = _lib.load()
    for N, nv in ((1000, 16), (4096, 8), (1 << 20, 32), (300, 2), (3, 32)):
        for simd in (0, 1):
            na = C.c_int64()
            assert lib.ssq_log_scales(N, nv, simd, C.byref(na), None) == 0
            ref = o.log_scales(N, nv, simd_variant=bool(simd))
            assert na.value == ref.shape[0]
            if na.value:
                s = np.empty(na.value)
                assert lib.ssq_log_scales(N, nv, simd, C.byref(na), _vp(s)) == 0
                assert np.array_equal(s, ref), (N, nv, simd)
    sc = o.log_scales(4096, 8)
    for maprange, name in ((0, "peak"), (1, "maximal")):
        for dist, dname in ((0, "log"), (1, "linear")):
            f = np.empty(sc.shape[0])
            assert lib.ssq_cwt_ssq_freqs(_vp(sc), sc.shape[0], 4096, 0.01, maprange, dist, _vp(f)) == 0
            if maprange:
                fmin, fmax = 1.0 / (4096 * 0.01), 0.5 / 0.01
            else:
                fmin, fmax = 1.0 / sc[-1], 1.0 / sc[0]
            assert np.array_equal(f, o.cwt_ssq_freqs(sc.shape[0], fmin, fmax, dname))


def test_plan_and_compute_calls_fail_loudly_without_a_gpu():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    lib = _lib.load()
    plan = C.c_void_p()
    win = np.hanning(256)
    rc = lib.ssq_stft_plan_create(C.byref(plan), _lib.SSQ_F32, 4096, _vp(win), 256, 64, 1.0, 0, 0, -1.0, 0)
    assert rc != 0 and lib.ssq_last_error()
    from ssqueeze_rs_amd import _rs
    with pytest.raises(_lib.SsqHipError):
        _rs.stft(np.zeros(1000), 256, 64, win, "reflect")
    with pytest.raises(_lib.SsqHipError):
        _rs.ssq_cwt(np.zeros(1000), wavelet="morlet", nv=4)
