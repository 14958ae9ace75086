"""HIP-graph capture of the plan execs (include/ssq_hip.h: ssq_graph_*): the STFT pass (interior + edge launch) and the
ssq_cwt call (a dozen launches with a side-stream fork / join for the Tx clear) replayed as ONE launch give the bits of the
direct calls, also after the input buffer's contents change."""
import ctypes as C

import numpy as np
import pytest

from oracle import ssq_oracle as o
from ssqueeze_rs_amd import _lib

pytestmark = pytest.mark.gpu


def _dev(lib, nbytes):
    p = C.c_void_p()
    _lib.check(lib.ssq_dev_malloc(C.byref(p), nbytes))
    return p


def test_stft_pass_and_ssq_cwt_call_replay_from_a_graph():
    lib = _lib.load()
    st = C.c_void_p()
    _lib.check(lib.ssq_stream_create(C.byref(st)))
    # ---- ssq_stft, batch 40 x 2^16: the two-launch pass ----
    N, B, n_fft, hop = 1 << 16, 40, 1024, 256
    win = np.hanning(n_fft)
    plan = C.c_void_p()
    _lib.check(lib.ssq_stft_plan_create(C.byref(plan), _lib.SSQ_F32, N, win.ctypes.data_as(C.c_void_p), n_fft, hop, 1.0, 0, 0,
                                        -1.0, 0))
    bins = 513 * ((N - 1) // hop + 1)
    dx, dT = _dev(lib, B * N * 4), _dev(lib, B * bins * 8)
    xs = [np.stack([o.synth_signal(N, 100 * r + b, np.float32) for b in range(B)]) for r in range(2)]

    def fetch():
        out = np.empty(B * bins, np.complex64)
        _lib.check(lib.ssq_memcpy_d2h(out.ctypes.data_as(C.c_void_p), dT, out.nbytes, st))
        _lib.check(lib.ssq_stream_sync(st))
        return out

    def direct(x):
        _lib.check(lib.ssq_memcpy_h2d(dx, x.ctypes.data_as(C.c_void_p), x.nbytes, st))
        _lib.check(lib.ssq_stft_plan_exec(plan, _lib.OUT_TX, dx, B, dT, None, 0, st))
        return fetch()

    want = [direct(x) for x in xs]
    _lib.check(lib.ssq_graph_capture_begin(st))
    _lib.check(lib.ssq_stft_plan_exec(plan, _lib.OUT_TX, dx, B, dT, None, 0, st))
    g = C.c_void_p()
    _lib.check(lib.ssq_graph_capture_end(st, C.byref(g)))
    try:
        for x, w in zip(xs, want):
            _lib.check(lib.ssq_memcpy_h2d(dx, x.ctypes.data_as(C.c_void_p), x.nbytes, st))
            _lib.check(lib.ssq_dev_memset(dT, 0xFF, B * bins * 8, st))
            _lib.check(lib.ssq_graph_launch(g, st))
            assert np.array_equal(fetch(), w)
    finally:
        lib.ssq_graph_destroy(g)
        lib.ssq_stft_plan_destroy(plan)
        lib.ssq_dev_free(dx)
        lib.ssq_dev_free(dT)
    # ---- ssq_cwt, 2^18 x 64 scales fp32: time tiles + side-stream clear ----
    N, na = 1 << 18, 64
    scales = 2.0 ** np.linspace(1, 17, na)
    cplan = C.c_void_p()
    _lib.check(lib.ssq_cwt_plan_create(C.byref(cplan), _lib.SSQ_F32, N, _lib.WAVELET["morlet"],
                                       scales.ctypes.data_as(C.c_void_p), na, 1.0, 0))
    wsb = lib.ssq_cwt_plan_workspace_bytes(cplan, 1)
    dx, dT, ws = _dev(lib, N * 4), _dev(lib, na * N * 8), _dev(lib, wsb)
    xs = [o.synth_signal(N, 7 + r, np.float32) for r in range(2)]

    def cfetch():
        out = np.empty(na * N, np.complex64)
        _lib.check(lib.ssq_memcpy_d2h(out.ctypes.data_as(C.c_void_p), dT, out.nbytes, st))
        _lib.check(lib.ssq_stream_sync(st))
        return out

    def cexec():
        _lib.check(lib.ssq_cwt_plan_exec_ssq(cplan, dx, 1, 0, 0, 0, 1, -1.0, dT, None, None, None, ws, wsb, st))

    want = []
    for x in xs:
        _lib.check(lib.ssq_memcpy_h2d(dx, x.ctypes.data_as(C.c_void_p), x.nbytes, st))
        cexec()
        want.append(cfetch())
    _lib.check(lib.ssq_graph_capture_begin(st))
    cexec()
    g = C.c_void_p()
    _lib.check(lib.ssq_graph_capture_end(st, C.byref(g)))
    try:
        for x, w in zip(xs, want):
            _lib.check(lib.ssq_memcpy_h2d(dx, x.ctypes.data_as(C.c_void_p), x.nbytes, st))
            _lib.check(lib.ssq_graph_launch(g, st))
            assert np.array_equal(cfetch(), w)
    finally:
        lib.ssq_graph_destroy(g)
        lib.ssq_cwt_plan_destroy(cplan)
        for p in (dx, dT, ws):
            lib.ssq_dev_free(p)
        lib.ssq_stream_destroy(st)
