"""The C-ABI collective (include/ssq_hip.h: ssq_rccl_*, ssq_gather_shards): RCCL dlopen'd, no torch on the data path.
CPU: the symbols load and fail loudly without a device.  GPU (one rank: all a 1-GPU box can run): communicator of
world size 1, all-gather of a real Tx shard equals the shard."""
import ctypes as C

import numpy as np
import pytest

from ssqueeze_rs_amd import _lib


def test_rccl_entry_points_fail_loudly_without_a_gpu():
    lib = _lib.load()
    assert lib.ssq_rccl_available() in (0, 1)
    if _lib.device_count() > 0:
        pytest.skip("a device is visible")
    buf = C.create_string_buffer(128)
    assert lib.ssq_rccl_unique_id(buf) != 0 and lib.ssq_last_error()
    comm = C.c_void_p()
    assert lib.ssq_rccl_comm_init(C.byref(comm), 0, buf, 0) != 0               # bad n_ranks: refused before RCCL is asked
    assert lib.ssq_gather_shards(None, None, None, 16, None) != 0
    assert b"comm is NULL" in lib.ssq_last_error()


@pytest.mark.gpu
def test_single_rank_gather_of_a_tx_shard():
    from oracle import ssq_oracle as o
    lib = _lib.load()
    assert lib.ssq_rccl_available() == 1
    N, n_fft, hop, B = 1 << 14, 1024, 256, 3
    x = np.stack([o.synth_signal(N, b, np.float32) for b in range(B)])
    win = np.hanning(n_fft)
    plan = C.c_void_p()
    _lib.check(lib.ssq_stft_plan_create(C.byref(plan), _lib.SSQ_F32, N, win.ctypes.data_as(C.c_void_p), n_fft, hop, 1.0,
                                        0, 0, -1.0, 0))
    nbytes = B * 513 * ((N - 1) // hop + 1) * 8
    dx, dT, dG = C.c_void_p(), C.c_void_p(), C.c_void_p()
    for p, n in ((dx, x.nbytes), (dT, nbytes), (dG, nbytes)):
        _lib.check(lib.ssq_dev_malloc(C.byref(p), n))
    st = C.c_void_p()
    _lib.check(lib.ssq_stream_create(C.byref(st)))
    idb = C.create_string_buffer(128)
    _lib.check(lib.ssq_rccl_unique_id(idb))
    comm = C.c_void_p()
    _lib.check(lib.ssq_rccl_comm_init(C.byref(comm), 1, idb, 0))
    try:
        n, me = C.c_int(-1), C.c_int(-1)
        _lib.check(lib.ssq_rccl_comm_info(comm, C.byref(n), C.byref(me)))
        assert (n.value, me.value) == (1, 0)
        _lib.check(lib.ssq_memcpy_h2d(dx, x.ctypes.data_as(C.c_void_p), x.nbytes, st))
        _lib.check(lib.ssq_stft_plan_exec(plan, _lib.OUT_TX, dx, B, dT, None, 0, st))
        _lib.check(lib.ssq_dev_memset(dG, 0, nbytes, st))
        _lib.check(lib.ssq_gather_shards(comm, dT, dG, nbytes, st))           # same stream: ordered behind the kernels
        a, g = np.empty(nbytes, np.uint8), np.empty(nbytes, np.uint8)
        _lib.check(lib.ssq_memcpy_d2h(a.ctypes.data_as(C.c_void_p), dT, nbytes, st))
        _lib.check(lib.ssq_memcpy_d2h(g.ctypes.data_as(C.c_void_p), dG, nbytes, st))
        _lib.check(lib.ssq_stream_sync(st))
        assert np.array_equal(a, g) and np.abs(a.view(np.complex64)).max() > 0
    finally:
        lib.ssq_rccl_comm_destroy(comm)
        for p in (dx, dT, dG):
            lib.ssq_dev_free(p)
        lib.ssq_stream_destroy(st)
        lib.ssq_stft_plan_destroy(plan)
