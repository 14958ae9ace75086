"""PCIe-inclusive rate of the host-pointer entry point (what `_rs.ssq_stft` on NumPy arrays costs end to end):
    python tools/bench_host_path.py [--batch 8] [--reps 10]
Two lines of JSON: the drop-in call as a user makes it (results in the library's pinned pool), and the same call
with the results forced into ordinary pageable NumPy memory (what round 1 measured).  Never the bench's `value`
(bench.py times device-resident inputs)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssqueeze_rs_amd import _lib, _rs  # noqa: E402
from ssqueeze_rs_amd.synth import synth_signal  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
N = 1 << 20
xb = np.stack([synth_signal(N, b, np.float32) for b in range(a.batch)])
win = np.hanning(1024)
bins = a.batch * 513 * 4096
moved = a.batch * (4 * N + 8 * 513 * 4096)


def timed(fn, what):
    fn()
    fn()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        fn()
    dt = (time.perf_counter() - t0) / a.reps
    print(json.dumps({"workload": f"_rs.ssq_stft on host arrays, batch={a.batch} x 2^20 fp32, {what}", "ms": dt * 1e3,
                      "ms_per_signal": dt * 1e3 / a.batch, "tf_bins_per_s": bins / dt,
                      "host_bytes_moved_GBps": moved / dt / 1e9}))


timed(lambda: _rs.ssq_stft(xb, win, n_fft=1024, hop_len=256),
      "cached plan + device buffers, two-stream pipeline, results in the pinned pool")

lib = _lib.load()
Tx = np.empty((a.batch, 513, 4096), dtype=np.complex64)          # pageable destination, page-faulted once
Tx[:] = 0
f = np.empty(513)


def pageable():
    _lib.check(lib.ssq_ssq_stft_host(_lib.SSQ_F32, xb.ctypes.data_as(C.c_void_p), a.batch, N, win.ctypes.data_as(C.c_void_p),
                                     1024, 256, 1.0, 0, 0, -1.0, Tx.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p),
                                     None, None, None))


timed(pageable, "same call, results into pageable NumPy memory")
