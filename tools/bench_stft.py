"""Time the ssq_stft plan for any dtype / n_fft / hop / batch (secondary measurements; bench.py is the headline).
    python tools/bench_stft.py [--dtype f64] [--n-fft 1024] [--hop 256] [--batch 64] [--log2n 20] [--steps 10] [--out tx|sx]
Prints one JSON line (TF-bins/s, algorithmic GB/s against the 8 TB/s roof)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssqueeze_rs_amd import _lib  # noqa: E402
from ssqueeze_rs_amd.synth import synth_signal  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f64")
ap.add_argument("--n-fft", type=int, default=1024)
ap.add_argument("--hop", type=int, default=256)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--log2n", type=int, default=20)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--out", default="tx")
a = ap.parse_args()
lib = _lib.load()
N, B = 1 << a.log2n, a.batch
code = _lib.SSQ_F32 if a.dtype == "f32" else _lib.SSQ_F64
npd = np.float32 if a.dtype == "f32" else np.float64
es = 4 if a.dtype == "f32" else 8
nf, nfr = a.n_fft // 2 + 1, (N - 1) // a.hop + 1
win = np.hanning(a.n_fft)
plan = C.c_void_p()
_lib.check(lib.ssq_stft_plan_create(C.byref(plan), code, N, win.ctypes.data_as(C.c_void_p), a.n_fft, a.hop, 1.0, 0, 0,
                                    -1.0, 0))
kind = _lib.OUT_TX if a.out == "tx" else _lib.OUT_SX
ws = lib.ssq_stft_plan_workspace_bytes(plan, B, kind)
dx, do, dw = C.c_void_p(), C.c_void_p(), C.c_void_p()
_lib.check(lib.ssq_dev_malloc(C.byref(dx), B * N * es))
_lib.check(lib.ssq_dev_malloc(C.byref(do), B * nf * nfr * 2 * es))
_lib.check(lib.ssq_dev_malloc(C.byref(dw), max(ws, 16)))
x = np.stack([synth_signal(N, b % 8, npd) for b in range(min(B, 8))])
for b in range(B):
    _lib.check(lib.ssq_memcpy_h2d(C.c_void_p(dx.value + b * N * es), x[b % x.shape[0]].ctypes.data_as(C.c_void_p), N * es, None))


def run():
    _lib.check(lib.ssq_stft_plan_exec(plan, kind, dx, B, do, dw, ws, None))


run()
_lib.check(lib.ssq_device_sync())
t0 = time.perf_counter()
for _ in range(a.steps):
    run()
_lib.check(lib.ssq_device_sync())
dt = (time.perf_counter() - t0) / a.steps
alg = B * (es * N + 2 * es * nf * nfr)
print(json.dumps({"workload": f"ssq_stft({a.out}) {a.dtype} batch={B} x 2^{a.log2n} n_fft={a.n_fft} hop={a.hop}",
                  "fused": int(lib.ssq_stft_plan_is_fused(plan)), "ms": dt * 1e3, "tf_bins_per_s": B * nf * nfr / dt,
                  "alg_GBps": alg / dt / 1e9, "frac_of_8TBps": alg / dt / 8e12}))
