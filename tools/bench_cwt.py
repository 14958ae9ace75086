"""Time the ssq_cwt plan on BASELINE config 4 (1 x 2^20, Morlet, 256 log scales, fp32) -- secondary metric.
    python tools/bench_cwt.py [--log2n 20] [--na 256] [--dtype f32|f64] [--steps 3]
Prints one JSON line (bins/s, algorithmic GB/s against the 8 TB/s roof)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssqueeze_rs_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, default=20)
ap.add_argument("--na", type=int, default=256)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--batch", type=int, default=1)
a = ap.parse_args()
lib = _lib.load()
N, na, B = 1 << a.log2n, a.na, a.batch
code = _lib.SSQ_F32 if a.dtype == "f32" else _lib.SSQ_F64
es = 4 if code == _lib.SSQ_F32 else 8
scales = 2.0 ** np.linspace(1, a.log2n - 1, na)
plan = C.c_void_p()
_lib.check(lib.ssq_cwt_plan_create(C.byref(plan), code, N, _lib.WAVELET["morlet"], scales.ctypes.data_as(C.c_void_p),
                                   na, 1.0, 0))
wsb = lib.ssq_cwt_plan_workspace_bytes(plan, B)
dx, dT, ws = C.c_void_p(), C.c_void_p(), C.c_void_p()
_lib.check(lib.ssq_dev_malloc(C.byref(dx), B * N * es))
_lib.check(lib.ssq_dev_malloc(C.byref(dT), B * na * N * 2 * es))
_lib.check(lib.ssq_dev_malloc(C.byref(ws), wsb))
from ssqueeze_rs_amd.synth import synth_signal  # noqa: E402
x = np.concatenate([synth_signal(N, b, np.float32 if es == 4 else np.float64) for b in range(B)])   # SURVEY §8d workload
_lib.check(lib.ssq_memcpy_h2d(dx, x.ctypes.data_as(C.c_void_p), x.nbytes, None))


def run():
    _lib.check(lib.ssq_cwt_plan_exec_ssq(plan, dx, B, 0, 0, 0, 1, -1.0, dT, None, None, None, ws, wsb, None))
    _lib.check(lib.ssq_device_sync())


run()
t0 = time.perf_counter()
for _ in range(a.steps):
    run()
dt = (time.perf_counter() - t0) / a.steps
bins = B * na * N
alg = B * (es * N + 2 * es * na * N)
print(json.dumps({"workload": f"ssq_cwt morlet na={na} batch={B} x 2^{a.log2n} {a.dtype}", "ms": dt * 1e3,
                  "bins_per_s": bins / dt, "alg_GBps": alg / dt / 1e9, "frac_of_8TBps": alg / dt / 8e12,
                  "workspace_GB": wsb / 1e9}))
