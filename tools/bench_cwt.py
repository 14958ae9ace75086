"""Time the ssq_cwt plan on BASELINE config 4 (1 x 2^20, Morlet, 256 log scales, fp32) -- secondary metric.
    python tools/bench_cwt.py [--log2n 20] [--na 256] [--dtype f32|f64] [--steps 3]
Prints one JSON line: bins/s and BOTH rooflines -- algorithmic bytes against the 8 TB/s HBM roof and the transforms'
flops against the fp32 / fp64 vector roof (MI355X_MICROARCH.md: 157.3 / 78.6 TFLOP/s) -- plus, for C4, the measured
traffic per call from the committed PMC passes (profiles/r02_cwt_traffic.json)."""
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
ap.add_argument("--n", type=int, default=0, help="signal length (default 2^log2n); scales stay 2^linspace(1, log2n - 1)")
ap.add_argument("--na", type=int, default=256)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--wavelet", default="morlet")
ap.add_argument("--mode", default="ssq", help="ssq (ssq_cwt) | cwt (Wx and dWx out, L1 norm, unpadded) | cwt1 (Wx only)")
a = ap.parse_args()
lib = _lib.load()
N, na, B = (a.n if a.n > 0 else 1 << a.log2n), a.na, a.batch
n_label = f"2^{a.log2n}" if a.n <= 0 else str(N)
code = _lib.SSQ_F32 if a.dtype == "f32" else _lib.SSQ_F64
es = 4 if code == _lib.SSQ_F32 else 8
scales = 2.0 ** np.linspace(1, a.log2n - 1, na)
plan = C.c_void_p()
_lib.check(lib.ssq_cwt_plan_create(C.byref(plan), code, N, _lib.WAVELET[a.wavelet], scales.ctypes.data_as(C.c_void_p),
                                   na, 1.0, 0))
wsb = lib.ssq_cwt_plan_workspace_bytes(plan, B)
dx, dT, ws = C.c_void_p(), C.c_void_p(), C.c_void_p()
_lib.check(lib.ssq_dev_malloc(C.byref(dx), B * N * es))
_lib.check(lib.ssq_dev_malloc(C.byref(dT), B * na * N * 2 * es))
_lib.check(lib.ssq_dev_malloc(C.byref(ws), wsb))
from ssqueeze_rs_amd.synth import synth_signal  # noqa: E402
x = np.concatenate([synth_signal(N, b, np.float32 if es == 4 else np.float64) for b in range(B)])   # SURVEY §8d workload
_lib.check(lib.ssq_memcpy_h2d(dx, x.ctypes.data_as(C.c_void_p), x.nbytes, None))


dW = C.c_void_p()
if a.mode == "cwt":
    _lib.check(lib.ssq_dev_malloc(C.byref(dW), B * na * N * 2 * es))


def run():
    if a.mode in ("cwt", "cwt1"):
        _lib.check(lib.ssq_cwt_plan_exec_cwt(plan, dx, B, 1, 0, dT, dW if a.mode == "cwt" else None, ws, wsb, None))
    else:
        _lib.check(lib.ssq_cwt_plan_exec_ssq(plan, dx, B, 0, 0, 0, 1, -1.0, dT, None, None, None, ws, wsb, None))
    _lib.check(lib.ssq_device_sync())


run()
t0 = time.perf_counter()
for _ in range(a.steps):
    run()
dt = (time.perf_counter() - t0) / a.steps
bins = B * na * N
alg = B * (es * N + 2 * es * na * N)
# flops of the algorithm as the reference runs it (cwt.rs:228-310): one forward and 2*na inverse FFTs of the padded
# length P (5 P log2 P each) + the wavelet multiply (2*na * 6 P... counted as 2 flops per real multiply: 4 P) + the phase
# transform and bin (~30 flops per bin)
P = 1 << int(np.ceil(np.log2(N + N // 2)))
lp = int(np.log2(P))
flops = B * ((1 + 2 * na) * 5.0 * P * lp + 2 * na * 4.0 * P + 30.0 * na * N)
vec_peak = 157.3e12 if es == 4 else 78.6e12
traffic = None
tf = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_cwt_traffic.json")
if a.log2n == 20 and na == 256 and es == 4 and a.wavelet == "morlet" and os.path.exists(tf):
    with open(tf) as fh:
        traffic = json.load(fh)["total_GB_per_call"] * 1e9 * B
print(json.dumps({"workload": f"{a.mode if a.mode != 'ssq' else 'ssq_cwt'} {a.wavelet} na={na} batch={B} x {n_label} {a.dtype}", "ms": dt * 1e3,
                  "bins_per_s": bins / dt, "alg_GBps": alg / dt / 1e9, "frac_of_8TBps": alg / dt / 8e12,
                  "roofline_hbm": {"bound": "hbm", "achieved": alg / dt / 1e9, "peak": 8000.0, "unit": "GB/s",
                                   "frac": alg / dt / 8e12, "traffic": traffic,
                                   "traffic_over_algorithmic": (traffic / alg) if traffic else None},
                  "roofline_vector": {"bound": "valu", "achieved": flops / dt / 1e12, "peak": vec_peak / 1e12,
                                      "unit": "TFLOP/s", "frac": flops / dt / vec_peak},
                  "workspace_GB": wsb / 1e9}))
