"""Interleaved A/B of kernel variants in ONE process on ONE device (cdna_hip_programming.md rule 24).

    python tools/ab_inproc.py [--batch 256] [--rounds 5] [--steps 10] [--dtype f32] VARIANT [VARIANT ...]

A VARIANT is a comma-separated list of environment switches the launch code reads PER LAUNCH
(e.g. "SSQ_FREERUN=0" "SSQ_FREERUN=1"), or "-" for none.  Every round times every variant (HIP events on the launch
stream, `steps` launches each); prints per-variant median / min of the per-step kernel time and the fraction of
the 8 TB/s roof.  Also checks that every variant produces the same bits as the first one (signal 0 and the last).
"""
import argparse
import zlib
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssqueeze_rs_amd import _lib  # noqa: E402
from ssqueeze_rs_amd.synth import synth_signal  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="+")
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--log2n", type=int, default=20)
ap.add_argument("--n-fft", type=int, default=1024)
ap.add_argument("--hop", type=int, default=256)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--distinct", type=int, default=16)
a = ap.parse_args()

lib = _lib.load()
N, B = 1 << a.log2n, a.batch
f32 = a.dtype == "f32"
es = 4 if f32 else 8
npd = np.float32 if f32 else np.float64
nf, nfr = a.n_fft // 2 + 1, (N - 1) // a.hop + 1
bins = nf * nfr
win = np.hanning(a.n_fft)
plan = C.c_void_p()
_lib.check(lib.ssq_stft_plan_create(C.byref(plan), _lib.SSQ_F32 if f32 else _lib.SSQ_F64, N,
                                    win.ctypes.data_as(C.c_void_p), a.n_fft, a.hop, 1.0, 0, 0, -1.0, 0))
stream = C.c_void_p()
_lib.check(lib.ssq_stream_create(C.byref(stream)))
dx, do = C.c_void_p(), C.c_void_p()
_lib.check(lib.ssq_dev_malloc(C.byref(dx), B * N * es))
_lib.check(lib.ssq_dev_malloc(C.byref(do), B * bins * 2 * es))
nd = min(a.distinct, B)
x = np.stack([synth_signal(N, b, npd) for b in range(nd)])
for b in range(B):
    _lib.check(lib.ssq_memcpy_h2d(C.c_void_p(dx.value + b * N * es), x[b % nd].ctypes.data_as(C.c_void_p), N * es, stream))
_lib.check(lib.ssq_stream_sync(stream))


def set_variant(v):
    for kv in all_keys:
        os.environ.pop(kv, None)
    if v != "-":
        for item in v.split(","):
            k, val = item.split("=")
            os.environ[k] = val


all_keys = sorted({item.split("=")[0] for v in a.variants if v != "-" for item in v.split(",")})


def run():
    _lib.check(lib.ssq_stft_plan_exec(plan, _lib.OUT_TX, dx, B, do, None, 0, stream))


def fetch(b):
    out = np.empty((nf, nfr), dtype=np.complex64 if f32 else np.complex128)
    _lib.check(lib.ssq_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(do.value + b * bins * 2 * es), out.nbytes, stream))
    _lib.check(lib.ssq_stream_sync(stream))
    return out


times = {v: [] for v in a.variants}
crc = {}
ref = None
same = {}
for v in a.variants:                      # warm-up + bit comparison
    set_variant(v)
    run()
    run()
    _lib.check(lib.ssq_stream_sync(stream))
    got = (fetch(0), fetch(B - 1))
    crc[v] = "%08x" % (zlib.crc32(got[0].tobytes()) ^ zlib.crc32(got[1].tobytes()))
    if ref is None:
        ref = got
    same[v] = bool(np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]))
e0, e1 = C.c_void_p(), C.c_void_p()
_lib.check(lib.ssq_event_create(C.byref(e0)))
_lib.check(lib.ssq_event_create(C.byref(e1)))
for r in range(a.rounds):
    for v in a.variants:
        set_variant(v)
        _lib.check(lib.ssq_event_record(e0, stream))
        for _ in range(a.steps):
            run()
        _lib.check(lib.ssq_event_record(e1, stream))
        _lib.check(lib.ssq_stream_sync(stream))
        ms = C.c_float(0)
        _lib.check(lib.ssq_event_elapsed_ms(e0, e1, C.byref(ms)))
        times[v].append(ms.value / a.steps)
alg = B * (es * N + 2 * es * bins)
print(f"# ab_inproc: {a.dtype} batch={B} x 2^{a.log2n} n_fft={a.n_fft} hop={a.hop}; {a.rounds} rounds x {a.steps} steps, interleaved")
for v in a.variants:
    t = np.array(times[v])
    print(json.dumps({"variant": v, "ms_median": round(float(np.median(t)), 4), "ms_min": round(float(t.min()), 4),
                      "frac_median": round(alg / (float(np.median(t)) * 1e-3) / 8e12, 4),
                      "same_bits_as_first": same[v], "crc": crc[v], "ms_all": [round(float(z), 4) for z in t]}))
