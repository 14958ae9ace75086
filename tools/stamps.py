"""Per-phase cycle shares of the fused kernel from a -DSSQ_STAMPS diagnostic build.
    python -m ssqueeze_rs_amd.build --stamps && SSQ_HIP_LIB=ssqueeze_rs_amd/libssq_hip_diag.so python tools/stamps.py
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssqueeze_rs_amd import _lib  # noqa: E402

NAMES = ["wait prefetch + window", "decode + issue loads", "FFT (3 passes, 2 LDS exchanges)", "partner shuffles",
         "unpack + phase + bins", "column reduce + scale", "atomics (+drain)", "barrier 1 wait", "read-out",
         "barrier 2 wait", "pk: window + pass 0", "pk: exchange 1 (LDS)", "pk: loads issue + pass 1",
         "pk: exchange 2 (permlane)", "-", "-"]
lib = _lib.load()
B, N = 64, 1 << 20
# SSQ_STAMP_NFFT / _HOP: another frame length (the generic 8-wave template: 12 stamps per wave, one launch in the
# any-length modes)
n_fft, hop = int(os.environ.get("SSQ_STAMP_NFFT", "1024")), int(os.environ.get("SSQ_STAMP_HOP", "256"))
F64 = os.environ.get("SSQ_STAMP_DTYPE", "f32") == "f64"
ES = 8 if F64 else 4
GENERIC = F64 or n_fft != 1024 or os.environ.get("SSQ_HIOCC", "1") == "0"
STRIDE = 12 if GENERIC else 16
buf = C.c_void_p()
W = int(os.environ.get("SSQ_STAMP_WAVES", ("4" if F64 else "8") if GENERIC else "16"))     # waves per block of the kernel under test
nwaves = 256 * W
_lib.check(lib.ssq_dev_malloc(C.byref(buf), nwaves * STRIDE * 8))
_lib.check(lib.ssq_dev_memset(buf, 0, nwaves * STRIDE * 8, None))
os.environ["SSQ_STAMPS_PTR"] = str(buf.value)
win = np.hanning(n_fft)
plan = C.c_void_p()
_lib.check(lib.ssq_stft_plan_create(C.byref(plan), 1 if F64 else 0, N, win.ctypes.data_as(C.c_void_p), n_fft, hop, 1.0, 0, 0, -1.0, 0))
dx, do = C.c_void_p(), C.c_void_p()
_lib.check(lib.ssq_dev_malloc(C.byref(dx), B * N * ES))
nfr = (N - 1) // hop + 1
_lib.check(lib.ssq_dev_malloc(C.byref(do), B * (n_fft // 2 + 1) * nfr * 2 * ES))
x = np.random.default_rng(0).standard_normal(B * N).astype(np.float64 if F64 else np.float32)
_lib.check(lib.ssq_memcpy_h2d(dx, x.ctypes.data_as(C.c_void_p), x.nbytes, None))
for _ in range(2):
    _lib.check(lib.ssq_stft_plan_exec(plan, 0, dx, B, do, None, 0, None))
_lib.check(lib.ssq_device_sync())
out = np.zeros(nwaves * STRIDE, dtype=np.uint64)
_lib.check(lib.ssq_memcpy_d2h(out.ctypes.data_as(C.c_void_p), buf, out.nbytes, None))
_lib.check(lib.ssq_device_sync())
acc = out.reshape(nwaves, STRIDE).astype(np.float64)
# the edge-tile launch (B*3 one-tile blocks) overwrites the first blocks' slots: keep interior-only blocks
if n_fft == 1024 and not F64:
    acc = acc[W * (B * 3 + 8):]
tot = acc.sum(1).mean()
frames_per_wave = B * (nfr - (3 * 16 if n_fft == 1024 else 0)) / nwaves
print(f"mean cycles per wave {tot:.0f}; per frame {tot / frames_per_wave:.0f}")
for i, n in enumerate(NAMES[:STRIDE if GENERIC else 14]):
    print(f"{n:36s} {acc[:, i].mean() / frames_per_wave:9.0f} cyc/frame  {100 * acc[:, i].mean() / tot:5.1f} %")
