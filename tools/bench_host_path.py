"""PCIe-inclusive rate of the host-pointer entry point (what `_rs.ssq_stft` on NumPy arrays costs end to end):
    python tools/bench_host_path.py [--batch 8]
One JSON line; never the bench's `value` (bench.py times device-resident inputs)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssqueeze_rs_amd import _rs  # noqa: E402
from ssqueeze_rs_amd.synth import synth_signal  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
a = ap.parse_args()
N = 1 << 20
xb = np.stack([synth_signal(N, b, np.float32) for b in range(a.batch)])
win = np.hanning(1024)
_rs.ssq_stft(xb, win, n_fft=1024, hop_len=256)
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    Tx, f = _rs.ssq_stft(xb, win, n_fft=1024, hop_len=256)
dt = (time.perf_counter() - t0) / reps
bins = a.batch * 513 * 4096
print(json.dumps({"workload": f"_rs.ssq_stft on host arrays, batch={a.batch} x 2^20 fp32 (plan + malloc + H2D + kernels + D2H)",
                  "ms": dt * 1e3, "tf_bins_per_s": bins / dt, "host_bytes_moved_GBps": a.batch * (4 * N + 8 * 513 * 4096) / dt / 1e9}))
