"""Deterministic synthetic workload: signal b of a batch = three sines + a linear chirp + 1e-3 white
noise, `rng = np.random.default_rng(1234 + b)`.  Used by bench.py, tools/ and (re-exported by the oracle) the tests;
it is input data, not part of the computation."""
import math

import numpy as np


def synth_signal(N: int, b: int = 0, dtype=np.float64) -> np.ndarray:
    """Signal b of a batch: 3 sines + linear chirp + 1e-3 white noise, seed 1234+b."""
    rng = np.random.default_rng(1234 + b)
    n = np.arange(N, dtype=np.float64)
    x = np.zeros(N, dtype=np.float64)
    for _ in range(3):
        f = rng.uniform(0.01, 0.45)
        A = rng.uniform(0.5, 1.0)
        ph = rng.uniform(0.0, 2.0 * math.pi)
        x += A * np.sin(2.0 * math.pi * f * n + ph)
    f0, f1 = sorted(rng.uniform(0.01, 0.45, size=2))
    Ac = rng.uniform(0.5, 1.0)
    x += Ac * np.sin(2.0 * math.pi * (f0 * n + 0.5 * (f1 - f0) / N * n * n))
    x += 1e-3 * rng.standard_normal(N)
    return x.astype(dtype)
