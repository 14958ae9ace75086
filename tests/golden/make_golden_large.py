"""Full-size ssq_cwt summaries for BASELINE configs 4 and 5 (tests/golden/c4_summary.npz, c5_summary.npz), produced
by the NumPy oracle scale by scale so that the container's memory suffices (the dense oracle.ssq_cwt would hold
~35 GB at C4 and ~140 GB at C5).  Like every fixture here they pin the ORACLE (parity unpinned).

    python tests/golden/make_golden_large.py c4      # 1 x 2^20, 256 scales 2**linspace(1,19,256), Morlet  (~3 min)
    python tests/golden/make_golden_large.py c5      # 1 x 2^22, 256 scales 2**linspace(1,21,256), Morlet  (~15 min)

The streamed loop follows oracle.ssq_cwt line by line (ssq_cwt.rs:329-435 CWT + dCWT, :15-47 phase, :116-222
reassignment, scales ascending); tests/test_oracle.py checks it against the dense oracle on a small case.
Kept per config (a few hundred kB): block sums of the column sums (invariant under bin flips), row energies, the
bin histogram, the norm, and 64 full columns of Tx.
"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ssq_oracle as o  # noqa: E402

CONFIGS = {"c4": dict(log2n=20, top=19), "c5": dict(log2n=22, top=21), "tiny": dict(log2n=12, top=11)}
N_BLOCKS = 4096
N_COLS = 64


def config_inputs(name):
    c = CONFIGS[name]
    N = 1 << c["log2n"]
    x = o.synth_signal(N, 0, np.float64)
    scales = 2.0 ** np.linspace(1, c["top"], 256)           # ssq_cwt.rs:304-322 with the count forced to 256
    return x, scales


def streamed_ssq_cwt(x, scales, wavelet="morlet", dt=1.0, padtype="reflect", flipud=True, gamma=None):
    """oracle.ssq_cwt with one scale in memory at a time; returns (Tx, ssq_freqs, k_hist, n_dropped)."""
    x = np.asarray(x, dtype=np.float64)
    N = x.shape[0]
    na = scales.shape[0]
    P = o.next_power_of_2(N + N // 2)
    padded = o.cwt_pad(x, P, padtype)
    xh = np.fft.fft(padded.astype(np.complex128))
    n1 = (P - N) // 2
    xi = o.xifn(1.0, P)
    norm = 1.0 / float(P)
    g = o.DEFAULT_GAMMA if gamma is None else float(gamma)
    fmin, fmax = 1.0 / float(scales[-1]), 1.0 / float(scales[0])       # maprange "peak"
    freqs = o.cwt_ssq_freqs(na, fmin, fmax, "log")
    Tx = np.zeros((na, N), dtype=np.complex128)
    cols = np.arange(N)
    hist = np.zeros(na, dtype=np.int64)
    dropped = 0
    ixi = 1j * (xi / dt)
    for i in range(na):
        psih = o.wavelet_fourier(xi, float(scales[i]), wavelet)
        r = np.fft.ifft(xh * psih, norm="forward")
        W = ((r.real * norm) + 1j * (r.imag * norm))[n1:n1 + N]
        r = np.fft.ifft(xh * ((psih + 0j) * ixi), norm="forward")
        dW = ((r.real * norm) + 1j * (r.imag * norm))[n1:n1 + N]
        w = o.phase_cwt(W, dW, g)
        b, valid, is_log = o.cwt_bins(w, freqs)
        kk = (na - 1 - b) if flipud else b
        m = valid
        dropped += int((~np.isinf(w) & ~np.isnan(w) & ~valid).sum())
        if m.any():
            Tx[kk[m], cols[m]] += W[m]
            hist += np.bincount(kk[m], minlength=na)
        if i % 16 == 0:
            print(f"  scale {i}/{na}", flush=True)
    return Tx, freqs, hist, dropped, is_log


def summarize(Tx, freqs, hist, dropped, is_log, scales):
    na, N = Tx.shape
    col = Tx.sum(0)
    nb = min(N_BLOCKS, N)
    block = col.reshape(nb, N // nb).sum(1)
    idx = (np.arange(N_COLS) * (N // N_COLS) + (N // (2 * N_COLS))).astype(np.int64)
    return dict(block_col_sums=block, row_energy=np.abs(Tx).sum(1), k_hist=hist, n_dropped=np.array(dropped),
                norm2=np.array(math.sqrt(float((Tx.real ** 2 + Tx.imag ** 2).sum()))),
                col_index=idx, cols=np.ascontiguousarray(Tx[:, idx]), ssq_freqs=freqs,
                absmax=np.array(np.abs(Tx).max()), is_log=np.array(is_log), scales=scales)


def main():
    name = sys.argv[1]
    x, scales = config_inputs(name)
    out = summarize(*streamed_ssq_cwt(x, scales), scales)
    path = os.path.join(HERE, f"{name}_summary.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
