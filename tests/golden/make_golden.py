"""Generate the committed golden vectors under tests/golden/ from the NumPy oracle
(oracle/ssq_oracle.py, a restatement of the reference's Rust algorithm -- the reference itself
cannot be built or imported here, SURVEY.md §8c, so these pin the ORACLE, not the reference:
"parity unpinned").  Inputs follow the reference's own smoke scripts where they exist.

    python tests/golden/make_golden.py        # rewrites the .npz files (deterministic)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ssq_oracle as o  # noqa: E402


def ref_smoke_signal():
    """tests/stft_test.py:137-141 / tests/cwt_test.py:19-22: 1 s @ 1 kHz, 100 Hz sine."""
    t = np.linspace(0, 1, 1000, endpoint=False)
    return np.sin(2 * np.pi * 100 * t)


def main():
    out = {}
    x = ref_smoke_signal()
    win = np.hanning(256)
    # (i) BASELINE config 1: stft + ssq_stft on the reference's smoke input
    Sx, freqs = o.stft(x, 256, 64, win, "reflect")
    Tx, sf, im = o.ssq_stft(x, win, n_fft=256, hop_len=64, fs=1000, return_intermediates=True)
    np.savez_compressed(os.path.join(HERE, "c1_stft_ssq_stft.npz"), x=x, window=win, Sx=Sx, freqs=freqs,
                        Tx=Tx, ssq_freqs=sf, k=im["k"].astype(np.int32), w=im["w"], dSx=im["dSx"])
    # (ii) the reference's CWT smoke: N=1000, scales=logspace(1,5,32)/fs, GMW and Morlet
    scales = np.logspace(1, 5, 32) / 1000
    d = dict(x=x, scales=scales)
    for wv in ("gmw", "morlet"):
        Wx, _, dWx = o.cwt(x, wv, scales=scales, fs=1000, nv=16, derivative=True)
        T, f, imc = o.ssq_cwt(x, wv, scales=scales, fs=1000, nv=16, return_intermediates=True)
        # every 4th time column only (fixture size); tests compare on the same columns
        d.update({f"Wx_{wv}": Wx[:, ::4], f"dWx_{wv}": dWx[:, ::4], f"Tx_{wv}": T[:, ::4], f"ssq_freqs_{wv}": f,
                  f"k_{wv}": np.where(imc["valid"], imc["k"], -1).astype(np.int8)[:, ::4]})
    np.savez_compressed(os.path.join(HERE, "smoke_cwt_ssq_cwt.npz"), **d)
    # (iii) N=4096 multi-sine+chirp, n_fft=256/hop=64: sum/lebesgue x reflect/zero
    xs = o.synth_signal(4096, 0)
    d = dict(x=xs, window=win)
    for sq in ("sum", "lebesgue"):
        for pad in ("reflect", "zero"):
            T, f, imm = o.ssq_stft(xs, win, n_fft=256, hop_len=64, fs=2.0, padtype=pad, squeezing=sq,
                                   return_intermediates=True)
            d[f"Tx_{sq}_{pad}"] = T
            d[f"k_{sq}_{pad}"] = np.where(np.isinf(imm["w"]), -1, imm["k"]).astype(np.int16)
    d["ssq_freqs"] = f
    np.savez_compressed(os.path.join(HERE, "stft4096_modes.npz"), **d)
    # (iv) N=2048, 32 log scales over [2, N/2] (is_log False quirk), option grid
    xc = o.synth_signal(2048, 1)
    d = dict(x=xc)
    for name, kw in (("default", {}), ("noflip", dict(flipud=False)), ("maximal", dict(maprange="maximal")),
                     ("linear", dict(ssq_freqs="linear")), ("lebesgue", dict(squeezing="lebesgue"))):
        T, f, imc = o.ssq_cwt(xc, "morlet", nv=4, return_intermediates=True, **kw)
        d[f"Tx_{name}"] = T[:, ::16]             # every 16th time column
        d[f"f_{name}"] = f
        d[f"is_log_{name}"] = np.array(imc["is_log"])
    d["scales"] = imc["scales"]
    np.savez_compressed(os.path.join(HERE, "cwt2048_options.npz"), **d)
    # (v) full-size summary statistics for BASELINE config 2 (1 x 2^20, n_fft=1024, hop=256): seeds + checksums
    x2 = o.synth_signal(1 << 20, 0)
    T2, f2, im2 = o.ssq_stft(x2, np.hanning(1024), n_fft=1024, hop_len=256, fs=1.0, return_intermediates=True)
    np.savez_compressed(os.path.join(HERE, "c2_summary.npz"), col_sums=T2.sum(0), row_energy=np.abs(T2).sum(1),
                        k_hist=np.bincount(im2["k"][~np.isinf(im2["w"])].ravel(), minlength=513),
                        norm2=np.array(np.linalg.norm(T2)), sx_absmax=np.array(np.abs(im2["Sx"]).max()))
    for fn in sorted(os.listdir(HERE)):
        if fn.endswith(".npz"):
            print(fn, os.path.getsize(os.path.join(HERE, fn)) // 1024, "KiB")


def make_c2_k_cols():
    """(vi) oracle bins of BASELINE config 2 on every 16th frame (256 columns), for bench.py's `validation`: the measured
    end-to-end fp32 bin mismatch rate (SURVEY 8(c) asks to report it).  Input = the fp32 workload signal, seed 0."""
    x32 = o.synth_signal(1 << 20, 0, np.float32)
    T, f, im = o.ssq_stft(x32.astype(np.float64), np.hanning(1024), n_fft=1024, hop_len=256, fs=1.0,
                          return_intermediates=True)
    cols = np.arange(0, 4096, 16)
    keep = ~np.isinf(im["w"][:, cols])
    strong = keep & (np.abs(im["Sx"][:, cols]) > 1e-3 * np.abs(im["Sx"]).max())
    np.savez_compressed(os.path.join(HERE, "c2_k_cols.npz"), col_index=cols,
                        k=np.where(keep, im["k"][:, cols], -1).astype(np.int16), strong=np.packbits(strong, axis=None),
                        shape=np.array(strong.shape))
    print("c2_k_cols.npz", os.path.getsize(os.path.join(HERE, "c2_k_cols.npz")) // 1024, "KiB")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "c2k":
        make_c2_k_cols()
    else:
        main()
