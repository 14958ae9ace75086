"""BASELINE configs 4 and 5 as the full `_rs.ssq_cwt` (CWT + dCWT + phase transform + reassignment) on the GPU:
  C4: Morlet, 256 log scales 2**linspace(1, 19, 256), 1 x 2^20 samples, fp32   (ssq_cwt.rs:116-222, :329-435)
  C5: Morlet, 256 scales 2**linspace(1, 21, 256), ONE signal of the 64 x 2^22 batch, fp64
against the committed oracle summaries (tests/golden/make_golden_large.py: the oracle streamed scale by scale) and
the size-independent invariants of the path: column sums (unchanged by bin flips), row energies, bin histogram,
sampled full columns, run-to-run determinism.  Nothing here recomputes the oracle or reads /root/reference.
"""
import os

import numpy as np
import pytest

from ssqueeze_rs_amd import _rs
from ssqueeze_rs_amd.synth import synth_signal

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _check_against_summary(Tx, f, g, f32):
    na, N = Tx.shape
    assert np.array_equal(f, g["ssq_freqs"])
    assert not bool(g["is_log"])                      # 256 log scales: ratio 1.05 -> the linear bin formula (SURVEY a-18)
    amax = float(g["absmax"])
    # (1) block sums of the column sums: invariant under bin flips
    nb = g["block_col_sums"].shape[0]
    col = Tx.sum(0, dtype=np.complex128)
    blk = col.reshape(nb, N // nb).sum(1)
    e_blk = np.abs(blk - g["block_col_sums"]).max() / np.abs(g["block_col_sums"]).max()
    # (2) row energies
    row = np.abs(Tx).sum(1, dtype=np.float64)
    e_row = np.abs(row - g["row_energy"]).max() / g["row_energy"].max()
    # (3) norm
    nrm = np.sqrt(float((Tx.real.astype(np.float64) ** 2 + Tx.imag.astype(np.float64) ** 2).sum()))
    e_nrm = abs(nrm - float(g["norm2"])) / float(g["norm2"])
    # (4) sampled full columns: elementwise, except where a bin decision flips to the neighbouring row
    cols = Tx[:, g["col_index"]].astype(np.complex128)
    d = np.abs(cols - g["cols"])
    frac_off = float((d > (2e-5 if f32 else 1e-9) * amax).mean())
    e_colsum = np.abs(cols.sum(0) - g["cols"].sum(0)).max() / amax
    print(f"blk {e_blk:.3e} row {e_row:.3e} norm {e_nrm:.3e} cols_off {frac_off:.3e} colsum {e_colsum:.3e}")
    assert e_blk <= (2e-4 if f32 else 1e-9)
    assert e_row <= (1e-3 if f32 else 1e-6)
    assert e_nrm <= (1e-4 if f32 else 1e-7)
    assert frac_off <= (5e-3 if f32 else 1e-4)
    assert e_colsum <= (1e-4 if f32 else 1e-9)


def test_c4_ssq_cwt_full_f32():
    g = np.load(os.path.join(G, "c4_summary.npz"), allow_pickle=False)
    N = 1 << 20
    x = synth_signal(N, 0, np.float32)
    Tx, f = _rs.ssq_cwt(x, wavelet="morlet", scales=g["scales"])
    assert Tx.shape == (256, N) and Tx.dtype == np.complex64
    _check_against_summary(Tx, f, g, True)
    Tx2, _ = _rs.ssq_cwt(x, wavelet="morlet", scales=g["scales"])
    assert np.array_equal(Tx, Tx2)                    # deterministic (no float atomics)


def test_c4_ssq_cwt_full_f64_matches_oracle_tightly():
    """The reference's own arithmetic at C4's size (the fp32 run above is the extension)."""
    g = np.load(os.path.join(G, "c4_summary.npz"), allow_pickle=False)
    N = 1 << 20
    x = synth_signal(N, 0, np.float64)
    Tx, f = _rs.ssq_cwt(x, wavelet="morlet", scales=g["scales"])
    assert Tx.dtype == np.complex128
    _check_against_summary(Tx, f, g, False)


@pytest.mark.skipif(not os.path.exists(os.path.join(G, "c5_summary.npz")), reason="c5_summary.npz not generated")
def test_c5_ssq_cwt_one_signal_f64():
    g = np.load(os.path.join(G, "c5_summary.npz"), allow_pickle=False)
    N = 1 << 22
    x = synth_signal(N, 0, np.float64)
    Tx, f = _rs.ssq_cwt(x, wavelet="morlet", scales=g["scales"])
    assert Tx.shape == (256, N) and Tx.dtype == np.complex128
    _check_against_summary(Tx, f, g, False)
