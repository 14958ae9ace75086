"""GPU parity tests of the code path bench.py times: batch > 1 through `ssq_stft_plan_exec`, large enough that a
pass is the interior launch + the edge launch of the 16-wave kernel (`stft_tx1024_kernel<false,..>` +
`<true,..>`, persistent blocks walking tiles across signal boundaries -- csrc/stft_fused.hip::launch_one).

What is pinned here:
  * every signal of the batch is BITWISE equal to the same signal through the single-signal call (one launch of
    the edge-capable kernel) -- the fixed-point tile makes the result independent of tile order and launch split;
  * signal 0 against the fp64 oracle with the bins of the Tx kernel ITSELF: `SSQ_OUT_WK` is served by the
    `WKDBG` instantiation of the kernel that serves `SSQ_OUT_TX`, so `k == stft_bins_f32_model(w)` index-exactly on
    the hot arithmetic and `|Tx - reaccumulate(Sx, k)| <= 2e-5 max|Tx|` strictly (no "isolated swaps" allowance);
  * BASELINE config 3's per-GPU share (32 x 2^20) against the committed C2 checksums.
Reference: the channel loop tests/stft_ssq_test.py:230-248 (one call per channel) and ssq_stft.rs:276-301.
"""
import os

import numpy as np
import pytest

from oracle import ssq_oracle as o
from tests.helpers.binrule import end_to_end_rate, stft_bins_follow_reference_rule
from ssqueeze_rs_amd import _lib, _rs
from ssqueeze_rs_amd.batch import SsqStftBatch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _batch(N, B, dtype, first=0):
    return np.stack([o.synth_signal(N, first + b, dtype) for b in range(B)])


def _wk(arr):
    return arr.real.copy(), np.rint(arr.imag).astype(np.int64)


def _check_sig_vs_oracle_f32(x32, win, n_fft, hop, Tx, Sx, w, k, squeezing="sum"):
    """Tx, Sx, (w, k) of ONE signal from the batch launch; (w, k) are the Tx kernel's own."""
    Tx_o, f_o, im = o.ssq_stft(x32.astype(np.float64), win, n_fft=n_fft, hop_len=hop, fs=1.0,
                               squeezing=squeezing, return_intermediates=True)
    smax = np.abs(im["Sx"]).max()
    assert np.abs(Sx - im["Sx"]).max() <= 2e-6 * smax
    keep = k >= 0
    # index-exact on the hot kernel's own w: under the reference's scan (ssq_stft.rs:280-289) outside the 2^-22 tie
    # window, and equal to the documented fp32 formula everywhere
    stft_bins_follow_reference_rule(k, w, f_o, keep)
    assert np.array_equal(k[keep], o.stft_bins_f32_model(w[keep], im["dw"], Tx.shape[0]))
    # strict scatter check: re-accumulate from the kernel's Sx and the Tx kernel's own k
    Tx_re = o.accumulate_tx(Sx.astype(np.complex128), np.where(keep, k, 0), keep, float(np.float32(im["dw"])),
                            Tx.shape[0], lebesgue=(squeezing == "lebesgue"))
    tmax = np.abs(Tx_re).max()
    assert np.abs(Tx - Tx_re).max() <= 2e-5 * tmax
    # end to end against the fp64 oracle: bins move at most to a neighbour, rarely
    keep_o = ~np.isinf(im["w"])
    both = keep_o & keep & (np.abs(im["Sx"]) > 1e-3 * smax)
    rate = end_to_end_rate(k, im["k"], both)               # measured 2.4e-5 at 2^18 (profiles/r03_bin_parity.json)
    assert rate <= max(5e-5, 2.0 / max(1, int(both.sum()))), f"end-to-end fp32 bin mismatch rate {rate:.2e}"
    assert np.abs(k[both] - im["k"][both]).max() <= 1
    assert np.abs(Tx.astype(np.complex128).sum(0) - Tx_o.sum(0)).max() <= 1e-4 * smax * im["dw"]


def test_batch32_two_launch_path_bitwise_and_oracle(monkeypatch):
    """32 x 2^18 fp32, n_fft 1024: 2048 tiles > 4 x 256 blocks -> interior + edge launch, blocks cross signals."""
    monkeypatch.delenv("SSQ_SINGLE_LAUNCH", raising=False)
    N, B, n_fft, hop = 1 << 18, 32, 1024, 256
    assert ((N - 1) // hop + 1 + 15) // 16 * B > 4 * 256
    x = _batch(N, B, np.float32)
    win = np.hanning(n_fft)
    eng = SsqStftBatch(N, win, n_fft, hop, fs=1.0, dtype=np.float32, max_batch=B)
    try:
        Tx = eng.run(x, _lib.OUT_TX)
        for b in range(B):
            assert np.array_equal(Tx[b], eng.run(x[b:b + 1], _lib.OUT_TX)[0]), f"signal {b}"
        one, _ = _rs.ssq_stft(x[5], win, n_fft=n_fft, hop_len=hop, fs=1.0)       # the drop-in call, same bits
        assert np.array_equal(Tx[5], one)
        assert np.array_equal(Tx, eng.run(x, _lib.OUT_TX))                         # run-to-run deterministic
        WK = eng.run(x, _lib.OUT_WK)                                               # two-launch path, Tx kernel's own bins
        SX = eng.run(x[:2], _lib.OUT_SX)
        for b in (0, 1):
            w, k = _wk(WK[b])
            _check_sig_vs_oracle_f32(x[b], win, n_fft, hop, Tx[b], SX[b], w, k)
        # the hook itself is launch-split invariant too
        assert np.array_equal(WK[7], eng.run(x[7:8], _lib.OUT_WK)[0])
    finally:
        eng.close()


@pytest.mark.parametrize("dtype,n_fft,hop,N,B", [(np.float32, 1024, 256, 70000, 3), (np.float32, 256, 64, 50000, 4),
                                                 (np.float64, 512, 128, 60000, 3), (np.float64, 1024, 256, 40000, 2)])
def test_forced_two_launch_with_batch_gt1(monkeypatch, dtype, n_fft, hop, N, B):
    """SSQ_SINGLE_LAUNCH=0 forces interior + edge launches on small jobs: every kernel family, batch > 1,
    ragged last tile (N not a multiple of the tile span)."""
    x = _batch(N, B, dtype, first=40)
    win = np.hanning(n_fft)
    eng = SsqStftBatch(N, win, n_fft, hop, fs=1.0, dtype=dtype, max_batch=B)
    try:
        outs = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("SSQ_SINGLE_LAUNCH", mode)
            outs[mode] = (eng.run(x, _lib.OUT_TX), eng.run(x, _lib.OUT_WK))
        assert np.array_equal(outs["1"][0], outs["0"][0])
        assert np.array_equal(outs["1"][1], outs["0"][1])
        monkeypatch.setenv("SSQ_SINGLE_LAUNCH", "0")
        for b in range(B):
            assert np.array_equal(outs["0"][0][b], eng.run(x[b:b + 1], _lib.OUT_TX)[0])
    finally:
        eng.close()
    if dtype == np.float32:
        monkeypatch.setenv("SSQ_SINGLE_LAUNCH", "0")
        eng = SsqStftBatch(N, win, n_fft, hop, fs=1.0, dtype=dtype, max_batch=B)
        try:
            w, k = _wk(outs["0"][1][1])
            _check_sig_vs_oracle_f32(x[1], win, n_fft, hop, outs["0"][0][1], eng.run(x[1:2], _lib.OUT_SX)[0], w, k)
        finally:
            eng.close()


def test_lebesgue_two_launch_hot_bins(monkeypatch):
    monkeypatch.setenv("SSQ_SINGLE_LAUNCH", "0")
    N, B, n_fft, hop = 60000, 2, 1024, 256
    x = _batch(N, B, np.float32, first=50)
    win = np.hanning(n_fft)
    eng = SsqStftBatch(N, win, n_fft, hop, fs=1.0, squeezing="lebesgue", dtype=np.float32, max_batch=B)
    try:
        Tx, WK, SX = eng.run(x, _lib.OUT_TX), eng.run(x, _lib.OUT_WK), eng.run(x, _lib.OUT_SX)
        w, k = _wk(WK[1])
        _check_sig_vs_oracle_f32(x[1], win, n_fft, hop, Tx[1], SX[1], w, k, squeezing="lebesgue")
    finally:
        eng.close()


def test_c3_per_gpu_share_32x2pow20_vs_c2_checksums(monkeypatch):
    """BASELINE config 3's share of one GPU: 32 x 2^20 fp32 (538 MB of Tx) through the two-launch path; signal 0 is
    the C2 signal -> committed checksums; every signal bitwise equal to the single-signal path."""
    monkeypatch.delenv("SSQ_SINGLE_LAUNCH", raising=False)
    N, B, n_fft, hop = 1 << 20, 32, 1024, 256
    x = _batch(N, B, np.float32)
    win = np.hanning(n_fft)
    g = np.load(os.path.join(G, "c2_summary.npz"), allow_pickle=False)
    eng = SsqStftBatch(N, win, n_fft, hop, fs=1.0, dtype=np.float32, max_batch=B)
    try:
        Tx = eng.run(x, _lib.OUT_TX)
        assert Tx.shape == (B, 513, 4096)
        for b in range(B):
            assert np.array_equal(Tx[b], eng.run(x[b:b + 1], _lib.OUT_TX)[0]), f"signal {b}"
        dw = 0.5 / 512
        scale = float(g["sx_absmax"]) * dw
        assert np.abs(Tx[0].astype(np.complex128).sum(0) - g["col_sums"]).max() <= 1e-4 * scale
        e = np.abs(Tx[0]).sum(1)
        assert np.abs(e - g["row_energy"]).max() <= 2e-2 * g["row_energy"].max()
        assert abs(np.linalg.norm(Tx[0].astype(np.complex128)) - float(g["norm2"])) <= 1e-2 * float(g["norm2"])
        # bins of the Tx kernel itself on the C2 signal: histogram against the oracle's, index-exact against the model
        WK = eng.run(x[:1], _lib.OUT_WK)            # (single-signal path; the two-launch hook is covered above)
        w, k = _wk(WK[0])
        keep = k >= 0
        assert np.array_equal(k[keep], o.stft_bins_f32_model(w[keep], dw, 513))
        stft_bins_follow_reference_rule(k, w, o.stft_ssq_freqs(513, 1.0), keep)     # ssq_stft.rs:280-289 on own w
        hist = np.bincount(k[keep].ravel(), minlength=513)
        # measured 1.1e-3 of the bins (all kept bins, weak ones included; profiles/r03_bin_parity.json: 5.4e-4 flip)
        assert np.abs(hist - g["k_hist"]).sum() <= 4e-3 * g["k_hist"].sum()
        # size-independent invariant on every signal: column sums = dw * sum of the kept Sx (needs Sx: 4 signals)
        SX = eng.run(x[:4], _lib.OUT_SX)
        WK4 = eng.run(x[:4], _lib.OUT_WK)
        for b in range(4):
            kb = np.rint(WK4[b].imag) >= 0
            rhs = dw * np.where(kb, SX[b].astype(np.complex128), 0).sum(0)
            assert np.abs(Tx[b].astype(np.complex128).sum(0) - rhs).max() <= 1e-4 * np.abs(SX[b]).max() * dw
    finally:
        eng.close()
