"""GPU parity tests of the STFT family: HIP kernels (through the C-ABI / ctypes mirror) against
the NumPy oracle on the same seeded inputs.

Tolerances (stated per SURVEY.md §8c):
  fp64:  |dSx|/max|Sx| <= 1e-11; bin index k exact except within 1e-6 of a half-bin tie;
         Tx compared after re-accumulating the oracle with the kernel's own k (<= 1e-10).
  fp32:  |dSx|/max|Sx| <= 3e-6 (n_fft <= 1024); k == fp32 bin model of the kernel's own w
         (index-exact); Tx vs re-accumulation from the kernel's own (Sx, k) <= 2e-5;
         end-to-end k mismatch rate against the fp64 oracle only reported and loosely bounded.
"""
import numpy as np
import pytest

from oracle import ssq_oracle as o
from tests.helpers.binrule import end_to_end_rate, stft_bins_follow_reference_rule
from ssqueeze_rs_amd import _rs

pytestmark = pytest.mark.gpu


def _relerr(a, b):
    s = max(np.abs(b).max(), 1e-300)
    return np.abs(a - b).max() / s


def _sig(N, seed=0, dtype=np.float64):
    return o.synth_signal(N, seed, dtype)


# ------------------------------------------------------------------ stft ----
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 2e-6)])
def test_stft_c1_plumbing(dtype, tol):
    """BASELINE config 1: 1 s @ 1 kHz, 100 Hz sine, n_fft=256, hop=64, Hann (tests/stft_test.py:137-151)."""
    t = np.linspace(0, 1, 1000, endpoint=False)
    x = np.sin(2 * np.pi * 100 * t).astype(dtype)
    win = np.hanning(256)
    Sx, freqs = _rs.stft(x, 256, 64, win, "reflect")
    Sx_o, freqs_o = o.stft(x.astype(np.float64), 256, 64, win, "reflect")
    assert Sx.shape == (129, 16) and freqs.shape == (129,)
    assert Sx.dtype == (np.complex128 if dtype == np.float64 else np.complex64)
    assert np.array_equal(freqs, freqs_o)
    assert _relerr(Sx, Sx_o) <= tol
    assert np.abs(Sx).sum(1).argmax() == 26


@pytest.mark.parametrize("n_fft,hop", [(64, 16), (128, 32), (256, 64), (512, 100), (1024, 256),
                                       (2048, 512), (4096, 1024)])
@pytest.mark.parametrize("dtype,tol", [(np.float64, 2e-12), (np.float32, 4e-6)])
def test_stft_fused_sizes(n_fft, hop, dtype, tol):
    x = _sig(3 * n_fft + 777, 1, dtype)
    win = np.hanning(n_fft) + 0.1
    for pad in ("reflect", "zero"):
        Sx, _ = _rs.stft(x, n_fft, hop, win, pad)
        Sx_o, _ = o.stft(x.astype(np.float64), n_fft, hop, win, pad)
        assert Sx.shape == Sx_o.shape
        assert _relerr(Sx, Sx_o) <= tol, (n_fft, pad)


@pytest.mark.parametrize("n_fft,hop", [(100, 25), (1000, 250), (48, 7), (17, 5), (1, 1), (2, 1), (6000, 3000)])
def test_stft_generic_any_length(n_fft, hop):
    """rustfft plans any length; lengths the fused kernel does not cover run the generic kernels."""
    x = _sig(2 * n_fft + 301, 2)
    win = np.hanning(n_fft) + 0.05 if n_fft > 2 else np.ones(n_fft)
    Sx, freqs = _rs.stft(x, n_fft, hop, win, "reflect")
    Sx_o, freqs_o = o.stft(x, n_fft, hop, win, "reflect")
    assert Sx.shape == Sx_o.shape
    assert np.array_equal(freqs, freqs_o)
    assert _relerr(Sx, Sx_o) <= 1e-11


def test_stft_short_signal_and_batch():
    win = np.hanning(256)
    x = _sig(100, 3)                       # shorter than n_fft: reflect leaves zeros beyond the mirror
    Sx, _ = _rs.stft(x, 256, 64, win, "reflect")
    Sx_o, _ = o.stft(x, 256, 64, win, "reflect")
    assert _relerr(Sx, Sx_o) <= 1e-12
    xb = np.stack([_sig(5000, b) for b in range(5)])
    Sb, _ = _rs.stft(xb, 256, 64, win, "reflect")
    assert Sb.shape == (5, 129, (5000 - 1) // 64 + 1)
    for b in range(5):
        Sx_o, _ = o.stft(xb[b], 256, 64, win, "reflect")
        assert _relerr(Sb[b], Sx_o) <= 1e-12


# -------------------------------------------------------------- ssq_stft ----
def _check_ssq_f64(x, win, n_fft, hop, fs, pad, squeezing, gamma=None, win_len=None):
    Tx, f, dbg = _rs.ssq_stft(x, win, n_fft=n_fft, win_len=win_len, hop_len=hop, fs=fs, padtype=pad,
                              squeezing=squeezing, gamma=gamma, _debug=True)
    Tx_o, f_o, im = o.ssq_stft(x, win, n_fft=n_fft, win_len=win_len, hop_len=hop, fs=fs, padtype=pad,
                               squeezing=squeezing, gamma=gamma, return_intermediates=True)
    assert Tx.shape == Tx_o.shape and Tx.dtype == np.complex128
    assert np.array_equal(f, f_o)
    smax = np.abs(im["Sx"]).max()
    assert np.abs(dbg["Sx"] - im["Sx"]).max() <= 1e-11 * smax
    assert np.abs(dbg["dSx"] - im["dSx"]).max() <= 1e-11 * max(np.abs(im["dSx"]).max(), 1e-300)
    keep_o = ~np.isinf(im["w"])
    keep_g = dbg["k"] >= 0
    # bins whose |Sx| sits at the gamma threshold may flip keep/skip; they carry ~gamma of energy
    flip = keep_o != keep_g
    assert np.abs(im["Sx"][flip]).max(initial=0.0) <= 1e-6 * smax + 1e-12
    both = keep_o & keep_g
    mism = both & (dbg["k"] != im["k"])
    # a mismatch is only legitimate next to a half-bin tie, or where w itself is ill-conditioned
    # (|Sx| tiny => w amplifies rounding); everything else must be index-exact
    if mism.any():
        tq = im["w"][mism] / im["dw"]
        near_tie = np.abs(tq - np.floor(tq) - 0.5) < 1e-9 * np.maximum(1.0, np.abs(tq))
        tiny = np.abs(im["Sx"][mism]) <= 1e-6 * smax
        assert (near_tie | tiny).all(), f"{mism.sum()} unexplained bin mismatches"
    assert mism.mean() <= 1e-3
    # scatter itself: re-accumulate with the kernel's own k and keep
    Tx_re = o.accumulate_tx(im["Sx"], np.where(keep_g, dbg["k"], 0), keep_g, im["dw"], Tx.shape[0],
                            lebesgue=(squeezing == "lebesgue"))
    assert np.abs(Tx - Tx_re).max() <= 1e-10 * max(np.abs(Tx_re).max(), 1e-300)
    return mism.sum(), both.sum()


@pytest.mark.parametrize("n_fft,hop", [(256, 64), (1024, 256), (64, 16), (2048, 512)])
@pytest.mark.parametrize("pad", ["reflect", "zero"])
def test_ssq_stft_f64_fused(n_fft, hop, pad):
    x = _sig(4096 + 3 * n_fft, 4)
    _check_ssq_f64(x, np.hanning(n_fft), n_fft, hop, 1.0, pad, "sum")


def test_ssq_stft_f64_options():
    x = _sig(4096, 5)
    _check_ssq_f64(x, np.hanning(256), 256, 64, 1000.0, "reflect", "lebesgue")
    _check_ssq_f64(x, np.hanning(200), 256, 64, 24414.0625, "reflect", "sum")       # window shorter than n_fft
    _check_ssq_f64(x, np.hanning(300), 256, 32, 2.0, "reflect", "sum", win_len=256)  # longer: centre-cropped
    _check_ssq_f64(x, np.hanning(256), 256, 64, 1.0, "reflect", "sum", gamma=1e-3)  # gamma -> inf rows
    _check_ssq_f64(x, np.hanning(100), 100, 25, 1.0, "reflect", "sum")              # generic path


def test_ssq_stft_c1_reference_smoke():
    """tests/stft_ssq_test.py:130-152 inputs: shape (129, 16), energy concentrates in row 25."""
    t = np.linspace(0, 1, 1000, endpoint=False)
    x = np.sin(2 * np.pi * 100 * t)
    Tx, f = _rs.ssq_stft(x, window=np.hanning(256), n_fft=256, hop_len=64, fs=1000,
                         padtype="reflect", squeezing="sum")
    assert Tx.shape == (129, 16) and f.shape == (129,)
    assert np.abs(Tx).sum(1).argmax() == 25
    _check_ssq_f64(x, np.hanning(256), 256, 64, 1000.0, "reflect", "sum")


def _check_ssq_f32(x32, win, n_fft, hop, fs, pad="reflect", squeezing="sum"):
    Tx, f, dbg = _rs.ssq_stft(x32, win, n_fft=n_fft, hop_len=hop, fs=fs, padtype=pad,
                              squeezing=squeezing, _debug=True)
    Tx_o, f_o, im = o.ssq_stft(x32.astype(np.float64), win, n_fft=n_fft, hop_len=hop, fs=fs, padtype=pad,
                               squeezing=squeezing, return_intermediates=True)
    assert Tx.dtype == np.complex64 and Tx.shape == Tx_o.shape
    assert np.array_equal(f, f_o)
    smax = np.abs(im["Sx"]).max()
    assert np.abs(dbg["Sx"] - im["Sx"]).max() <= 2e-6 * smax             # SURVEY 8(c); measured 1.0-1.5e-7
    assert np.abs(dbg["dSx"] - im["dSx"]).max() <= 2e-6 * np.abs(im["dSx"]).max()
    keep_g = dbg["k"] >= 0
    # (a) w: the kernel's own Sx/dSx through the reference formula
    Sg, dSg = dbg["Sx"].astype(np.complex128), dbg["dSx"].astype(np.complex128)
    w_m = o.phase_stft(Sg, dSg, im["Sfs"], o.DEFAULT_GAMMA)
    strong = keep_g & (np.abs(Sg) > 1e-3 * smax) & np.isfinite(w_m)
    assert np.abs(dbg["w"][strong] - w_m[strong]).max() <= 1e-6 * (0.5 * fs)   # measured 3e-8 .. 1.3e-7
    # (b) index-exact under the REFERENCE's rule (ssq_stft.rs:280-289, first-minimum scan) given the kernel's own w,
    #     outside a 2^-22-relative window around half-bin ties (tests/helpers/binrule.py; measured: 0 differences)
    stft_bins_follow_reference_rule(dbg["k"], dbg["w"], f, keep_g)
    # (b') and equal to the documented fp32 bin formula everywhere (ties included)
    k_model = o.stft_bins_f32_model(dbg["w"][keep_g], im["dw"], Tx.shape[0])
    assert np.array_equal(dbg["k"][keep_g], k_model)
    # (c) the scatter: re-accumulate from the kernel's own Sx and k
    Tx_re = o.accumulate_tx(Sg, np.where(keep_g, dbg["k"], 0), keep_g, float(np.float32(im["dw"])),
                            Tx.shape[0], lebesgue=(squeezing == "lebesgue"))
    # (w, k) come from the WKDBG instantiation of the very kernel that serves Tx (csrc/stft_fused.hip), so the
    # check is strict: every element, no allowance for neighbour swaps
    tmax = np.abs(Tx_re).max()
    assert np.abs(Tx - Tx_re).max() <= 2e-5 * tmax
    # (d) end-to-end against the fp64 oracle: per-column energy moves at most between neighbours
    keep_o = ~np.isinf(im["w"])
    both = keep_o & keep_g & (np.abs(im["Sx"]) > 1e-3 * smax)
    # measured (profiles/r03_bin_parity.json): 0 at these sizes, 2.4e-5 at 2^18, 1.3e-5 at C2 -- bound = 2x the worst,
    # or two bins where the case is too small for the rate to resolve
    rate = end_to_end_rate(dbg["k"], im["k"], both)
    assert rate <= max(5e-5, 2.0 / max(1, int(both.sum()))), f"fp32 end-to-end bin mismatch rate {rate:.2e}"
    assert np.abs(dbg["k"][both] - im["k"][both]).max(initial=0) <= 1
    return rate


@pytest.mark.parametrize("n_fft,hop", [(256, 64), (1024, 256), (128, 32), (4096, 1024)])
def test_ssq_stft_f32_fused(n_fft, hop):
    x = _sig(8192 + 2 * n_fft, 6, np.float32)
    _check_ssq_f32(x, np.hanning(n_fft), n_fft, hop, 1.0)


def test_ssq_stft_f32_lebesgue_and_fs():
    x = _sig(8192, 7, np.float32)
    _check_ssq_f32(x, np.hanning(512), 512, 128, 24414.0625, "zero", "lebesgue")


def test_ssq_stft_fused_vs_generic_kernels():
    """Two independent GPU implementations (fused LDS-tile kernel vs unfused generic kernels)."""
    import ctypes as C
    from ssqueeze_rs_amd import _lib
    lib = _lib.load()
    N, n_fft, hop = 20000, 512, 128
    x = _sig(N, 8)
    win = np.hanning(n_fft)
    outs = []
    for force_generic in (0, 1):
        plan = C.c_void_p()
        _lib.check(lib.ssq_stft_plan_create(C.byref(plan), _lib.SSQ_F64, N, win.ctypes.data_as(C.c_void_p),
                                            n_fft, hop, 1.0, 0, 0, -1.0, force_generic))
        assert lib.ssq_stft_plan_is_fused(plan) == (0 if force_generic else 1)
        nf, nfr = n_fft // 2 + 1, (N - 1) // hop + 1
        d_x, d_out, d_ws = C.c_void_p(), C.c_void_p(), C.c_void_p()
        ws = lib.ssq_stft_plan_workspace_bytes(plan, 1, _lib.OUT_TX)
        _lib.check(lib.ssq_dev_malloc(C.byref(d_x), N * 8))
        _lib.check(lib.ssq_dev_malloc(C.byref(d_out), nf * nfr * 16))
        _lib.check(lib.ssq_dev_malloc(C.byref(d_ws), max(ws, 16)))
        _lib.check(lib.ssq_memcpy_h2d(d_x, x.ctypes.data_as(C.c_void_p), N * 8, None))
        _lib.check(lib.ssq_stft_plan_exec(plan, _lib.OUT_TX, d_x, 1, d_out, d_ws, ws, None))
        Tx = np.empty((nf, nfr), dtype=np.complex128)
        _lib.check(lib.ssq_device_sync())
        _lib.check(lib.ssq_memcpy_d2h(Tx.ctypes.data_as(C.c_void_p), d_out, Tx.nbytes, None))
        _lib.check(lib.ssq_device_sync())
        for p in (d_x, d_out, d_ws):
            lib.ssq_dev_free(p)
        lib.ssq_stft_plan_destroy(plan)
        outs.append(Tx)
    # same bins up to ties; compare column sums (bin-flip invariant) and values
    assert np.abs(outs[0].sum(0) - outs[1].sum(0)).max() <= 1e-9 * np.abs(outs[1]).max()
    frac = (np.abs(outs[0] - outs[1]) > 1e-9 * np.abs(outs[1]).max()).mean()
    assert frac <= 1e-3


def test_ssq_stft_full_size_properties():
    """BASELINE config 2 (1 x 2^20, n_fft=1024, hop=256, fp32): size-independent properties.
    Column sums are invariant under bin flips:  sum_k Tx[k,j] == dw * sum_{i kept} Sx[i,j]."""
    N = 1 << 20
    x = _sig(N, 0, np.float32)
    win = np.hanning(1024)
    Tx, f, dbg = _rs.ssq_stft(x, win, n_fft=1024, hop_len=256, fs=1.0, _debug=True)
    assert Tx.shape == (513, 4096)
    keep = dbg["k"] >= 0
    dw = f[1] - f[0]
    lhs = Tx.astype(np.complex128).sum(0)
    rhs = dw * np.where(keep, dbg["Sx"].astype(np.complex128), 0).sum(0)
    scale = np.abs(dbg["Sx"]).max() * dw
    assert np.abs(lhs - rhs).max() <= 1e-4 * scale
    k_model = o.stft_bins_f32_model(dbg["w"][keep], dw, 513)
    assert np.array_equal(dbg["k"][keep], k_model)
    # the reference's first-minimum scan on the kernel's own w: index-exact outside the 2^-22 tie window, 2.1 M bins
    stft_bins_follow_reference_rule(dbg["k"], dbg["w"], f, keep)
    # and against the fp64 oracle on the same input
    Tx_o, _, im = o.ssq_stft(x.astype(np.float64), win, n_fft=1024, hop_len=256, fs=1.0,
                             return_intermediates=True)
    assert np.abs(dbg["Sx"] - im["Sx"]).max() <= 2e-6 * np.abs(im["Sx"]).max()
    smax = np.abs(im["Sx"]).max()
    both = ~np.isinf(im["w"]) & keep & (np.abs(im["Sx"]) > 1e-3 * smax)
    rate = end_to_end_rate(dbg["k"], im["k"], both)        # measured 1.3e-5 (profiles/r03_bin_parity.json)
    assert rate <= 5e-5, f"C2 end-to-end fp32 bin mismatch rate {rate:.2e}"
    assert np.abs(dbg["k"][both] - im["k"][both]).max() <= 1
    assert np.abs(Tx.sum(0) - Tx_o.sum(0)).max() <= 1e-4 * scale
    # row-energy profile (where the ridges are) agrees
    e_g, e_o = np.abs(Tx).sum(1), np.abs(Tx_o).sum(1)
    assert np.abs(e_g - e_o).max() <= 2e-2 * e_o.max()


def test_ssq_stft_interior_edge_split_equals_single_launch(monkeypatch):
    """A fused pass is either one launch of the edge-capable kernel over all tiles (small jobs) or an interior launch
    (no padding logic) plus an edge launch (large jobs: csrc/stft_fused.hip::launch_one).  Both must give the same
    bits -- the fixed-point tile is order-exact -- for the 16-wave fp32 kernel and for the generic fused kernel."""
    for dtype, n_fft, hop, N in ((np.float32, 1024, 256, 300000), (np.float64, 512, 128, 100000),
                                 (np.float32, 256, 64, 70000)):
        x = _sig(N, 31, dtype)
        win = np.hanning(n_fft)
        outs = []
        for mode in ("1", "0"):
            monkeypatch.setenv("SSQ_SINGLE_LAUNCH", mode)
            Tx, f = _rs.ssq_stft(x, win, n_fft=n_fft, hop_len=hop)
            outs.append(Tx)
        assert np.array_equal(outs[0], outs[1]), (dtype, n_fft)


# ------------------------------------------------------------- API parity ----
def test_error_behaviour_matches_reference():
    x = _sig(1000, 9)
    with pytest.raises(ValueError):                      # ssq_stft.rs:96-101
        _rs.ssq_stft(x, np.hanning(300), n_fft=256)
    with pytest.raises(_rs.PanicException):              # stft.rs:67 length mismatch panics
        _rs.stft(x, 256, 64, np.hanning(200), "reflect")
    with pytest.raises(TypeError):                       # PyReadonlyArray1<f64> extraction
        _rs.stft(list(x), 256, 64, np.hanning(256), "reflect")
    with pytest.raises(TypeError):
        _rs.stft(x.astype(np.int32), 256, 64, np.hanning(256), "reflect")
    with pytest.raises(_rs.PanicException):
        _rs.ssq_stft(x, np.hanning(256), n_fft=256, hop_len=0)
    # unknown strings silently fall back (ssq_stft.rs:127,:295)
    a, _ = _rs.ssq_stft(x, np.hanning(256), n_fft=256, hop_len=64, padtype="bogus", squeezing="bogus")
    b, _ = _rs.ssq_stft(x, np.hanning(256), n_fft=256, hop_len=64)
    assert np.array_equal(a, b)
    # defaults: n_fft = min(N, 512), hop_len = 1
    Tx, f = _rs.ssq_stft(x[:300], np.hanning(300))
    assert Tx.shape == (151, 300)
    assert _rs.hello_from_bin() == "Hello from ssqueeze!"


# ------------------------- any-length n_fft inside the fused kernel: mixed-radix passes (prime factors <= 13) or Bluestein ----
@pytest.mark.parametrize("n_fft,hop", [(1000, 250), (999, 100), (1001, 333), (100, 25), (24, 6), (33, 8), (600, 150),
                                       (1025, 256), (2047, 512), (1536, 384), (2187, 500), (4095, 1024), (3000, 750),
                                       (77, 20), (1920, 480), (96, 24), (2002, 500), (4000, 1000), (26, 5), (1014, 250),
                                       (968, 242), (130, 32), (2058, 512)])
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-11), (np.float32, 6e-6)])
def test_stft_bluestein_lengths(n_fft, hop, dtype, tol):
    """rustfft plans any length (stft.rs:43-44, ssq_stft.rs:198-199); 24 <= n_fft <= 4096 with prime factors <= 13 run
    mixed-radix passes (fft_mixed.h: every radix 2..16 of its list is met here), other n_fft <= 2048 Bluestein's chirp-z
    through two power-of-two transforms, all inside the fused kernel (no O(n^2) direct sums)."""
    import ctypes as C
    from ssqueeze_rs_amd import _lib
    x = _sig(5 * n_fft + 123, 21, dtype)
    win = np.hanning(n_fft) + 0.07
    lib = _lib.load()
    plan = C.c_void_p()
    _lib.check(lib.ssq_stft_plan_create(C.byref(plan), _lib.SSQ_F32 if dtype == np.float32 else _lib.SSQ_F64, x.shape[0],
                                        win.ctypes.data_as(C.c_void_p), n_fft, hop, 1.0, 0, 0, -1.0, 0))
    assert lib.ssq_stft_plan_is_fused(plan) == 1                       # not the direct-sum kernels
    lib.ssq_stft_plan_destroy(plan)
    for pad in ("reflect", "zero"):
        Sx, f = _rs.stft(x, n_fft, hop, win, pad)
        Sx_o, f_o = o.stft(x.astype(np.float64), n_fft, hop, win, pad)
        assert Sx.shape == Sx_o.shape == (n_fft // 2 + 1, (x.shape[0] - 1) // hop + 1)
        assert np.array_equal(f, f_o)
        assert _relerr(Sx, Sx_o) <= tol, (n_fft, pad)


@pytest.mark.parametrize("n_fft,hop", [(1000, 250), (333, 83), (48, 12), (1536, 384), (3000, 750), (231, 50)])
def test_ssq_stft_bluestein(n_fft, hop):
    x = _sig(6000 + 3 * n_fft, 22)
    _check_ssq_f64(x, np.hanning(n_fft), n_fft, hop, 2.0, "reflect", "sum")
    _check_ssq_f64(x, np.hanning(n_fft), n_fft, hop, 1.0, "zero", "lebesgue")
    _check_ssq_f32(x.astype(np.float32), np.hanning(n_fft), n_fft, hop, 1.0)
    xb = np.stack([_sig(4000, 30 + b, np.float32) for b in range(3)])      # NaN beyond a frame must not leak into it
    Tb, _ = _rs.ssq_stft(xb, np.hanning(n_fft), n_fft=n_fft, hop_len=hop)
    for b in range(3):
        assert np.array_equal(Tb[b], _rs.ssq_stft(xb[b], np.hanning(n_fft), n_fft=n_fft, hop_len=hop)[0])


def test_bluestein_reads_only_the_frame():
    """The kernel's transform is longer than the frame; samples beyond the frame's n_fft (here a NaN right behind a
    short signal's last frame position cannot exist, so put NaNs INSIDE and check they stay local)."""
    n_fft, hop = 1000, 250
    x = _sig(8000, 23)
    x[4000] = np.nan
    Sx, _ = _rs.stft(x, n_fft, hop, np.hanning(n_fft), "reflect")
    bad = np.isnan(Sx).any(axis=0)
    frames = np.arange(Sx.shape[1])
    start = frames * hop - (n_fft - 1) // 2
    touches = (start <= 4000) & (4000 < start + n_fft)
    assert np.array_equal(bad, touches)


@pytest.mark.parametrize("n_fft,hop", [(8192, 2048), (3000, 750), (5000, 1250), (16384, 4096), (4097, 1000)])
def test_stft_long_lengths_through_the_batched_device_fft(n_fft, hop):
    """n_fft beyond the fused kernels (> 4096, or > 2048 and not a power of two): pack -> batched any-length FFT
    (Stockham passes / Bluestein) -> unpack, instead of O(n_fft) sums per bin."""
    x = _sig(3 * n_fft + 1001, 24)
    win = np.hanning(n_fft) + 0.03
    Sx, f = _rs.stft(x, n_fft, hop, win, "reflect")
    Sx_o, f_o = o.stft(x, n_fft, hop, win, "reflect")
    assert np.array_equal(f, f_o) and _relerr(Sx, Sx_o) <= 1e-11
    S32, _ = _rs.stft(x.astype(np.float32), n_fft, hop, win, "zero")
    S32_o, _ = o.stft(x.astype(np.float32).astype(np.float64), n_fft, hop, win, "zero")
    assert _relerr(S32, S32_o) <= 2e-5


def test_ssq_stft_long_length():
    x = _sig(20000, 25)
    _check_ssq_f64(x, np.hanning(3000), 3000, 750, 1.0, "reflect", "sum")
    xb = np.stack([_sig(20000, 40 + b) for b in range(2)])
    Tb, _ = _rs.ssq_stft(xb, np.hanning(8192), n_fft=8192, hop_len=2048)
    for b in range(2):
        assert np.array_equal(Tb[b], _rs.ssq_stft(xb[b], np.hanning(8192), n_fft=8192, hop_len=2048)[0])
