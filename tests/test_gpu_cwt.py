"""GPU parity tests of the CWT family (cwt / cwt_simd / ssq_cwt) against the NumPy oracle.

Tolerances: fp64 |dWx|/max|Wx| <= 1e-11; fp32 <= 1e-5 (length-P FFTs up to 2^15 here and an
un-normalised GMW).  Reassignment bins: exact given the kernel's own w except
within 1e-6 (fp64) of a rounding boundary; Tx is compared after re-accumulation with the
kernel's own bins.
"""
import numpy as np
import pytest

from oracle import ssq_oracle as o
from ssqueeze_rs_amd import _lib, _rs

pytestmark = pytest.mark.gpu


def _sig(N, seed=0, dtype=np.float64):
    return o.synth_signal(N, seed, dtype)


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_cwt_reference_smoke_shapes():
    """tests/cwt_test.py:17-66: 1 kHz / 100 Hz sine, scales = logspace(1,5,32)/fs, GMW;
    the one thing the reference states: Wx.shape == (len(scales), len(x))."""
    fs = 1000
    t = np.linspace(0, 1, fs, endpoint=False)
    x = np.sin(2 * np.pi * 100 * t)
    scales = np.logspace(1, 5, 32) / fs
    Wx, sc, dWx = _rs.cwt(x, wavelet="gmw", scales=scales, fs=fs, nv=16, l1_norm=True, derivative=True)
    assert Wx.shape == (32, 1000) and dWx.shape == (32, 1000) and np.array_equal(sc, scales)
    Wx_o, _, dWx_o = o.cwt(x, "gmw", scales=scales, fs=fs, nv=16, derivative=True)
    assert _rel(Wx, Wx_o) <= 1e-11 and _rel(dWx, dWx_o) <= 1e-11
    out = _rs.cwt(x, wavelet="gmw", scales=scales, fs=fs)       # derivative=False: still a 3-tuple
    assert len(out) == 3 and out[2] is None


@pytest.mark.parametrize("N", [5, 11, 40, 300, 1000, 2730, 2731, 6000, 20000])
@pytest.mark.parametrize("wavelet", ["morlet", "gmw"])
def test_cwt_f64_sizes(N, wavelet):
    """N <= 10: P < 16 direct sums; N <= 2730: one-step LDS FFT; above: two-step FFT."""
    x = _sig(N, 1)
    Wx, sc, dWx = _rs.cwt(x, wavelet=wavelet, nv=4, derivative=True) if N >= 40 else \
        _rs.cwt(x, wavelet=wavelet, scales=np.array([1.0, 2.0, 3.5]), derivative=True)
    kw = dict(nv=4) if N >= 40 else dict(scales=np.array([1.0, 2.0, 3.5]))
    Wx_o, sc_o, dWx_o = o.cwt(x, wavelet, derivative=True, **kw)
    assert Wx.shape == Wx_o.shape
    assert np.array_equal(sc, sc_o)
    assert _rel(Wx, Wx_o) <= 1e-11, N
    assert _rel(dWx, dWx_o) <= 1e-11, N


def test_cwt_options_f64():
    x = _sig(3000, 2)
    sc = 2.0 ** np.linspace(1, 9, 24)
    for kw in (dict(l1_norm=False), dict(rpadded=True), dict(padtype="zero"), dict(fs=250.0),
               dict(t=np.arange(3000) * 0.004), dict(l1_norm=False, rpadded=True, derivative=True)):
        a = _rs.cwt(x, wavelet="morlet", scales=sc, **kw)
        b = o.cwt(x, "morlet", scales=sc, **kw)
        assert a[0].shape == b[0].shape
        assert _rel(a[0], b[0]) <= 1e-11, kw
        if kw.get("derivative"):
            assert _rel(a[2], b[2]) <= 1e-11
    # cwt_simd: same numbers, scale generation via exp(p*ln2)
    a = _rs.cwt_simd(x, wavelet="morlet", nv=8)
    b = o.cwt_simd(x, "morlet", nv=8)
    assert np.array_equal(a[1], b[1]) and _rel(a[0], b[0]) <= 1e-11
    with pytest.raises(ValueError):
        _rs.cwt(x, t=np.array([0.0]))


@pytest.mark.parametrize("N", [1000, 20000])
@pytest.mark.parametrize("wavelet", ["morlet", "gmw"])
def test_cwt_f32(N, wavelet):
    x = _sig(N, 3, np.float32)
    Wx, sc, dWx = _rs.cwt(x, wavelet=wavelet, nv=4, derivative=True)
    assert Wx.dtype == np.complex64
    Wx_o, _, dWx_o = o.cwt(x.astype(np.float64), wavelet, nv=4, derivative=True)
    assert _rel(Wx, Wx_o) <= 1e-5
    assert _rel(dWx, dWx_o) <= 1e-5


def _check_ssq_cwt(x, tol_w, **kw):
    Tx, f, dbg = _rs.ssq_cwt(x, _debug=True, **kw)
    Tx_o, f_o, im = o.ssq_cwt(x.astype(np.float64), return_intermediates=True, **kw)
    assert Tx.shape == Tx_o.shape == (im["scales"].shape[0], x.shape[0])
    assert np.array_equal(f, f_o)
    wmax = np.abs(im["Wx"]).max()
    assert np.abs(dbg["Wx"] - im["Wx"]).max() <= tol_w * wmax
    assert np.abs(dbg["dWx"] - im["dWx"]).max() <= tol_w * np.abs(im["dWx"]).max()
    # bins from the kernel's own w through the oracle's binning (ssq_cwt.rs:135-196)
    w_g = dbg["w"].astype(np.float64)
    b, valid, is_log = o.cwt_bins(w_g, f_o)
    flipud = kw.get("flipud", True)
    na = Tx.shape[0]
    k_model = np.where(valid, (na - 1 - b) if flipud else b, -1)
    mism = k_model != dbg["k"]
    if mism.any():
        # only allowed at a rounding boundary of the bin formula (fp32 evaluates it in fp32)
        with np.errstate(all="ignore"):
            if is_log:
                lmin = np.log2(f_o[0]); lstep = (np.log2(f_o[-1]) - lmin) / (na - 1)
                v = (np.log2(w_g[mism]) - lmin) / lstep
            else:
                lstep = (f_o[-1] - f_o[0]) / (na - 1)
                v = (w_g[mism] - f_o[0]) / lstep
        # fp32 evaluates the bin formula (log2 / divide) in fp32; measured worst distance from a rounding boundary among
        # the differing elements: 3.6e-6 bins, rate 1.5e-6 (profiles/r03_bin_parity.json) -> bounds ~3x / ~7x that
        tie_tol = 1e-6 if x.dtype == np.float64 else 1e-5
        assert (np.abs(np.abs(v - np.trunc(v)) - 0.5) < tie_tol).all(), f"{mism.sum()} unexplained"
        assert mism.mean() <= (1e-4 if x.dtype == np.float64 else 1e-5)
    keep = dbg["k"] >= 0
    # scatter: re-accumulate with the kernel's own bins (no dw factor in the CWT path)
    Tx_re = np.zeros_like(Tx_o)
    cols = np.arange(x.shape[0])
    leb = kw.get("squeezing") == "lebesgue"
    Wg = dbg["Wx"].astype(np.complex128)
    for i in range(na):
        m = keep[i]
        if leb:
            Tx_re[dbg["k"][i, m], cols[m]] += 1.0 / na
        else:
            Tx_re[dbg["k"][i, m], cols[m]] += Wg[i, m]
    assert np.abs(Tx - Tx_re).max() <= (1e-10 if x.dtype == np.float64 else 2e-5) * max(np.abs(Tx_re).max(), 1e-300)
    # end to end: column sums are invariant under bin flips (not under keep/drop flips)
    # keep/drop may only differ where Wx is rounding noise (the un-normalised GMW leaves ~1e17*eps there)
    flips = keep != im["valid"]
    noise = (1e-9 if x.dtype == np.float64 else 2e-4) * wmax
    assert np.abs(im["Wx"][flips]).max(initial=0.0) <= noise
    if not leb:
        d = np.abs(Tx.sum(0) - Tx_o.sum(0))
        assert d.max(initial=0.0) <= (1e-9 if x.dtype == np.float64 else 1e-3) * wmax + flips.sum(0).max() * noise
    return dbg, im


def test_ssq_cwt_reference_smoke():
    """tests/ssq_cwt_test.py:17-63: Tx.shape == (len(scales), len(x)); is_log path (ratio 1.346)."""
    fs = 1000
    t = np.linspace(0, 1, fs, endpoint=False)
    x = np.sin(2 * np.pi * 100 * t)
    scales = np.logspace(1, 5, 32) / fs
    dbg, im = _check_ssq_cwt(x, 1e-11, wavelet="gmw", scales=scales, fs=fs, nv=16, padtype="reflect",
                             squeezing="sum", maprange="peak")
    assert im["is_log"]


@pytest.mark.parametrize("kw", [
    dict(wavelet="morlet", nv=8),                                   # 64+ log scales: is_log False quirk
    dict(wavelet="morlet", nv=8, flipud=False),
    dict(wavelet="morlet", nv=8, maprange="maximal"),
    dict(wavelet="morlet", nv=8, ssq_freqs="linear"),
    dict(wavelet="morlet", nv=8, squeezing="lebesgue"),
    dict(wavelet="gmw", nv=2, fs=100.0),                            # few scales: is_log True
    dict(wavelet="morlet", nv=8, padtype="zero", gamma=1e-2),
])
def test_ssq_cwt_f64_options(kw):
    _check_ssq_cwt(_sig(4096, 4), 1e-11, **kw)


def test_ssq_cwt_f64_two_step():
    _check_ssq_cwt(_sig(12000, 5), 1e-11, wavelet="morlet", nv=4)


@pytest.mark.parametrize("kw", [dict(wavelet="morlet", nv=8), dict(wavelet="gmw", nv=2, fs=100.0)])
def test_ssq_cwt_f32(kw):
    _check_ssq_cwt(_sig(4096, 6, np.float32), 1e-5, **kw)
    _check_ssq_cwt(_sig(12000, 6, np.float32), 1e-5, **kw)


def test_ssq_cwt_batch_and_errors():
    xb = np.stack([_sig(2000, b) for b in range(3)])
    Tb, f = _rs.ssq_cwt(xb, wavelet="morlet", nv=4)
    for b in range(3):
        T1, _ = _rs.ssq_cwt(xb[b], wavelet="morlet", nv=4)
        assert np.array_equal(Tb[b], T1)
    with pytest.raises(ValueError):
        _rs.ssq_cwt(xb[0], t=np.array([1.0]))
    with pytest.raises(TypeError):
        _rs.ssq_cwt(xb[0], ssq_freqs=np.arange(4.0))     # a string in the reference (ssq_cwt.rs:268)
    a, _ = _rs.ssq_cwt(xb[0], wavelet="nonsense", nv=4)  # unknown wavelet -> gmw (cwt.rs:522)
    b, _ = _rs.ssq_cwt(xb[0], wavelet="gmw", nv=4)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-11), (np.float32, 2e-5)])
def test_ssq_cwt_large_two_step(dtype, tol):
    """N = 2^16 (P = 2^17 = 512 x 256 two-step FFT), 64 log scales over [2, N/2]: the C4/C5 code path at a size
    the oracle finishes in seconds."""
    N = 1 << 16
    x = _sig(N, 7, dtype)
    scales = 2.0 ** np.linspace(1, 15, 64)
    Tx, f, dbg = _rs.ssq_cwt(x, wavelet="morlet", scales=scales, _debug=True)
    Tx_o, f_o, im = o.ssq_cwt(x.astype(np.float64), "morlet", scales=scales, return_intermediates=True)
    assert Tx.shape == (64, N) and np.array_equal(f, f_o)
    wmax = np.abs(im["Wx"]).max()
    assert np.abs(dbg["Wx"] - im["Wx"]).max() <= tol * wmax
    assert np.abs(dbg["dWx"] - im["dWx"]).max() <= tol * np.abs(im["dWx"]).max()
    keep = dbg["k"] >= 0
    assert (keep == im["valid"]).mean() >= 0.999
    both = keep & im["valid"]
    assert (dbg["k"][both] == im["k"][both]).mean() >= (0.9999 if dtype == np.float64 else 0.99)
    d = np.abs(Tx.astype(np.complex128).sum(0) - Tx_o.sum(0))
    assert np.median(d) <= (1e-9 if dtype == np.float64 else 1e-4) * wmax


@pytest.mark.parametrize("wavelet", ["morlet", "gmw"])
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 2e-6)])
def test_cwt_band_limited_paths_match_plain_two_step(wavelet, dtype, tol, monkeypatch):
    """Band-limited scales take the single-pass path (mode Z) and step A skips dead rows; both rely on the wavelet
    table being exactly zero beyond a bound (csrc/api_cwt.hip::wavelet_support).  SSQ_CWT_NOPRUNE=1 sends every
    scale through the plain two-step transform: the two must agree to rounding, for scales on both sides of every
    switch (Q = 16 ... 2048, two-step) and for both wavelets."""
    N = 40000                                    # P = 65536 = 256 x 256
    x = _sig(N, 11, dtype)
    scales = 2.0 ** np.linspace(0.5, 14.5, 57)
    Wx, sc, dWx = _rs.cwt(x, wavelet=wavelet, scales=scales, fs=50.0, l1_norm=False, derivative=True)
    monkeypatch.setenv("SSQ_CWT_NOPRUNE", "1")
    Wx0, sc0, dWx0 = _rs.cwt(x, wavelet=wavelet, scales=scales, fs=50.0, l1_norm=False, derivative=True)
    assert np.array_equal(sc, sc0)
    for a, b in ((Wx, Wx0), (dWx, dWx0)):
        row_max = np.abs(b).max(axis=1, keepdims=True)
        assert (np.abs(a - b) <= tol * np.maximum(row_max, np.abs(b).max() * 1e-30)).all()
    # and against the oracle, per scale (a dropped non-zero tail would show up in the large scales' rows)
    Wx_o, _, dWx_o = o.cwt(x.astype(np.float64), wavelet, scales=scales, fs=50.0, l1_norm=False, derivative=True)
    otol = 1e-11 if dtype == np.float64 else 2e-5
    assert np.abs(Wx - Wx_o).max() <= otol * np.abs(Wx_o).max()
    assert np.abs(dWx - dWx_o).max() <= otol * np.abs(dWx_o).max()


def test_cwt_c5_size_scale_subset():
    """BASELINE config 5 geometry (N = 2^22 -> P = 2^23 = 2048 x 4096 two-step FFT, fp64) on a subset of its 256
    log scales that spans every path (plain two-step, dead-row skipping, single-pass Q = 2048 ... 16): index
    arithmetic at full size against the oracle."""
    N = 1 << 22
    x = _sig(N, 21)
    scales = (2.0 ** np.linspace(1, 21, 256))[[0, 37, 90, 140, 171, 200, 231, 255]]
    Wx, sc, dWx = _rs.cwt(x, wavelet="morlet", scales=scales, derivative=True)
    Wx_o, _, dWx_o = o.cwt(x, "morlet", scales=scales, derivative=True)
    assert Wx.shape == (8, N)
    for i in range(8):                                    # per scale: a lost tail would hide behind the largest row
        assert np.abs(Wx[i] - Wx_o[i]).max() <= 1e-11 * np.abs(Wx_o[i]).max(), i
        assert np.abs(dWx[i] - dWx_o[i]).max() <= 1e-11 * np.abs(dWx_o[i]).max(), i


@pytest.mark.parametrize("wavelet", ["morlet", "gmw"])
def test_cwt_c4_size_scale_subset_f32(wavelet):
    """BASELINE config 4 geometry (N = 2^20 -> P = 2^21 = 2048 x 1024 two-step FFT, fp32) on a subset of its 256 log
    scales spanning the plain two-step path (2048-point multi-wave columns, 8-row step-B tiles), dead-row skipping and
    the single-pass scales: per-scale agreement with the fp64 oracle."""
    N = 1 << 20
    x = _sig(N, 22, np.float32)
    scales = (2.0 ** np.linspace(1, 19, 256))[[0, 30, 77, 120, 150, 165, 200, 255]]
    Wx, sc, dWx = _rs.cwt(x, wavelet=wavelet, scales=scales, derivative=True)
    Wx_o, _, dWx_o = o.cwt(x.astype(np.float64), wavelet, scales=scales, derivative=True)
    assert Wx.dtype == np.complex64 and Wx.shape == (8, N)
    for i in range(8):
        assert np.abs(Wx[i] - Wx_o[i]).max() <= 2e-5 * np.abs(Wx_o[i]).max(), i
        assert np.abs(dWx[i] - dWx_o[i]).max() <= 2e-5 * np.abs(dWx_o[i]).max(), i


def test_fused_step_b_equals_unfused(monkeypatch):
    """SSQ_CWT_FUSED=1 (phase transform + bin inside inverse step B / mode Z, Wx + 16-bit row index out) against the
    default path (Wx and dWx materialised, phase in the reassignment kernel).  The two run different tile
    configurations of the same FFT (differently contracted FMAs), so Wx / dWx agree to rounding, the bin of an element
    may differ only where the phase lands on opposite sides of a rounding boundary, and Tx agrees to rounding in every
    column without such an element.  N = 20000 -> P = 32768: two-step scales and mode-Z scales."""
    for dtype, tol, rate in ((np.float64, 1e-13, 1e-5), (np.float32, 2e-6, 5e-3)):
        x = _sig(20000, 11, dtype)
        outs = []
        for mode in ("0", "1"):
            if not _lib.load().ssq_build_has_tuning():
                pytest.skip("SSQ_CWT_FUSED exists only in -DSSQ_TUNING builds (python -m ssqueeze_rs_amd.build --tune)")
            monkeypatch.setenv("SSQ_CWT_FUSED", mode)
            outs.append(_rs.ssq_cwt(x, wavelet="morlet", nv=6, _debug=True))
        (T0, f0, d0), (T1, f1, d1) = outs
        assert np.array_equal(f0, f1)
        wmax = np.abs(d0["Wx"]).max()
        assert np.abs(d0["Wx"] - d1["Wx"]).max() <= tol * wmax
        assert np.abs(d0["dWx"] - d1["dWx"]).max() <= tol * np.abs(d0["dWx"]).max()
        diff = d0["k"] != d1["k"]
        assert diff.mean() <= rate, diff.mean()
        clean = ~diff.any(axis=0)
        assert clean.mean() > 0.9
        assert np.abs(T0[:, clean] - T1[:, clean]).max() <= 50 * tol * np.abs(T0).max()
        assert np.abs(T0.sum(0) - T1.sum(0)).max() <= 50 * tol * np.abs(T0).max() + np.abs(d0["Wx"][diff]).max(initial=0.0)


def test_big_padded_length_path(monkeypatch):
    """Padded lengths above 2^24 (the reference takes any N, cwt.rs:87) run the same pipeline through the batched generic
    device FFT; SSQ_CWT_FORCE_BIG=1 selects that path at a testable size.  Checked against the oracle like every other
    path, and against the tile transforms."""
    monkeypatch.setenv("SSQ_CWT_FORCE_BIG", "1")
    _check_ssq_cwt(_sig(12000, 5), 1e-11, wavelet="morlet", nv=4)
    _check_ssq_cwt(_sig(5000, 6, np.float32), 1e-5, wavelet="gmw", nv=2, fs=100.0)
    x = _sig(20000, 12)
    W1, sc, dW1 = _rs.cwt(x, wavelet="morlet", nv=5, derivative=True, l1_norm=False, rpadded=True)
    monkeypatch.setenv("SSQ_CWT_FORCE_BIG", "0")
    W0, _, dW0 = _rs.cwt(x, wavelet="morlet", nv=5, derivative=True, l1_norm=False, rpadded=True)
    assert _rel(W1, W0) <= 1e-12 and _rel(dW1, dW0) <= 1e-12


@pytest.mark.parametrize("wavelet", ["morlet", "gmw"])
@pytest.mark.parametrize("N", [600_000, 1 << 20])
def test_cwt_register_core_path(wavelet, N, monkeypatch):
    """fp32 plans with P = 2^20 (N = 600 000: one residue) and P = 2^21 (N = 2^20, C4's geometry: two residues and the
    k = P/2 term) run their two-step scales on the per-wave register FFT core (csrc/cwt_reg.hip).  Per scale against the
    oracle, and against the tile kernels (SSQ_CWT_REG=0) to rounding; the scales sit on both sides of the band switch
    (full band with a live k = P/2 term, partly dead rows, single-pass scales)."""
    monkeypatch.setenv("SSQ_CWT_OS_STORE", "0")          # (the tiles that store Wx / dWx have their own test)
    x = _sig(N, 5, np.float32)
    scales = np.array([1.0, 1.7, 3.1, 6.0, 19.0, 77.0, 150.0, 900.0, 20000.0])
    Wx, sc, dWx = _rs.cwt(x, wavelet=wavelet, scales=scales, fs=20.0, l1_norm=False, derivative=True)
    Wp, _, dWp = _rs.cwt(x, wavelet=wavelet, scales=scales, fs=20.0, l1_norm=True, derivative=True, rpadded=True)
    W1, _, none = _rs.cwt(x, wavelet=wavelet, scales=scales, fs=20.0, l1_norm=False, derivative=False)
    assert none is None and np.array_equal(W1, Wx)       # one transform per scale: the same arithmetic per transform
    monkeypatch.setenv("SSQ_CWT_REG", "0")
    Wx0, sc0, dWx0 = _rs.cwt(x, wavelet=wavelet, scales=scales, fs=20.0, l1_norm=False, derivative=True)
    Wp0, _, dWp0 = _rs.cwt(x, wavelet=wavelet, scales=scales, fs=20.0, l1_norm=True, derivative=True, rpadded=True)
    assert np.array_equal(sc, sc0) and Wp.shape == Wp0.shape
    for a, b in ((Wx, Wx0), (dWx, dWx0), (Wp, Wp0), (dWp, dWp0)):
        row_max = np.abs(b).max(axis=1, keepdims=True)
        assert (np.abs(a - b) <= 4e-6 * row_max).all()
    Wx_o, _, dWx_o = o.cwt(x.astype(np.float64), wavelet, scales=scales, fs=20.0, l1_norm=False, derivative=True)
    for i in range(len(scales)):
        assert np.abs(Wx[i] - Wx_o[i]).max() <= 2e-5 * np.abs(Wx_o[i]).max(), i
        assert np.abs(dWx[i] - dWx_o[i]).max() <= 2e-5 * np.abs(dWx_o[i]).max(), i


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_ssq_cwt_sweep_reassignment_equals_clear_and_rmw(dtype, monkeypatch):
    """The reassignment keeps a per-lane bitmap of the rows it has written: the first run of a row stores without reading
    (default), optionally the untouched rows are stored as zeros by the kernel and nothing clears Tx (SSQ_CWT_SWEEP=2);
    SSQ_CWT_SWEEP=0 is the plain read-modify-write of a cleared Tx.  Same sums in the same order: the three must be
    identical.  N is not a multiple of 64 and na not a multiple of 32 (dead lanes, a partial bitmap word); the output
    buffer is poisoned first."""
    N = 5000 + 37
    x = _sig(N, 3, dtype)
    scales = 2.0 ** np.linspace(1, 10, 45)
    import ssqueeze_rs_amd._lib as L
    poison = L.pinned_empty((45, N), np.complex64 if dtype == np.float32 else np.complex128)
    poison[:] = np.nan
    del poison                                            # goes back to the pool the next result comes from
    Tx, f = _rs.ssq_cwt(x, wavelet="morlet", scales=scales)
    assert np.isfinite(Tx.view(dtype)).all()
    assert np.count_nonzero(Tx) > 0
    for mode in ("0", "2"):                               # read-modify-write of a cleared Tx | bitmap + own zero rows
        if not _lib.load().ssq_build_has_tuning():
            continue                              # the alternatives exist only in -DSSQ_TUNING builds
        monkeypatch.setenv("SSQ_CWT_SWEEP", mode)
        Tx0, f0 = _rs.ssq_cwt(x, wavelet="morlet", scales=scales)
        assert np.array_equal(f, f0) and np.array_equal(Tx, Tx0), mode



def test_ssq_cwt_register_core_one_residue(monkeypatch):
    """ssq_cwt on a padded length of 2^20 (N = 600 000: ONE residue per transform on the register core, no k = P/2 term),
    batch of two signals through one workspace: Wx / dWx to rounding against the tile kernels (SSQ_CWT_REG=0), the same
    bins for all but a handful of elements, the same column sums of Tx."""
    N = 600_000
    xb = np.stack([_sig(N, 31, np.float32), _sig(N, 32, np.float32)])
    scales = 2.0 ** np.linspace(1.0, 17.0, 24)
    monkeypatch.setenv("SSQ_CWT_OS", "0")               # (the time-tiled path has its own test)
    out = [_rs.ssq_cwt(x, wavelet="gmw", scales=scales, _debug=True) for x in xb]
    monkeypatch.setenv("SSQ_CWT_REG", "0")
    ref = [_rs.ssq_cwt(x, wavelet="gmw", scales=scales, _debug=True) for x in xb]
    for (Tx, f, dbg), (Tx0, f0, dbg0) in zip(out, ref):
        assert np.array_equal(f, f0)
        for key in ("Wx", "dWx"):
            row_max = np.abs(dbg0[key]).max(axis=1, keepdims=True)
            assert (np.abs(dbg[key] - dbg0[key]) <= 4e-6 * row_max).all(), key
        assert (dbg["k"] == dbg0["k"]).mean() >= 0.999
        cs, cs0 = Tx.astype(np.complex128).sum(0), Tx0.astype(np.complex128).sum(0)
        assert np.abs(cs - cs0).max() <= 1e-3 * np.abs(cs0).max()


@pytest.mark.parametrize("N", [(1 << 20) - 1234, 1 << 19, 700_000])
@pytest.mark.parametrize("wavelet", ["morlet", "gmw"])
def test_ssq_cwt_time_tiled_scales(wavelet, N, monkeypatch):
    """fp32 plans at C4's geometry run the scales whose wavelet is short in time by overlap-save tiles
    (csrc/cwt_os.hip: 4096 output samples from 8192-point transforms, bins and run merging on chip, Wx / dWx of those
    scales never in memory); SSQ_CWT_OS=0 keeps the frequency-domain path for every scale.  The two differ only by the
    wavelet's tail beyond the tile halo (< 1e-8 of its peak): Wx / dWx per scale to fp32 rounding (1e-5 of the row maximum), the same bins for
    all but a handful of elements, the same Tx column sums -- and the oracle's Wx on a subset of the tiled scales.
    N = 2^20 - 1234 is not a multiple of the tile length (a partial last tile); N = 2^19 has P = 2 N, where the
    band-limited scales too run as phase blocks over the whole padded signal (exact circular convolution, the
    reference's own) instead of mode Z + the column reassignment; N = 700 000 has 2 N < P = 2^21 (the same blocks, their
    window wider than the kept samples)."""
    x = _sig(N, 41, np.float32)
    scales = 2.0 ** np.linspace(1.0, 19.0, 64)
    Tx, f, dbg = _rs.ssq_cwt(x, wavelet=wavelet, scales=scales, _debug=True)
    monkeypatch.setenv("SSQ_CWT_OS", "0")
    Tx0, f0, dbg0 = _rs.ssq_cwt(x, wavelet=wavelet, scales=scales, _debug=True)
    assert np.array_equal(f, f0)
    differ = [i for i in range(64) if not np.array_equal(dbg["Wx"][i], dbg0["Wx"][i])]
    assert len(differ) >= 8                               # the tiled path really ran (other scales are bit-identical)
    for key in ("Wx", "dWx"):
        row_max = np.abs(dbg0[key]).max(axis=1, keepdims=True)
        assert (np.abs(dbg[key] - dbg0[key]) <= 1e-5 * row_max).all(), key     # two fp32 pipelines; oracle below: 2e-5
    assert (dbg["k"] == dbg0["k"]).mean() >= 0.999
    cs, cs0 = Tx.astype(np.complex128).sum(0), Tx0.astype(np.complex128).sum(0)
    assert np.abs(cs - cs0).max() <= 1e-3 * np.abs(cs0).max()
    re, re0 = (np.abs(Tx.astype(np.complex128)) ** 2).sum(1), (np.abs(Tx0.astype(np.complex128)) ** 2).sum(1)
    assert np.abs(re - re0).max() <= 2e-3 * re0.max()
    sub = np.array(differ[:: max(1, len(differ) // 4)])
    Wx_o, _, dWx_o = o.cwt(x.astype(np.float64), wavelet, scales=scales[sub], derivative=True)
    # (5e-5: the coarse scales' response is small against the fp32 rounding noise of the 2^20-point spectrum both paths
    # share -- the frequency-domain path misses 2e-5 by the same 10 % there)
    for j, i in enumerate(sub):
        assert np.abs(dbg["Wx"][i] - Wx_o[j]).max() <= 5e-5 * np.abs(Wx_o[j]).max(), i
        assert np.abs(dbg["dWx"][i] - dWx_o[j]).max() <= 5e-5 * np.abs(dWx_o[j]).max(), i


def test_ssq_cwt_time_tiled_batch_equals_single_signals():
    """Two signals through one plan / one workspace with the time-tiled path active: each must equal its own
    single-signal call bit for bit (the tiles' spectra and the Tx clear are per signal)."""
    N = 1 << 19
    xb = np.stack([_sig(N, 51, np.float32), _sig(N, 52, np.float32)])
    scales = 2.0 ** np.linspace(1.5, 16.0, 40)
    Txb, fb = _rs.ssq_cwt(xb, wavelet="morlet", scales=scales)
    for i in range(2):
        Tx, f = _rs.ssq_cwt(xb[i], wavelet="morlet", scales=scales)
        assert np.array_equal(f, fb) and np.array_equal(Tx, Txb[i]), i
    assert np.count_nonzero(Txb[1]) > 0


@pytest.mark.parametrize("opts", [
    dict(squeezing="lebesgue", flipud=False),
    dict(fs=50.0, gamma=1e-3, maprange="maximal"),
    dict(padtype="zero", ssq_freqs="linear"),
])
def test_ssq_cwt_time_tiled_options(opts, monkeypatch):
    """The time-tile family takes the same options as the frequency-domain path (lebesgue weights, no flip, a sampling
    rate, a threshold, the other frequency maps, zero padding): Tx against SSQ_CWT_OS=0 by column sums and row energies,
    the same bins for all but a handful of elements.  N = 2^19: plain, decimated and full-circle tiles all run."""
    N = 1 << 19
    x = _sig(N, 61, np.float32)
    scales = 2.0 ** np.linspace(1.0, 18.0, 48)
    Tx, f, dbg = _rs.ssq_cwt(x, wavelet="morlet", scales=scales, _debug=True, **opts)
    monkeypatch.setenv("SSQ_CWT_OS", "0")
    Tx0, f0, dbg0 = _rs.ssq_cwt(x, wavelet="morlet", scales=scales, _debug=True, **opts)
    assert np.array_equal(f, f0)
    assert sum(not np.array_equal(dbg["Wx"][i], dbg0["Wx"][i]) for i in range(48)) >= 20
    assert (dbg["k"] == dbg0["k"]).mean() >= 0.999
    cs, cs0 = Tx.astype(np.complex128).sum(0), Tx0.astype(np.complex128).sum(0)
    # (a column may differ by whole elements whose keep / bin decision sits on a threshold: with lebesgue weights one such
    # element moves a column sum by 1 / na)
    assert (np.abs(cs - cs0) > 1e-3 * np.abs(cs0).max()).mean() <= 2e-3
    re, re0 = (np.abs(Tx.astype(np.complex128)) ** 2).sum(1), (np.abs(Tx0.astype(np.complex128)) ** 2).sum(1)
    assert np.abs(re - re0).max() <= 2e-3 * re0.max()


@pytest.mark.parametrize("l1_norm", [True, False])
def test_cwt_short_wavelet_scales_by_storing_tiles(l1_norm, monkeypatch):
    """`cwt` with the derivative runs the scales whose wavelet is short in time (and, on register-core plans, the finest
    ones through the analytic signal) on the same time tiles as ssq_cwt, which then STORE Wx / dWx instead of binning
    them; SSQ_CWT_OS_STORE=0 keeps the frequency-domain transforms.  Per scale to fp32 rounding against those and against
    the oracle, both norms, N not a multiple of the tile length."""
    N = (1 << 20) - 777
    x = _sig(N, 71, np.float32)
    scales = 2.0 ** np.linspace(1.0, 12.0, 23)
    Wx, sc, dWx = _rs.cwt(x, wavelet="morlet", scales=scales, fs=4.0, l1_norm=l1_norm, derivative=True)
    monkeypatch.setenv("SSQ_CWT_OS_STORE", "0")
    Wx0, _, dWx0 = _rs.cwt(x, wavelet="morlet", scales=scales, fs=4.0, l1_norm=l1_norm, derivative=True)
    if l1_norm:                                           # a batch of two through one workspace = the single calls
        xb = np.stack([x, _sig(N, 72, np.float32)])
        Wb, _, dWb = _rs.cwt(xb, wavelet="morlet", scales=scales[:6], fs=4.0, l1_norm=True, derivative=True)
        W1, _, dW1 = _rs.cwt(xb[1], wavelet="morlet", scales=scales[:6], fs=4.0, l1_norm=True, derivative=True)
        assert np.array_equal(Wb[1], W1) and np.array_equal(dWb[1], dW1)
    differ = [i for i in range(len(scales)) if not np.array_equal(Wx[i], Wx0[i])]
    assert len(differ) >= 10
    for a, b in ((Wx, Wx0), (dWx, dWx0)):
        row_max = np.abs(b).max(axis=1, keepdims=True)
        assert (np.abs(a - b) <= 1e-5 * row_max).all()
    sub = np.array(differ[::4])
    Wx_o, _, dWx_o = o.cwt(x.astype(np.float64), "morlet", scales=scales[sub], fs=4.0, l1_norm=l1_norm, derivative=True)
    for j, i in enumerate(sub):
        assert np.abs(Wx[i] - Wx_o[j]).max() <= 2e-5 * np.abs(Wx_o[j]).max(), i
        assert np.abs(dWx[i] - dWx_o[j]).max() <= 2e-5 * np.abs(dWx_o[j]).max(), i


@pytest.mark.parametrize("wavelet,a_lo,a_hi", [("morlet", 4.0, 2048.0 / 6.0), ("gmw", 1.3, 2048.0 / (6.0 * 4.943))])
def test_time_tile_eligibility_limits_are_pinned(wavelet, a_lo, a_hi, monkeypatch):
    """the plain time tiles cut the wavelet's time response at the tile halo (6 sigma_t <= 2048 samples,
    sigma_t = a for the Morlet wavelet, 4.943 a for the GMW) and need psih negligible at Nyquist (a >= 4 / 1.3): the
    eligibility limits of csrc/api_cwt.hip.  Scales EXACTLY at both limits (and just outside them, which must take the
    exact frequency-domain path) against that path: the truncated tail stays below 1e-5 of each row's maximum, so a later
    change of the thresholds cannot silently widen the error."""
    N = 1 << 18                                            # >= 64 tiles of 4096: the family is on
    x = _sig(N, 83, np.float32)
    inside = np.geomspace(a_lo, a_hi, 12)                  # first and last ARE the limits
    scales = np.concatenate([[a_lo * 0.97], inside, [a_hi * 1.03]])
    Wx, _, dWx = _rs.cwt(x, wavelet=wavelet, scales=scales, derivative=True)
    monkeypatch.setenv("SSQ_CWT_OS_STORE", "0")
    Wx0, _, dWx0 = _rs.cwt(x, wavelet=wavelet, scales=scales, derivative=True)
    tiled = [i for i in range(len(scales)) if not np.array_equal(Wx[i], Wx0[i])]
    assert 1 in tiled and len(scales) - 2 in tiled, tiled   # the limits themselves are tiled ...
    assert 0 not in tiled and len(scales) - 1 not in tiled  # ... the scales just outside are not
    for a, b in ((Wx, Wx0), (dWx, dWx0)):
        row_max = np.abs(b).max(axis=1, keepdims=True)
        err = (np.abs(a - b) / row_max).max(axis=1)
        assert err.max() <= 1e-5, err
