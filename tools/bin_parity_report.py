"""Measured fp32 bin parity of ssq_stft / ssq_cwt against (a) the reference's scan on the kernel's own w and (b) the
fp64 oracle end to end; writes gpurun_out/r03_bin_parity.json (copied to profiles/).  VERDICT r2 item 3."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ssq_oracle as o  # noqa: E402
from ssqueeze_rs_amd import _rs  # noqa: E402
from tests.helpers.binrule import TIE_REL  # noqa: E402

out = {"tie_rel": TIE_REL, "stft": [], "cwt": []}
cases = [(8192 + 2 * 256, 256, 64, 6), (8192 + 2048, 1024, 256, 6), (8192 + 256, 128, 32, 6), (8192 + 8192, 4096, 1024, 6),
         (1 << 18, 1024, 256, 0), (1 << 20, 1024, 256, 0)]
for N, n_fft, hop, seed in cases:
    x = o.synth_signal(N, seed, np.float32)
    win = np.hanning(n_fft)
    Tx, f, dbg = _rs.ssq_stft(x, win, n_fft=n_fft, hop_len=hop, fs=1.0, _debug=True)
    _, _, im = o.ssq_stft(x.astype(np.float64), win, n_fft=n_fft, hop_len=hop, fs=1.0, return_intermediates=True)
    keep = dbg["k"] >= 0
    w64 = dbg["w"].astype(np.float64)[keep]
    kref = o.nearest_bin_first_min(w64, f)
    diff = dbg["k"][keep] != kref
    tq = w64[diff] / (f[1] - f[0])
    dist = np.abs(tq - np.floor(tq) - 0.5) / np.maximum(1.0, np.abs(tq))
    smax = np.abs(im["Sx"]).max()
    keep_o = ~np.isinf(im["w"])
    both = keep_o & keep & (np.abs(im["Sx"]) > 1e-3 * smax)
    both_all = keep_o & keep
    rec = {"N": N, "n_fft": n_fft, "hop": hop, "kept_bins": int(keep.sum()),
           "differ_from_reference_scan_on_own_w": int(diff.sum()),
           "worst_relative_tie_distance_of_those": float(dist.max()) if diff.any() else 0.0,
           "end_to_end_rate_strong_bins": float((dbg["k"][both] != im["k"][both]).mean()),
           "end_to_end_rate_all_kept": float((dbg["k"][both_all] != im["k"][both_all]).mean()),
           "end_to_end_max_abs_dk_strong": int(np.abs(dbg["k"][both] - im["k"][both]).max()),
           "sx_rel_err": float(np.abs(dbg["Sx"] - im["Sx"]).max() / smax),
           "dsx_rel_err": float(np.abs(dbg["dSx"] - im["dSx"]).max() / np.abs(im["dSx"]).max())}
    # w through the reference formula on the kernel's own Sx/dSx
    Sg, dSg = dbg["Sx"].astype(np.complex128), dbg["dSx"].astype(np.complex128)
    w_m = o.phase_stft(Sg, dSg, im["Sfs"], o.DEFAULT_GAMMA)
    strong = keep & (np.abs(Sg) > 1e-3 * smax) & np.isfinite(w_m)
    rec["w_vs_formula_max_over_nyquist"] = float(np.abs(dbg["w"][strong] - w_m[strong]).max() / 0.5)
    out["stft"].append(rec)
    print(rec, flush=True)

for N, wavelet, nv in [(4096, "morlet", 32), (4096, "gmw", 32), (1 << 16, "morlet", 16), (1 << 16, "gmw", 16)]:
    x = o.synth_signal(N, 3, np.float32)
    kw = dict(wavelet=wavelet, nv=nv)
    Tx, f, dbg = _rs.ssq_cwt(x, _debug=True, **kw)
    Tx_o, f_o, im = o.ssq_cwt(x.astype(np.float64), return_intermediates=True, **kw)
    w_g = dbg["w"].astype(np.float64)
    b, valid, is_log = o.cwt_bins(w_g, f_o)
    na = Tx.shape[0]
    k_model = np.where(valid, na - 1 - b, -1)
    mism = k_model != dbg["k"]
    with np.errstate(all="ignore"):
        if is_log:
            lmin = np.log2(f_o[0]); lstep = (np.log2(f_o[-1]) - lmin) / (na - 1)
            v = (np.log2(w_g[mism]) - lmin) / lstep
        else:
            lstep = (f_o[-1] - f_o[0]) / (na - 1)
            v = (w_g[mism] - f_o[0]) / lstep
    d = np.abs(np.abs(v - np.trunc(v)) - 0.5)
    rec = {"N": N, "wavelet": wavelet, "nv": nv, "na": int(na), "is_log": bool(is_log), "elements": int(mism.size),
           "differ_from_reference_binning_on_own_w": int(mism.sum()), "rate": float(mism.mean()),
           "worst_tie_distance_bins": float(np.nanmax(d)) if mism.any() else 0.0}
    out["cwt"].append(rec)
    print(rec, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r03_bin_parity.json", "w"), indent=1)
