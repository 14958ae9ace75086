"""fp32 bin parity under the REFERENCE's rule.

The fp32 kernels compute the bin as k = ceil(fma(w, fl32(1/dw), -1/2)) -- one rounding of 1/dw and one of the fma,
so the argument of the ceil carries at most  u * 2^-24 (1/dw) + ulp(u)/2 <= u * 2^-23  of error, u = w/dw.  Given the
kernel's OWN fp32 `w`, the reference's first-minimum scan (ssq_stft.rs:280-289, `oracle.nearest_bin_first_min`) must
therefore give the same index except where w/dw lies within TIE_REL * max(1, w/dw) of a half-bin tie.
TIE_REL = 2^-22 (twice the bound above).  Everything else is index-exact."""
import numpy as np

from oracle import ssq_oracle as o

TIE_REL = 2.0 ** -22


def stft_bins_follow_reference_rule(k, w32, ssq_freqs, keep):
    """k, w32: the Tx kernel's own bins and phase transform (WKDBG hook); returns (n_checked, n_near_tie_exempted).
    Asserts index-exactness under ssq_stft.rs:280-289 outside the stated tie window."""
    w64 = np.asarray(w32, dtype=np.float64)[keep]
    kk = np.asarray(k)[keep]
    fin = np.isfinite(w64)                       # kept bins have finite w; NaN -> bin 0 by the scan
    k_ref = o.nearest_bin_first_min(w64, np.asarray(ssq_freqs, dtype=np.float64))
    diff = kk != k_ref
    if not diff.any():
        return int(kk.size), 0
    dw = float(ssq_freqs[1] - ssq_freqs[0])
    tq = w64[diff] / dw
    near = np.abs(tq - np.floor(tq) - 0.5) <= TIE_REL * np.maximum(1.0, np.abs(tq))
    assert fin[diff].all() and near.all(), (
        f"{int((~near).sum())} of {kk.size} fp32 bins differ from the reference scan away from a half-bin tie "
        f"(worst distance {np.abs(tq - np.floor(tq) - 0.5)[~near].max() if (~near).any() else 0:.3e} bins)")
    return int(kk.size), int(diff.sum())


def end_to_end_rate(k, k_oracle, both):
    """Share of bins (kept by both, above the |Sx| floor the caller chose) whose fp32 index differs from the fp64
    oracle's -- the figure SURVEY 8(c) asks to report."""
    return float((np.asarray(k)[both] != np.asarray(k_oracle)[both]).mean()) if both.any() else 0.0
