"""ssqueeze_rs_amd -- MI355X-native synchrosqueezing engine, drop-in for `ssqueeze._rs`
(jesusdpa1/ssqueeze_rs).  Mirrors the reference's package shim (src/ssqueeze/__init__.py:2-27):

    from ssqueeze_rs_amd import _rs
    Tx, ssq_freqs = _rs.ssq_stft(x, window, n_fft=1024, hop_len=256, fs=fs)

Unlike the reference shim there is no dummy fallback: without libssq_hip.so (hand-written HIP,
gfx950) every call raises.
"""
from . import _rs  # noqa: F401

__all__ = ["_rs"]


def main():
    """src/ssqueeze/__init__.py:26-27."""
    print(_rs.hello_from_bin())
