"""Import-name alias: the reference's callers write `from ssqueeze import _rs`
(/root/reference src/ssqueeze/__init__.py:2-3, README.md:74-79).  With this package on the path their import line
runs unchanged against the MI355X engine; everything lives in `ssqueeze_rs_amd`.  No dummy fallback
(src/ssqueeze/__init__.py:13-20 has one): without libssq_hip.so the import raises."""
import sys as _sys

from ssqueeze_rs_amd import _rs, main  # noqa: F401

_sys.modules[__name__ + "._rs"] = _rs      # `import ssqueeze._rs` and `from ssqueeze._rs import ssq_stft` work too

__all__ = ["_rs"]
