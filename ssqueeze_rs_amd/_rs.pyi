# Type stubs of ssqueeze_rs_amd._rs -- the MI355X drop-in for the reference's `_rs` extension module (same names,
# keyword arguments, defaults, return arity and dtypes; see INTEGRATION.md for the C-ABI underneath).
# Every function also accepts a 2-D `[batch, samples]` array and then returns arrays with a leading batch axis
# (the reference is called per channel in a Python loop, tests/stft_ssq_test.py:230).
from typing import Any, Dict, Optional, Tuple, Union

import numpy as np
from numpy.typing import NDArray

Real = Union[NDArray[np.float32], NDArray[np.float64]]
Cplx = Union[NDArray[np.complex64], NDArray[np.complex128]]

class PanicException(BaseException):
    """Raised where the Rust reference panics (empty input, zero hop, ...)."""

def hello_from_bin() -> str: ...

# ---- STFT family (rust/src/spectral/stft.rs, ssq_stft.rs) ----
def stft(x: Real, n_fft: int, hop_length: int, window: NDArray[np.float64], padtype: str) -> Tuple[Cplx, NDArray[np.float64]]:
    """(Sx [n_fft // 2 + 1, n_frames], freqs in cycles / sample)."""

def ssq_stft(
    x: Real,
    window: NDArray[np.float64],
    n_fft: Optional[int] = None,
    win_len: Optional[int] = None,
    hop_len: int = 1,
    fs: float = 1.0,
    padtype: str = "reflect",
    squeezing: str = "sum",
    gamma: Optional[float] = None,
    _debug: bool = False,
) -> Union[Tuple[Cplx, NDArray[np.float64]], Tuple[Cplx, NDArray[np.float64], Dict[str, Any]]]:
    """(Tx [n_fft // 2 + 1, n_frames], ssq_freqs); `_debug=True` adds a dict with Sx, dSx, w, k of the same kernel."""

# ---- CWT family (rust/src/spectral/cwt.rs, cwt_simd.rs, ssq_cwt.rs) ----
def cwt(
    x: Real,
    wavelet: str = "gmw",
    scales: Optional[NDArray[np.float64]] = None,
    fs: Optional[float] = None,
    t: Optional[NDArray[np.float64]] = None,
    nv: int = 32,
    l1_norm: bool = True,
    derivative: bool = False,
    padtype: str = "reflect",
    rpadded: bool = False,
    vectorized: bool = True,
    patience: int = 0,
) -> Tuple[Cplx, NDArray[np.float64], Optional[Cplx]]:
    """Always a 3-tuple (Wx [n_scales, N or padded length], scales, dWx or None)."""

def cwt_simd(
    x: Real,
    wavelet: str = "gmw",
    scales: Optional[NDArray[np.float64]] = None,
    fs: Optional[float] = None,
    t: Optional[NDArray[np.float64]] = None,
    nv: int = 32,
    l1_norm: bool = True,
    derivative: bool = False,
    padtype: str = "reflect",
    rpadded: bool = False,
    vectorized: bool = True,
    patience: int = 0,
) -> Tuple[Cplx, NDArray[np.float64], Optional[Cplx]]: ...

def ssq_cwt(
    x: Real,
    wavelet: str = "gmw",
    scales: Optional[NDArray[np.float64]] = None,
    fs: Optional[float] = None,
    t: Optional[NDArray[np.float64]] = None,
    ssq_freqs: Optional[str] = None,
    nv: int = 32,
    padtype: str = "reflect",
    squeezing: str = "sum",
    maprange: str = "peak",
    difftype: str = "trig",
    gamma: Optional[float] = None,
    vectorized: bool = True,
    flipud: bool = True,
    _debug: bool = False,
) -> Union[Tuple[Cplx, NDArray[np.float64]], Tuple[Cplx, NDArray[np.float64], Dict[str, Any]]]:
    """(Tx [n_scales, N], ssq_freqs); `_debug=True` adds a dict with Wx, dWx, w, k."""

def icwt(
    Wx: Cplx,
    wavelet: str = "gmw",
    scales: Optional[NDArray[np.float64]] = None,
    nv: Optional[int] = None,
    one_int: bool = True,
    x_len: Optional[int] = None,
    x_mean: float = 0.0,
    padtype: str = "reflect",
    rpadded: bool = False,
    l1_norm: bool = True,
) -> NDArray[np.float64]: ...

# ---- wavelet helpers (rust/src/wavelets/morlet.rs, gmw.rs): all return complex128 (frequency-domain ones with imag = 0) ----
def morlet(w: NDArray[np.float64], mu: float = 6.0, dtype: str = "float64") -> NDArray[np.complex128]: ...
def morlet_freq(n: int = 1024, scale: float = 1.0, mu: float = 6.0, dtype: str = "float64") -> NDArray[np.complex128]: ...
def morlet_time(n: int = 1024, scale: float = 1.0, mu: float = 6.0, dtype: str = "float64") -> NDArray[np.complex128]: ...
def gmw(w: NDArray[np.float64], gamma: float = 3.0, beta: float = 60.0, norm: str = "bandpass", order: int = 0,
        dtype: str = "float64") -> NDArray[np.complex128]: ...
def gmw_freq(n: int = 1024, scale: float = 1.0, gamma: float = 3.0, beta: float = 60.0, norm: str = "bandpass",
             order: int = 0, dtype: str = "float64") -> NDArray[np.complex128]: ...
def gmw_time(n: int = 1024, scale: float = 1.0, gamma: float = 3.0, beta: float = 60.0, norm: str = "bandpass",
             order: int = 0, dtype: str = "float64") -> NDArray[np.complex128]: ...
def gmw_center_frequency(gamma: float = 3.0, beta: float = 60.0, kind: str = "peak") -> float: ...

# Private switches of this mirror (not in the reference): `_debug=True` on ssq_stft / ssq_cwt returns the kernels' own
# intermediates; `_upstream=True` on stft / ssq_stft / cwt / ssq_cwt runs the numerics of the vendored upstream
# ssqueezepy.  `ssqueeze_rs_amd.upstream` mirrors upstream's own signatures and adds the inverses
# istft / issq_stft / icwt / issq_cwt.

# fp32 `cwt` (with derivative) / `ssq_cwt` run the scales whose wavelet is short in time on TIME TILES by default
# (csrc/cwt_os.hip): the plain and decimated tiles evaluate the same convolution with the wavelet's time response cut at
# the tile halo -- an approximation below 1e-5 of each row's maximum (tests pin the eligibility limits); the full-circle
# blocks and analytic-input tiles are exact.  SSQ_CWT_OS=0 / SSQ_CWT_OS_STORE=0 select the exact frequency-domain path;
# float64 always takes it.
