"""Chunked-overlap multi-channel front end: the MI355X counterpart of the Dask harness the reference's users
call `_rs.*` through.

Reference behaviour being replaced (tests/stft_ssq_test.py:163-283, tests/stft_test.py:163-270,
tests/ssq_cwt_test.py:66-195, tests/cwt_test.py:69-195): a `(samples, channels)` array in chunks of 1 000 000
samples (`chunks=(1_000_000, -1)`, stft_ssq_test.py:302) goes through
`data.map_overlap(process_chunk, depth={-2: depth}, boundary=..., new_axis=-3)`: every chunk is extended by `depth`
samples on both sides -- the neighbours' samples inside the array, the boundary rule at its two ends --
`process_chunk` loops over the channels in Python calling `_rs.*` once per channel on the extended chunk
(:230-248) and returns `np.transpose(stacked, (1, 2, 0))` = `(freq, frames, channels)` (:265-267); the chunk results
are concatenated along the frames axis.  depth = n_fft for the STFT family (:216), max(1024, samples // 10) for the
CWT family (ssq_cwt_test.py:118-120).

Here the channels are uploaded once, the halos live on the device and the stacking is a device pass.  STFT family: all
(channel, chunk) windows run as ONE strided batch through the plan (no Python loop over channels, no padded copies).
CWT family: the windows are contiguous views of the same device array too, but every (channel, chunk) is its own
batch-1 plan exec (a Python loop of launches, no host copies in between):
`include/ssq_hip.h`: ssq_chunk_halo_fill, ssq_stft_plan_exec_strided, ssq_chunks_relayout.

`trim`: what is kept of every extended chunk's output --
  "none"  everything `process_chunk` returns (the reference function's own output, chunk by chunk);
  "halo"  only the columns whose frame start / time sample lies inside the chunk proper: with depth AND chunk multiples
          of the hop (the default chunk of 1 000 000 is not one of 256: pass e.g. chunk=2**20) the concatenation is then
          the whole-signal frame grid, and away from the two array ends it is bitwise the whole-signal transform (the
          seams vanish; Dask's own trimming is meant to do this).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import numpy as np

from . import _lib

CHUNK_DEFAULT = 1_000_000            # tests/stft_ssq_test.py:302


def chunk_plan(samples: int, chunk: int) -> List[Tuple[int, int]]:
    """Dask's regular chunking of the samples axis: [(start, length)], the last one may be shorter."""
    if samples <= 0 or chunk <= 0:
        raise ValueError("samples and chunk must be positive")
    return [(s, min(chunk, samples - s)) for s in range(0, samples, chunk)]


def overlap_extend(x: np.ndarray, depth: int, boundary: str = "reflect") -> np.ndarray:
    """NumPy statement of `dask.array.overlap` boundaries along axis 0 of `(samples, ...)`: "reflect" mirrors the
    array INCLUDING its edge sample (position -m holds x[m-1]); anything else pads zeros.  Host-side twin of
    ssq_chunk_halo_fill, used by the tests."""
    if depth == 0:
        return x.copy()
    if depth > x.shape[0]:
        raise ValueError("overlap depth larger than the array")
    if boundary == "reflect":
        left, right = x[depth - 1::-1], x[:-depth - 1:-1]
    else:
        left = right = np.zeros((depth,) + x.shape[1:], dtype=x.dtype)
    return np.concatenate([left, x, right], axis=0)


def _window(name: str, n_fft: int) -> np.ndarray:
    """tests/stft_ssq_test.py:204-213."""
    return {"hann": np.hanning, "hamming": np.hamming, "blackman": np.blackman}.get(name, np.hanning)(n_fft)


def _as_channels(data, dtype) -> np.ndarray:
    """(samples,) or (samples, channels) -> C-contiguous [channels, samples] of `dtype` (stft_ssq_test.py:199-203)."""
    a = np.asarray(data)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    elif a.ndim != 2:
        raise ValueError(f"Expected 1D or 2D input, got shape {a.shape}")
    return np.ascontiguousarray(a.T, dtype=dtype)


class _DeviceChannels:
    """[channels][depth + samples + depth] on the device, array-end halos filled there."""

    def __init__(self, xc: np.ndarray, depth: int, boundary: str):
        self.lib = _lib.load()
        _lib.require_gpu()
        self.channels, self.samples = xc.shape
        self.depth = int(depth)
        self.pitch = self.samples + 2 * self.depth
        self.dtype = xc.dtype
        self.code = _lib.SSQ_F32 if xc.dtype == np.float32 else _lib.SSQ_F64
        self.esz = xc.dtype.itemsize
        self.d = C.c_void_p()
        _lib.check(self.lib.ssq_dev_malloc(C.byref(self.d), self.channels * self.pitch * self.esz))
        for ch in range(self.channels):
            _lib.check(self.lib.ssq_memcpy_h2d(C.c_void_p(self.d.value + (ch * self.pitch + self.depth) * self.esz),
                                               xc[ch].ctypes.data_as(C.c_void_p), self.samples * self.esz, None))
        _lib.check(self.lib.ssq_chunk_halo_fill(self.code, self.d, self.channels, self.samples, self.depth,
                                                0 if boundary == "reflect" else 1, None))

    def window_ptr(self, start: int) -> C.c_void_p:
        """Extended chunk starting at array sample `start` (its halo starts `depth` earlier = buffer index start)."""
        return C.c_void_p(self.d.value + start * self.esz)

    def close(self):
        if self.d:
            self.lib.ssq_dev_free(self.d)
            self.d = None


def _kept(trim: str, depth: int, length: int, step: int, n_ext: int) -> Tuple[int, int]:
    """(first kept column, count) of an extended chunk's output whose column c sits at extended sample c * step."""
    if trim == "none":
        return 0, n_ext
    if trim != "halo":
        raise ValueError("trim must be 'none' or 'halo'")
    f0 = -(-depth // step)                                   # first column at or after the chunk's first sample
    f1 = -(-(depth + length) // step)                        # first column at or after the halo behind it
    return f0, max(0, min(f1, n_ext) - f0)


def _stft_family(data, out_kind, fs, n_fft, hop, window, squeezing, chunk, depth, trim, dtype, temp_bytes):
    lib = _lib.load()
    xc = _as_channels(data, dtype)
    Cn, S = xc.shape
    depth = n_fft if depth is None else int(depth)           # stft_ssq_test.py:216
    plan_chunks = chunk_plan(S, chunk)
    dev = _DeviceChannels(xc, depth, "reflect")              # boundary="reflect" (:277)
    cd = np.complex64 if dev.code == _lib.SSQ_F32 else np.complex128
    K = n_fft // 2 + 1
    groups = {}                                              # chunk length -> [chunk index]
    for j, (_, L) in enumerate(plan_chunks):
        groups.setdefault(L, []).append(j)
    shapes = {L: ((L + 2 * depth - 1) // hop + 1) for L in groups}
    kept = {L: _kept(trim, depth, L, hop, shapes[L]) for L in groups}
    out_cols = sum(kept[L][1] for _, L in plan_chunks)
    col_base = np.cumsum([0] + [kept[L][1] for _, L in plan_chunks])
    d_final, d_tmp = C.c_void_p(), C.c_void_p()
    win = np.ascontiguousarray(window, dtype=np.float64)
    plans = []
    try:
        _lib.check(lib.ssq_dev_malloc(C.byref(d_final), max(1, K * out_cols * Cn) * 2 * dev.esz))
        for L, idx in groups.items():
            n_ext, F = L + 2 * depth, shapes[L]
            plan = C.c_void_p()
            _lib.check(lib.ssq_stft_plan_create(C.byref(plan), dev.code, n_ext, win.ctypes.data_as(C.c_void_p), n_fft,
                                                hop, float(fs), 0, _lib.SQUEEZE.get(squeezing, 0), -1.0, 0))
            plans.append(plan)
            per = Cn * K * F * 2 * dev.esz                   # temp bytes of one chunk over all channels
            slab = max(1, min(len(idx), temp_bytes // max(per, 1)))
            ws = int(lib.ssq_stft_plan_workspace_bytes(plan, Cn * slab, out_kind))
            d_ws = C.c_void_p()
            try:                                             # (allocations inside: a failing one must not leak the other)
                _lib.check(lib.ssq_dev_malloc(C.byref(d_tmp), slab * per))
                _lib.check(lib.ssq_dev_malloc(C.byref(d_ws), max(ws, 16)))
                for a in range(0, len(idx), slab):           # consecutive equal-length chunks: one strided batch
                    js = idx[a:a + slab]
                    J = len(js)
                    _lib.check(lib.ssq_stft_plan_exec_strided(plan, out_kind, dev.window_ptr(plan_chunks[js[0]][0]), Cn, J,
                                                              dev.pitch, chunk, d_tmp, d_ws, ws, None))
                    f0, nf = kept[L]
                    _lib.check(lib.ssq_chunks_relayout(dev.code, d_tmp, Cn, J, K, F, f0, nf, d_final, out_cols,
                                                       int(col_base[js[0]]), Cn, 0, None))
            finally:
                lib.ssq_dev_free(d_tmp)
                lib.ssq_dev_free(d_ws)
                d_tmp = C.c_void_p()
        out = np.empty((K, out_cols, Cn), dtype=cd)
        _lib.check(lib.ssq_device_sync())
        if out.size:
            _lib.check(lib.ssq_memcpy_d2h(out.ctypes.data_as(C.c_void_p), d_final, out.nbytes, None))
            _lib.check(lib.ssq_device_sync())
        return out
    finally:
        for p in plans:
            lib.ssq_stft_plan_destroy(p)
        if d_final:
            lib.ssq_dev_free(d_final)
        dev.close()


def process_stft_ssq(data, fs: float = None, n_fft: int = 1024, hop_length: int = 256, window_name: str = "hann",
                     squeezing: str = "sum", *, chunk: int = CHUNK_DEFAULT, depth: Optional[int] = None,
                     trim: str = "none", dtype=np.float64, temp_bytes: int = 2 << 30) -> np.ndarray:
    """tests/stft_ssq_test.py:163-283 for a NumPy `(samples, channels)` array: `(freq_bins, frames, channels)`
    of `_rs.ssq_stft(channel, window, n_fft=n_fft, win_len=n_fft, hop_len=hop_length, fs=fs, padtype="reflect",
    squeezing=squeezing)` over the extended chunks."""
    if fs is None:
        raise ValueError("Sampling frequency (fs) must be provided")             # :197-198
    return _stft_family(data, _lib.OUT_TX, fs, n_fft, hop_length, _window(window_name, n_fft), squeezing, chunk,
                        depth, trim, dtype, temp_bytes)


def process_stft(data, fs: float = None, n_fft: int = 1024, hop_length: int = 256, window_name: str = "hann", *,
                 chunk: int = CHUNK_DEFAULT, depth: Optional[int] = None, trim: str = "none", dtype=np.float64,
                 temp_bytes: int = 2 << 30) -> np.ndarray:
    """tests/stft_test.py:163-270: `_rs.stft(channel, n_fft, hop_length, window, "reflect")` over the extended
    chunks, `(freq_bins, frames, channels)`."""
    return _stft_family(data, _lib.OUT_SX, 1.0 if fs is None else fs, n_fft, hop_length, _window(window_name, n_fft),
                        "sum", chunk, depth, trim, dtype, temp_bytes)


def _cwt_family(data, ssq, fs, wavelet, scales, nv, padtype, squeezing, maprange, derivative, chunk, depth, trim,
                dtype, temp_bytes, ssq_freqs, flipud, gamma):
    lib = _lib.load()
    xc = _as_channels(data, dtype)
    Cn, S = xc.shape
    depth = max(1024, S // 10) if depth is None else int(depth)                   # ssq_cwt_test.py:118-120
    plan_chunks = chunk_plan(S, chunk)
    dev = _DeviceChannels(xc, depth, padtype)                                     # boundary=padtype (:190)
    cd = np.complex64 if dev.code == _lib.SSQ_F32 else np.complex128
    dt = 1.0 / float(fs) if fs is not None else 1.0
    groups = {}
    for j, (_, L) in enumerate(plan_chunks):
        groups.setdefault(L, []).append(j)
    kept = {L: _kept(trim, depth, L, 1, L + 2 * depth) for L in groups}
    out_cols = sum(kept[L][1] for _, L in plan_chunks)
    col_base = np.cumsum([0] + [kept[L][1] for _, L in plan_chunks])
    from ._rs import _scales_or_default
    n_out = 1 if ssq or not derivative else 2
    finals = [C.c_void_p() for _ in range(n_out)]
    plans = []
    freqs = None
    sc_used = None
    try:
        for L, idx in groups.items():
            n_ext = L + 2 * depth
            sc = _scales_or_default(scales, n_ext, nv, False)                     # per extended chunk, as the reference
            na = sc.shape[0]
            if sc_used is None:
                sc_used = sc
                for f in finals:
                    _lib.check(lib.ssq_dev_malloc(C.byref(f), max(1, na * out_cols * Cn) * 2 * dev.esz))
            elif na != sc_used.shape[0]:
                raise ValueError("automatic scales differ between the full and the ragged chunk: pass `scales`")
            plan = C.c_void_p()
            _lib.check(lib.ssq_cwt_plan_create(C.byref(plan), dev.code, n_ext, _lib.WAVELET.get(wavelet, 0),
                                               sc.ctypes.data_as(C.c_void_p), na, dt, _lib.PAD.get(padtype, 0)))
            plans.append(plan)
            wsb = int(lib.ssq_cwt_plan_workspace_bytes(plan, 1))
            per = Cn * na * n_ext * 2 * dev.esz
            slab = max(1, min(len(idx), temp_bytes // max(per, 1)))
            d_ws = C.c_void_p()
            tmps = [C.c_void_p() for _ in range(n_out)]
            try:                                             # (allocations inside: a failing one must not leak the others)
                _lib.check(lib.ssq_dev_malloc(C.byref(d_ws), max(wsb, 16)))
                for t in tmps:
                    _lib.check(lib.ssq_dev_malloc(C.byref(t), slab * per))
                if ssq and freqs is None:
                    freqs = np.empty(na, dtype=np.float64)
                    _lib.check(lib.ssq_cwt_ssq_freqs(sc.ctypes.data_as(C.c_void_p), na, n_ext, dt,
                                                     1 if maprange == "maximal" else 0, 1 if ssq_freqs == "linear" else 0,
                                                     freqs.ctypes.data_as(C.c_void_p)))
                for a in range(0, len(idx), slab):
                    js = idx[a:a + slab]
                    J = len(js)
                    for ch in range(Cn):
                        for jj, j in enumerate(js):
                            src = C.c_void_p(dev.d.value + (ch * dev.pitch + plan_chunks[j][0]) * dev.esz)
                            off = ((ch * J + jj) * na * n_ext) * 2 * dev.esz
                            if ssq:
                                _lib.check(lib.ssq_cwt_plan_exec_ssq(
                                    plan, src, 1, 1 if ssq_freqs == "linear" else 0, 1 if maprange == "maximal" else 0,
                                    _lib.SQUEEZE.get(squeezing, 0), int(bool(flipud)), -1.0 if gamma is None else float(gamma),
                                    C.c_void_p(tmps[0].value + off), None, None, None, d_ws, wsb, None))
                            else:
                                _lib.check(lib.ssq_cwt_plan_exec_cwt(
                                    plan, src, 1, 1, 0, C.c_void_p(tmps[0].value + off),
                                    C.c_void_p(tmps[1].value + off) if n_out == 2 else None, d_ws, wsb, None))
                    f0, nf = kept[L]
                    for t, f in zip(tmps, finals):
                        _lib.check(lib.ssq_chunks_relayout(dev.code, t, Cn, J, na, n_ext, f0, nf, f, out_cols,
                                                           int(col_base[js[0]]), Cn, 0, None))
            finally:
                lib.ssq_dev_free(d_ws)
                for t in tmps:
                    lib.ssq_dev_free(t)
        na = sc_used.shape[0]
        outs = []
        _lib.check(lib.ssq_device_sync())
        for f in finals:
            o = np.empty((na, out_cols, Cn), dtype=cd)
            if o.size:
                _lib.check(lib.ssq_memcpy_d2h(o.ctypes.data_as(C.c_void_p), f, o.nbytes, None))
            outs.append(o)
        _lib.check(lib.ssq_device_sync())
        return outs, sc_used, freqs
    finally:
        for p in plans:
            lib.ssq_cwt_plan_destroy(p)
        for f in finals:
            if f:
                lib.ssq_dev_free(f)
        dev.close()


def process_ssq_cwt(data, fs: float = None, wavelet: str = "gmw", scales=None, nv: int = 32, padtype: str = "reflect",
                    squeezing: str = "sum", maprange: str = "peak", *, ssq_freqs: Optional[str] = None,
                    flipud: bool = True, gamma: Optional[float] = None, chunk: int = CHUNK_DEFAULT,
                    depth: Optional[int] = None, trim: str = "none", dtype=np.float64, temp_bytes: int = 8 << 30):
    """tests/ssq_cwt_test.py:66-195: `(Tx (frequencies, time, channels), ssq_freqs)` of `_rs.ssq_cwt` over the
    extended chunks (depth = max(1024, samples // 10), boundary = padtype)."""
    if fs is None:
        raise ValueError("Sampling frequency (fs) must be provided")             # :104-105
    outs, _, freqs = _cwt_family(data, True, fs, wavelet, scales, nv, padtype, squeezing, maprange, False, chunk, depth,
                                 trim, dtype, temp_bytes, ssq_freqs, flipud, gamma)
    return outs[0], freqs


def process_cwt(data, fs: float = None, wavelet: str = "gmw", scales=None, nv: int = 32, derivative: bool = True,
                padtype: str = "reflect", *, chunk: int = CHUNK_DEFAULT, depth: Optional[int] = None,
                trim: str = "none", dtype=np.float64, temp_bytes: int = 8 << 30):
    """tests/cwt_test.py:69-195: `(Wx (scales, time, channels), scales, dWx or None)` of `_rs.cwt` over the
    extended chunks."""
    if fs is None:
        raise ValueError("Sampling frequency (fs) must be provided")
    outs, sc, _ = _cwt_family(data, False, fs, wavelet, scales, nv, padtype, "sum", "peak", derivative, chunk, depth,
                              trim, dtype, temp_bytes, None, True, None)
    return outs[0], sc, (outs[1] if derivative else None)
