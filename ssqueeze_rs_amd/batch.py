"""Batched, device-resident front end over the plan API of libssq_hip.so.

The reference has no batch mechanism: its scripts loop over channels in Python and call `_rs.*` once
per channel per chunk (tests/stft_ssq_test.py:230-248).  Here a batch of independent signals is one
launch, stays in HBM between calls, and shards across the GPUs of a node by contiguous blocks with
no exchange during compute; the only collective is the optional final gather.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib


def shard_bounds(batch: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of a batch owned by `rank`: sizes differ by at most one and every
    signal belongs to exactly one rank."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class SsqStftBatch:
    """ssq_stft / stft of `[batch, N]` signals on the current device through one plan.
    Host arrays in, host arrays out; the plan, tables and device buffers persist across calls."""

    def __init__(self, n_signal: int, window: np.ndarray, n_fft: int, hop_len: int, fs: float = 1.0,
                 padtype: str = "reflect", squeezing: str = "sum", gamma: Optional[float] = None,
                 dtype=np.float32, max_batch: int = 1, device: Optional[int] = None):
        self.lib = _lib.load()
        _lib.require_gpu()
        if device is not None:
            _lib.check(self.lib.ssq_set_device(int(device)))
        self.dtype = np.dtype(dtype)
        self.code = _lib.SSQ_F32 if self.dtype == np.float32 else _lib.SSQ_F64
        self.cdtype = np.complex64 if self.code == _lib.SSQ_F32 else np.complex128
        self.N, self.n_fft, self.hop = int(n_signal), int(n_fft), int(hop_len)
        self.n_freqs, self.n_frames = self.n_fft // 2 + 1, (self.N - 1) // self.hop + 1
        win = np.ascontiguousarray(window, dtype=np.float64)
        sized = np.empty(self.n_fft, dtype=np.float64)
        _lib.check(self.lib.ssq_size_window(win.ctypes.data_as(C.c_void_p), win.shape[0], self.n_fft,
                                            sized.ctypes.data_as(C.c_void_p)))
        self.plan = C.c_void_p()
        _lib.check(self.lib.ssq_stft_plan_create(C.byref(self.plan), self.code, self.N,
                                                 sized.ctypes.data_as(C.c_void_p), self.n_fft, self.hop, float(fs),
                                                 _lib.PAD.get(padtype, 0), _lib.SQUEEZE.get(squeezing, 0),
                                                 -1.0 if gamma is None else float(gamma), 0))
        self.ssq_freqs = (np.arange(self.n_freqs, dtype=np.float64) * 0.5 * fs) / (float(self.n_freqs) - 1.0)
        self.max_batch = int(max_batch)
        self.d_x, self.d_out, self.d_ws = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self.ws_bytes = int(self.lib.ssq_stft_plan_workspace_bytes(self.plan, self.max_batch, _lib.OUT_TX))
        _lib.check(self.lib.ssq_dev_malloc(C.byref(self.d_x), self.max_batch * self.N * self.dtype.itemsize))
        _lib.check(self.lib.ssq_dev_malloc(C.byref(self.d_out),
                                           self.max_batch * self.n_freqs * self.n_frames * 2 * self.dtype.itemsize))
        _lib.check(self.lib.ssq_dev_malloc(C.byref(self.d_ws), max(self.ws_bytes, 16)))

    def run(self, x: np.ndarray, out_kind: int = _lib.OUT_TX) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=self.dtype)
        if x.ndim != 2 or x.shape[1] != self.N or x.shape[0] > self.max_batch:
            raise ValueError("x must be [batch <= max_batch, N]")
        b = x.shape[0]
        out = np.empty((b, self.n_freqs, self.n_frames), dtype=self.cdtype)
        if b == 0:
            return out
        _lib.check(self.lib.ssq_memcpy_h2d(self.d_x, x.ctypes.data_as(C.c_void_p), x.nbytes, None))
        _lib.check(self.lib.ssq_stft_plan_exec(self.plan, out_kind, self.d_x, b, self.d_out, self.d_ws,
                                               self.ws_bytes, None))
        _lib.check(self.lib.ssq_device_sync())
        _lib.check(self.lib.ssq_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.d_out, out.nbytes, None))
        _lib.check(self.lib.ssq_device_sync())
        return out

    def close(self):
        if getattr(self, "plan", None):
            self.lib.ssq_dev_free(self.d_x)
            self.lib.ssq_dev_free(self.d_out)
            self.lib.ssq_dev_free(self.d_ws)
            self.lib.ssq_stft_plan_destroy(self.plan)
            self.plan = None

    __del__ = close


def gather_shards(local: np.ndarray, counts, group=None) -> np.ndarray:
    """Optional final gather of per-rank result shards `[n_local, ...]` (all ranks get the whole batch).
    torch.distributed only: backend "nccl" (= RCCL over xGMI) with device tensors on a GPU node, "gloo"
    on CPU.  Never part of the timed hot path."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return local
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    is_cplx = np.iscomplexobj(local)
    flat = np.ascontiguousarray(local).view(np.float32 if local.dtype in (np.complex64, np.float32) else np.float64)
    per = int(np.prod(flat.shape[1:]))
    nmax = max(counts)
    buf = torch.zeros(nmax * per, dtype=torch.from_numpy(flat[:0]).dtype, device=dev)
    buf[: flat.size] = torch.from_numpy(flat.reshape(-1)).to(dev)
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    parts = [o.cpu().numpy()[: c * per].reshape((c,) + flat.shape[1:]) for o, c in zip(outs, counts)]
    full = np.concatenate(parts, axis=0)
    return full.view(local.dtype) if is_cplx else full
