"""Multi-process (world_size 2, gloo, CPU) test of the N>1 path: contiguous batch sharding with no
data-path collective, the max-over-ranks timing reduction bench.py uses, and the optional final gather."""
import os
import socket
import sys

import numpy as np
import pytest

from ssqueeze_rs_amd.batch import shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_partition():
    for batch in (0, 1, 5, 8, 256, 257):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = shard_bounds(batch, world, r)
                assert 0 <= lo <= hi <= batch
                seen += list(range(lo, hi))
            assert seen == list(range(batch))
            sizes = [shard_bounds(batch, world, r)[1] - shard_bounds(batch, world, r)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(256, 8, 3) == (96, 128)
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from ssqueeze_rs_amd.batch import gather_shards, shard_bounds as sb
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        batch, rows, cols = 5, 3, 4
        lo, hi = sb(batch, world, rank)
        counts = [sb(batch, world, r)[1] - sb(batch, world, r)[0] for r in range(world)]
        # stand-in for the per-rank hot path: shard b of the batch yields the value b everywhere
        local = np.stack([np.full((rows, cols), b + 1j * (b + 0.5), dtype=np.complex64) for b in range(lo, hi)]) \
            if hi > lo else np.zeros((0, rows, cols), np.complex64)
        full = gather_shards(local, counts)
        # bench.py's timing reduction: MAX over ranks
        t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = full.shape == (batch, rows, cols) and all(
            np.all(full[b] == np.complex64(b + 1j * (b + 0.5))) for b in range(batch))
        q.put((rank, bool(ok), float(t.item())))
    finally:
        dist.destroy_process_group()


def test_world2_gloo_shard_gather_and_timing_reduce():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert all(abs(r[2] - 0.2) < 1e-12 for r in res)
