#!/usr/bin/env python3
"""bench.py -- headline benchmark: TF-bins/s of ssq_stft (n_fft=1024, hop=256, fp32) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the fused ssq_stft hot path over one batch of B synthetic signals of
2^20 samples resident in HBM (default B = 256 per GPU: BASELINE.json's target workload
"batch=256 x 2^20-sample signals on 1xMI355X"; with N GPUs every rank processes its own B
signals -> weak scaling, no data-path collective).  Rank 0 prints ONE JSON line.

torch is plumbing only (process group, barrier, torch.cuda.synchronize); device memory, the
stream, the HIP events and the kernels all come from libssq_hip.so through its C-ABI.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def measured_traffic(B, log2n, n_fft, hop):
    """HBM bytes per launch from the committed PMC run of this kernel (profiles/r01_traffic.json: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 correction applied).  Traffic is per signal, so
    other batch sizes scale it; other shapes have no measurement -> None."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
            t = json.load(f)
        w = t["workload"]
        if (w["log2n"], w["n_fft"], w["hop"]) != (log2n, n_fft, hop):
            return None
        return int(t["bytes_per_signal"] * B)
    except (OSError, KeyError, ValueError):
        return None


def synth_batch(n_distinct, N):
    from ssqueeze_rs_amd.synth import synth_signal   # synthetic workload generator
    return np.stack([synth_signal(N, b, np.float32) for b in range(n_distinct)])


def cpu_baseline(N, n_fft, hop, budget_s=20.0):
    """The oracle's C restatement of the reference CPU path (oracle/ssq_ref.c, kind "port"), timed on
    this box's host cores on a bounded sample of the same workload."""
    from oracle import ref_c                         # the only use of oracle/ here: the timed CPU baseline
    from ssqueeze_rs_amd.synth import synth_signal
    win = np.hanning(n_fft)
    cores = ref_c.num_threads()
    bins = (n_fft // 2 + 1) * ((N - 1) // hop + 1)
    out = {}
    for mode, name in ((0, "faithful"), (1, "optimized")):
        done, t_total = 0, 0.0
        while done < 64 and t_total < budget_s / 2:
            x = synth_signal(N, done, np.float64)
            t0 = time.perf_counter()
            ref_c.ssq_stft(x, win, n_fft, hop, fs=1.0, mode=mode)
            t_total += time.perf_counter() - t0
            done += 1
        out[name] = (bins * done / t_total, done, t_total)
    v, done, tt = out["faithful"]
    return {
        "value": v, "unit": "TF-bins/s", "cores": cores, "kind": "port",
        "sample": f"{done} signal(s) x 2^{int(np.log2(N))} samples, fp64, reference-faithful structure "
                  f"(2 FFTs/frame over {cores} OpenMP threads, serial phase, serial linear-scan reassignment), "
                  f"{tt:.1f} s",
        "optimized_value": out["optimized"][0],
        "optimized_sample": f"{out['optimized'][1]} signal(s), arithmetic binning + column-parallel reassignment, "
                            f"{out['optimized'][2]:.1f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="signals per GPU")
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--n-fft", type=int, default=1024)
    ap.add_argument("--hop", type=int, default=256)
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic signals (tiled to the batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", action="store_true", help="also time the optional RCCL all_gather of Tx")
    args = ap.parse_args()

    # Native libraries (RCCL prints a version banner on its first collective) write to file descriptor 1: park the
    # real stdout and point fd 1 at stderr, so that the ONE JSON line is the only thing this process puts on stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = args.gpus

    # torch first: libssq_hip.so then binds to the HIP runtime torch already loaded
    import torch
    import torch.distributed as dist
    have_cuda = torch.cuda.is_available()
    if not have_cuda:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ      # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))   # "nccl" is RCCL on ROCm

    from ssqueeze_rs_amd import _lib
    lib = _lib.load()
    _lib.check(lib.ssq_set_device(local_rank))

    N = 1 << args.log2n
    n_fft, hop, B = args.n_fft, args.hop, args.batch
    n_freqs, n_frames = n_fft // 2 + 1, (N - 1) // hop + 1
    bins_per_signal = n_freqs * n_frames
    alg_bytes_per_signal = 4 * N + 8 * bins_per_signal          # x once in, Tx once out (fp32)

    win = np.hanning(n_fft)
    plan = C.c_void_p()
    _lib.check(lib.ssq_stft_plan_create(C.byref(plan), _lib.SSQ_F32, N, win.ctypes.data_as(C.c_void_p),
                                        n_fft, hop, 1.0, 0, 0, -1.0, 0))
    assert lib.ssq_stft_plan_is_fused(plan) == 1
    stream = C.c_void_p()
    _lib.check(lib.ssq_stream_create(C.byref(stream)))
    d_x, d_out = C.c_void_p(), C.c_void_p()
    _lib.check(lib.ssq_dev_malloc(C.byref(d_x), B * N * 4))
    _lib.check(lib.ssq_dev_malloc(C.byref(d_out), B * bins_per_signal * 8))
    nd = min(args.distinct, B)
    host = synth_batch(nd, N)                                    # seeds 0..nd-1 (+rank offset below)
    if rank:
        host = np.roll(host, rank, axis=0)
    for b in range(B):
        _lib.check(lib.ssq_memcpy_h2d(C.c_void_p(d_x.value + b * N * 4), host[b % nd].ctypes.data_as(C.c_void_p),
                                      N * 4, stream))
    _lib.check(lib.ssq_stream_sync(stream))

    def step():
        _lib.check(lib.ssq_stft_plan_exec(plan, _lib.OUT_TX, d_x, B, d_out, None, 0, stream))

    for _ in range(args.warmup):
        step()
    _lib.check(lib.ssq_stream_sync(stream))

    evs = []
    for _ in range(args.steps):
        a, b_ = C.c_void_p(), C.c_void_p()
        _lib.check(lib.ssq_event_create(C.byref(a)))
        _lib.check(lib.ssq_event_create(C.byref(b_)))
        evs.append((a, b_))

    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b_ in evs:
        _lib.check(lib.ssq_event_record(a, stream))
        step()
        _lib.check(lib.ssq_event_record(b_, stream))
    _lib.check(lib.ssq_stream_sync(stream))
    torch.cuda.synchronize()
    t1 = time.perf_counter()            # this rank's K steps are complete; the MAX over ranks is taken below
    if use_dist:
        dist.barrier()

    wall = t1 - t0
    kern_ms = []
    for a, b_ in evs:
        ms = C.c_float(0)
        _lib.check(lib.ssq_event_elapsed_ms(a, b_, C.byref(ms)))
        kern_ms.append(ms.value)
    if use_dist:
        tt = torch.tensor([wall], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())

    gather_ms = None
    if args.gather and use_dist:
        # optional final gather of the Tx shards over xGMI (RCCL); timed separately, never part of `value`
        shard = torch.empty(min(B, 8) * bins_per_signal * 2, device="cuda", dtype=torch.float32)
        outl = torch.empty(world * shard.numel(), device="cuda", dtype=torch.float32)
        dist.all_gather_into_tensor(outl, shard)
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        dist.all_gather_into_tensor(outl, shard)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3

    if rank == 0:
        ms_per_step = wall / args.steps * 1e3
        value = n_gpus * B * bins_per_signal * args.steps / wall
        k_avg = float(np.mean(kern_ms))
        achieved = B * alg_bytes_per_signal / (k_avg * 1e-3) / 1e9
        cu = C.c_int(0)
        name = C.create_string_buffer(128)
        lib.ssq_device_info(C.byref(cu), None, name, 128)
        line = {
            "metric": "TF-bins/sec (ssq_stft, n_fft=1024)",
            "value": value, "unit": "TF-bins/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": f"synthetic multi-sine+chirp+noise, {nd} distinct seeds tiled to the batch",
            "config": {"workload": f"ssq_stft batch={B}/GPU x 2^{args.log2n} samples, n_fft={n_fft} hop={hop} "
                                   f"Hann, fs=1, reflect, sum; inputs and Tx resident in HBM",
                       "batch_per_gpu": B, "n_signal": N, "n_fft": n_fft, "hop": hop,
                       "n_freqs": n_freqs, "n_frames": n_frames, "parallelism": f"batch-sharded x{n_gpus}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic(B, args.log2n, n_fft, hop),
                         "kernel": "stft_tx1024_kernel<false,false,16> (interior tiles; the edge-tile launch of the same pass is inside the timed events)", "kernel_ms_avg": k_avg,
                         "kernel_ms_min": float(np.min(kern_ms)),
                         "alg_bytes_per_launch": B * alg_bytes_per_signal},
            "device": name.value.decode(), "cu_count": cu.value,
        }
        if gather_ms is not None:
            line["gather_ms_8sig_shards"] = gather_ms
        if not args.no_cpu_baseline and n_gpus == 1:
            line["cpu_baseline"] = cpu_baseline(N, n_fft, hop)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())

    for a, b_ in evs:
        lib.ssq_event_destroy(a)
        lib.ssq_event_destroy(b_)
    lib.ssq_dev_free(d_x)
    lib.ssq_dev_free(d_out)
    lib.ssq_stft_plan_destroy(plan)
    lib.ssq_stream_destroy(stream)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
