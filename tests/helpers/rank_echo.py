"""Rank program for tests/test_bench_launcher.py: stands in for bench.py's rank body on CPU (gloo).
It takes its world from the environment exactly as bench.py does (bench.launch_plan / job_shape), reduces the
per-rank signal counts and prints ONE JSON line on rank 0."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def main():
    args = bench.parse_args(sys.argv[1:])
    plan = bench.launch_plan(args.gpus)
    if plan[0] != "run":
        print("rank_echo: " + str(plan), file=sys.stderr)
        sys.exit(2)
    _, world, rank, local_rank = plan
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    try:
        B, first, total, scaling = bench.job_shape(world, rank, args.batch, args.total_batch)
        t = torch.tensor([B, first if rank == world - 1 else 0], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        wall = torch.tensor([0.25 * (rank + 1)], dtype=torch.float64)
        dist.all_reduce(wall, op=dist.ReduceOp.MAX)             # bench.py: MAX over ranks
        if rank == 0:
            print(json.dumps({"n_gpus": dist.get_world_size(), "signals": int(t[0]), "last_first": int(t[1]),
                              "total": total, "scaling": scaling, "wall": float(wall.item())}), flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
