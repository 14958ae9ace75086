"""bench.py's N-GPU harness, on CPU: the number of GPUs is the world size (never the flag), a plain
`python bench.py --gpus N` spawns N fresh ranks before any GPU call, and the weak / strong job shapes.
World-size-2 run over gloo through the same launcher code (bench.spawn_ranks -> torch.distributed.run)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

HELPER = os.path.join(ROOT, "tests", "helpers", "rank_echo.py")


def test_launch_plan_uses_world_size_not_the_flag():
    assert bench.launch_plan(1, env={}) == ("run", 1, 0, 0)
    assert bench.launch_plan(4, env={}) == ("spawn", 4)
    env = {"RANK": "1", "WORLD_SIZE": "2", "LOCAL_RANK": "1"}
    assert bench.launch_plan(2, env=env) == ("run", 2, 1, 1)
    kind, msg = bench.launch_plan(8, env=env)
    assert kind == "error" and "WORLD_SIZE 2" in msg
    assert bench.launch_plan(0, env={})[0] == "error"
    # a single rank started by torch.distributed.run is still "run"
    assert bench.launch_plan(1, env={"RANK": "0", "WORLD_SIZE": "1"}) == ("run", 1, 0, 0)


def test_job_shape_weak_and_strong():
    # weak: every rank its own batch; global seeds do not overlap
    assert bench.job_shape(8, 3, 256, 0) == (256, 768, 2048, "weak")
    # strong: BASELINE config 3 = 256 signals over 8 GPUs = 32 each
    shapes = [bench.job_shape(8, r, 256, 256) for r in range(8)]
    assert [s[0] for s in shapes] == [32] * 8
    assert [s[1] for s in shapes] == [32 * r for r in range(8)]
    assert all(s[2] == 256 and s[3] == "strong" for s in shapes)
    # ragged strong split
    assert [bench.job_shape(3, r, 0, 10)[0] for r in range(3)] == [4, 3, 3]


def test_gpus_flag_mismatch_exits_2_before_touching_a_gpu():
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2
    assert "disagrees with WORLD_SIZE" in r.stderr and r.stdout == ""


@pytest.mark.parametrize("argv,signals,last_first,scaling", [
    (["--gpus", "2", "--batch", "5"], 10, 5, "weak"),
    (["--gpus", "2", "--total-batch", "7"], 7, 4, "strong"),
])
def test_world2_gloo_through_the_self_launcher(argv, signals, last_first, scaling):
    """The parent (no RANK in its environment) spawns 2 ranks; they see WORLD_SIZE=2 and agree with --gpus."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.spawn_ranks(2, %r, script=%r))" % (ROOT, argv, HELPER))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["signals"] == signals and d["last_first"] == last_first
    assert d["scaling"] == scaling and abs(d["wall"] - 0.5) < 1e-12


def test_world8_strong_c3_shape_end_to_end_over_gloo():
    """BASELINE config 3 as the driver launches it at N = 8: `--gpus 8 --total-batch 256` -> 8 ranks (gloo on CPU, the
    stub step of tests/helpers/rank_echo.py), 32 signals each, contiguous blocks, MAX-over-ranks timing, one JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    argv = ["--gpus", "8", "--total-batch", "256"]
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.spawn_ranks(8, %r, script=%r))" % (ROOT, argv, HELPER))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["signals"] == 256 and d["total"] == 256 and d["scaling"] == "strong"
    assert d["last_first"] == 224                       # rank 7 starts at signal 7 * 32
    assert abs(d["wall"] - 2.0) < 1e-12                 # MAX over ranks of 0.25 * (rank + 1)
