"""CPU: bench.py's N > 1 plumbing end to end -- its own launcher, one process per rank, the file communicator
(SLAM_BENCH_COMM=file), the final min-all-reduce of the best-loss vector and the JSON line -- against a stub context
(tests/bench_stub.py, through bench.py's SLAM_BENCH_TEST_STUB hook).  No kernel runs here: what is checked is the host
logic a first multi-GPU run depends on.  The GPU side of the same path: tests/test_gpu_round3.py."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra, timeout=180):
    env = dict(os.environ, SLAM_BENCH_TEST_STUB=os.path.join(ROOT, "tests", "bench_stub.py"), **env_extra)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-secondary", "--per-span-steps", "0"] + extra,
                       env=env, capture_output=True, text=True, timeout=timeout)
    return p


def _solved(first, count):
    return sum(1 for i in range(first, first + count) if i % 7 != 0)


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_ranks_through_the_launcher_and_the_file_communicator(scaling):
    steps, warmup, n = 4, 1, 512
    p = _run(["--gpus", "2", "--steps", str(steps), "--warmup", str(warmup), "--targets", str(n), "--restarts", "4", "--scaling", scaling,
              "--repeats", "3"], {"SLAM_BENCH_COMM": "file"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout  # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == steps and out["warmup"] == warmup and out["scaling"] == scaling
    assert out["comm"] == "FileComm" and out["rccl_world"] is None  # no RCCL in this run, and the line says so
    assert len(out["rank_devices"]) == 2 and all("stub-device" in d for d in out["rank_devices"])
    assert out["repetitions"] == 3 and out["ms_per_step_min"] <= out["ms_per_step"] <= out["ms_per_step_max"]
    assert out["data"].startswith("STUB")
    per_rank = n if scaling == "weak" else n // 2
    assert out["config"]["targets_per_step_per_gpu"] == per_rank
    # the merged best-loss vector: rank r's targets start at TARGET_SEED0 + r * (steps + warmup) * per_rank (disjoint
    # shards), timed steps are warmup .. warmup + steps - 1
    seed0 = 20260000
    want = sum(_solved(seed0 + r * (steps + warmup) * per_rank + warmup * per_rank, steps * per_rank) for r in range(2))
    assert abs(out["solved_fraction"] - want / (2 * steps * per_rank)) < 1e-12
    assert out["value"] > 0


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_eight_ranks_slice_arithmetic_and_rank_diagnostics(scaling):
    """The world size the driver's scaling run uses: 8 ranks through the launcher and the file communicator.  Checks the slice
    arithmetic of the merged best-loss vector (rank * n_loc + (s - first_step) * n_per_step) through the solved count, and the
    per-rank diagnostics a first 8-GPU run prints (own time, solved count, collective duration, evaluations per rank)."""
    world, steps, warmup, n = 8, 3, 1, 256
    p = _run(["--gpus", str(world), "--steps", str(steps), "--warmup", str(warmup), "--targets", str(n), "--restarts", "2", "--scaling", scaling,
              "--repeats", "3"], {"SLAM_BENCH_COMM": "file"}, timeout=400)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["scaling"] == scaling and len(out["rank_devices"]) == world
    per_rank = n if scaling == "weak" else n // world
    assert out["config"]["targets_per_step_per_gpu"] == per_rank
    seed0 = 20260000
    per_rank_solved = [_solved(seed0 + r * (steps + warmup) * per_rank + warmup * per_rank, steps * per_rank) for r in range(world)]
    assert abs(out["solved_fraction"] - sum(per_rank_solved) / (world * steps * per_rank)) < 1e-12
    d = out["rank_diag"]
    assert d["solved"] == per_rank_solved  # every rank's own count, in rank order
    assert len(d["own_ms"]) == world and all(v > 0 for v in d["own_ms"])
    assert len(d["collective_ms"]) == world and all(v >= 0 for v in d["collective_ms"])
    assert d["evals"] == [3 * 40 * steps * per_rank * 2] * world  # the stub's 40 evaluations per item and span
    assert out["ms_per_step"] * steps >= max(d["own_ms"]) * 0.999  # the line's time is the MAX over ranks, barrier included


@pytest.mark.parametrize("world", [2, 8])
def test_strong_default_workload_through_the_resident_merge_branch(world):
    """The driver's N > 1 command on the DEFAULT workload with `--scaling strong` (65 536 x 32 split over the ranks), through the
    branch an RCCL run takes -- `resident_merge`: parallel.merge_slices feeding merge_add per context window, then ONE min-all-reduce
    -- rehearsed with stub contexts and a host stand-in for the device-side merge (tests/bench_stub.py:StubRaw, which refuses
    overlapping slices and holes).  8 ranks: 8192 targets per rank per step, 16 calls in flight."""
    steps, warmup = 20, 5
    p = _run(["--gpus", str(world), "--steps", str(steps), "--warmup", str(warmup), "--scaling", "strong", "--repeats", "1"],
             {"SLAM_BENCH_COMM": "file", "SLAM_BENCH_STUB_RAW": "1"}, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    per_rank = 65536 // world
    assert out["n_gpus"] == world and out["scaling"] == "strong" and out["config"]["targets_per_step_per_gpu"] == per_rank
    assert "device to device" in out["config"]["final_collective"]
    seed0 = 20260000
    per_rank_solved = [_solved(seed0 + r * (steps + warmup) * per_rank + warmup * per_rank, steps * per_rank) for r in range(world)]
    assert abs(out["solved_fraction"] - sum(per_rank_solved) / (world * steps * per_rank)) < 1e-12
    assert out["rank_diag"]["solved"] == per_rank_solved
    assert out["rank_diag"]["host_threads"] == [out["config"]["batches_in_flight_per_gpu"] + 1] * world


def test_launcher_path_with_one_rank_and_plain_path_agree():
    """`SLAM_BENCH_FORCE_LAUNCH=1 bench.py --gpus 1` (launcher, rank process, communicator) and plain `bench.py` report the
    same work; and a rank whose communicator cannot come up ends the job non-zero instead of falling back."""
    a = _run(["--steps", "3", "--warmup", "1", "--targets", "256", "--restarts", "4"], {})
    b = _run(["--steps", "3", "--warmup", "1", "--targets", "256", "--restarts", "4"], {"SLAM_BENCH_FORCE_LAUNCH": "1", "SLAM_BENCH_COMM": "file"})
    assert a.returncode == 0 and b.returncode == 0, (a.stderr[-1500:], b.stderr[-1500:])
    ja, jb = (json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0]) for p in (a, b))
    assert ja["solved_fraction"] == jb["solved_fraction"] and ja["comm"] == "LocalComm" and jb["comm"] == "FileComm"
    assert ja["roofline"]["evals_per_span"] == jb["roofline"]["evals_per_span"]
    # no GPU here: RCCL cannot come up -> exit code != 0, no JSON line, no silent file fallback
    c = _run(["--gpus", "2", "--steps", "2", "--warmup", "0", "--targets", "64", "--restarts", "2"], {})
    assert c.returncode != 0
    assert not [l for l in c.stdout.splitlines() if l.startswith("{")]
    assert "RCCL communicator failed" in c.stderr


def test_two_ranks_under_torch_distributed_run():
    """The driver's N > 1 launch line -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- with bench.py as one of the ranks (RANK / LOCAL_RANK / WORLD_SIZE from the
    environment, rendezvous path from the launcher's pid + port).  torch only launches; bench.py itself never imports it."""
    pytest.importorskip("torch")
    env = dict(os.environ, SLAM_BENCH_TEST_STUB=os.path.join(ROOT, "tests", "bench_stub.py"), SLAM_BENCH_COMM="file")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SLAM_COMM_FILE", "SLAM_COMM_DIR"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--targets", "256",
                        "--restarts", "4", "--no-cpu-baseline", "--no-secondary", "--per-span-steps", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["comm"] == "FileComm" and len(out["rank_devices"]) == 2 and out["steps"] == 3
