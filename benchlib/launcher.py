"""bench: N > 1 -- the launcher (`python bench.py --gpus N` starts its N ranks itself), the communicator, small collectives."""
from __future__ import annotations

import os
import subprocess
import sys
import tempfile
import time

import numpy as np

from .workloads import ROOT

BENCH_PY = os.path.join(ROOT, "bench.py")


def gather_strings(comm, rank: int, world: int, text: str, width: int = 64):
    """Every rank's short string on every rank, through the communicator's sum-all-reduce (bytes as doubles)."""
    buf = np.zeros(world * width)
    raw = text.encode()[:width]
    buf[rank * width : rank * width + len(raw)] = list(raw)
    comm.allreduce_sum(buf)
    return [bytes(int(v) for v in buf[r * width : (r + 1) * width] if v > 0).decode(errors="replace") for r in range(world)]


# ------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without torchrun
# ------------------------------------------------------------------------------------------------
def launch_ranks(n: int) -> int:
    """Start the N ranks as fresh processes (nothing in this process has touched the GPU: no exec-after-HIP-init,
    no fork of an initialised runtime) and return the worst exit code.  Rank 0 prints the JSON line."""
    with tempfile.TemporaryDirectory(prefix="slam_bench_") as tmp:
        procs = []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), SLAM_COMM_FILE=os.path.join(tmp, "rccl.id"),
                       SLAM_COMM_DIR=os.path.join(tmp, "filecomm"), SLAM_BENCH_RANK_PROCESS="1")
            procs.append(subprocess.Popen([sys.executable, BENCH_PY] + sys.argv[1:], env=env))
        rc = 0
        try:
            live = list(procs)
            while live:
                for p in list(live):
                    code = p.poll()
                    if code is not None:
                        live.remove(p)
                        rc = rc or code
                if rc:
                    break  # a rank that died leaves the others waiting in a collective: do not wait for them
                time.sleep(0.05)
        finally:
            for p in procs:  # end what is still running, by pid
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
        return rc


class _StdoutToStderr:
    """RCCL prints its version banner on the C-level stdout when a communicator is created; rank 0's stdout must carry
    the JSON line only, so fd 1 points at stderr while the communicator comes up."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def make_comm(rank: int, world: int, local_rank: int):
    from slam_decomposition_amd import parallel

    if world == 1 and not os.environ.get("SLAM_BENCH_RANK_PROCESS"):
        return parallel.LocalComm()
    if os.environ.get("SLAM_BENCH_COMM", "rccl") == "file":
        # rehearsal of the N > 1 path on a one-GPU box (RCCL refuses several ranks on one device)
        return parallel.FileComm(rank, world, os.environ.get("SLAM_COMM_DIR") or parallel.rendezvous_path() + ".d")
    from slam_decomposition_amd import _ffi

    # No fallback: a rank whose RCCL communicator does not come up ends the job with a non-zero exit code (the launcher
    # then stops the other ranks).  A per-rank fallback would leave the healthy ranks blocked in ncclCommInitRank, and a
    # job-wide one would print a scaling number whose collective went through the file system.
    try:
        with _StdoutToStderr():
            comm = parallel.RcclComm(local_rank % max(1, _ffi.device_count()), rank, world, parallel.rendezvous_path())
            comm.barrier()  # first collective (lazy channel set-up and its messages) before anything is timed or printed
    except Exception as exc:
        print(f"[bench rank {rank}] RCCL communicator failed: {exc}  (SLAM_BENCH_COMM=file rehearses the N > 1 path without RCCL)",
              file=sys.stderr, flush=True)
        raise SystemExit(3)
    if (comm.rccl_rank, comm.rccl_world) != (rank, world):
        print(f"[bench rank {rank}] RCCL reports rank {comm.rccl_rank} of {comm.rccl_world}, the launcher said {rank} of {world}", file=sys.stderr, flush=True)
        raise SystemExit(3)
    return comm
