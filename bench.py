#!/usr/bin/env python3
"""bench.py -- Haar 2-qubit decompositions/sec on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (TemplateOptimizer._run span loop k = 1..3, R restarts per
span, BasicCost + analytic gradient + in-kernel BFGS) over one batch of synthetic Haar targets.
Default workload = BASELINE.json configs[2], the largest single-GPU configuration: sqrt(iSWAP)
basis, span <= 3, 65 536 targets x 32 restarts, fp64, per GPU (weak scaling: every rank gets its
own batches).  All target batches are resident in HBM before the timed region; each step ends with
the per-target results on the host.  Several batches are kept in flight per GPU (host threads, one
context + HIP stream each) so that the straggler tail of one batch overlaps the next.

N > 1: one process per GPU.  `python bench.py --gpus N` launches its N ranks itself (fresh
processes, before anything touches a GPU); under `python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N` it is one of the ranks (RANK / LOCAL_RANK / WORLD_SIZE from the environment).
Either way the ranks meet through libslamhip's RCCL communicator (slam_comm_*, no torch): barrier on
both sides of the timed region, MAX over ranks of the time, and the job's ONE collective -- the
final min-all-reduce of the best-loss vector over xGMI.

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement".
"""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
# one hardware queue per batch in flight (the HIP runtime maps streams onto 4 hardware queues by default;
# streams that share a queue serialise).  Must be set before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the bench lives in benchlib/ (workloads, CPU baseline + parity sample, launcher, secondary probes, timed region + CLI); the names dev tools
# and tests import from here stay importable
from benchlib.cpu_baseline import cpu_baseline, parity_sample, parity_sample_size, usable_cores, weyl_distance  # noqa: E402,F401
from benchlib.run import main, run_workload  # noqa: E402,F401
from benchlib.secondary import run_api, run_long, run_medium_call, run_v2  # noqa: E402,F401
from benchlib.workloads import (OPT_SEED, PEAK_FP64_VALU_TFLOPS, SUCCESS_LOSS, TARGET_SEED0, WORKLOADS, f_eval, f_eval_v2, f_forward,  # noqa: E402,F401
                                gate_table, make_targets, sweep_gate)

if __name__ == "__main__":
    main()
