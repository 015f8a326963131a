#!/usr/bin/env python3
"""bench.py -- Haar 2-qubit decompositions/sec on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (TemplateOptimizer._run span loop k = 1..3, R restarts per
span, BasicCost + analytic gradient + in-kernel BFGS) over one batch of synthetic Haar targets.
Default workload = BASELINE.json configs[1]: CNOT basis, span <= 3, 1024 targets x 16 restarts,
fp64, per GPU (weak scaling: every rank gets its own 1024-target batches).  All target batches
are uploaded before the timed region; each step ends with the per-target results on the host
(and, for N > 1, a min-all-reduce of the best-loss vector over RCCL).

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement".
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP64_VALU_TFLOPS = 78.6  # 256 CU x 4 SIMD x 16 fp64 FMA lanes/clk x 2 flop x 2.4 GHz (MI355X_MICROARCH.md)
SUCCESS_LOSS = 1e-8  # BASELINE.json metric: loss < 1e-8


def f_eval(k: int) -> int:
    """Algorithmic flops of one fused loss+gradient evaluation."""
    return 3036 * k + 1247


WORKLOADS = {
    # name: (gate builder name, targets per step, restarts, description)
    "cfg2": ("cx", 1024, 16, "BASELINE configs[1]: CNOT basis span<=3, 1024 Haar targets x 16 restarts, fp64"),
    "cfg3": ("sqiswap", 65536, 32, "BASELINE configs[2]: sqrt(iSWAP) basis span<=3, 65536 Haar targets x 32 restarts, fp64"),
    "cfg4": ("iswap+b", 32768, 16, "BASELINE configs[3] per-GPU shard: iSWAP + B mixed basis, 32768 Haar targets x 16 restarts"),
}


def gate_table(name: str) -> np.ndarray:
    from slam_decomposition_amd import gates as G

    if name == "cx":
        return np.stack([G.CXGate().to_matrix()])
    if name == "sqiswap":
        return np.stack([G.RiSwapGate(0.5).to_matrix()])
    if name == "iswap+b":
        return np.stack([G.RiSwapGate(1.0).to_matrix(), G.BerkeleyGate().to_matrix()])
    raise ValueError(name)


def make_targets(n: int, seed0: int) -> np.ndarray:
    """T_i = unitary_group.rvs(4, default_rng(seed0 + i)); generated with a small
    process pool because SciPy draws them one at a time."""
    from slam_decomposition_amd.sampler import random_unitary

    return np.stack([random_unitary(4, seed=seed0 + i) for i in range(n)])


def _cpu_one(args):
    """One target through the reference path on the CPU oracle (SciPy BFGS, finite differences:
    src/slam/optimizer.py:270-278), restarts sequential with early break like the reference."""
    seed0, idx, gname, restarts, seed = args
    from oracle import slam_oracle as o

    gates = {"cx": [o.cx_matrix()], "sqiswap": [o.riswap_matrix(0.5)], "iswap+b": [o.riswap_matrix(1.0), o.berkeley_matrix()]}[gname]
    target = o.haar_unitary(seed0 + idx)
    t0 = time.perf_counter()
    best, _, k, stats = o.run_reference(
        target, gates, range(1, 4), restarts, SUCCESS_LOSS, x0_fn=lambda kk, r: o.x0_philox(seed, idx, r, kk)
    )
    return best, k, time.perf_counter() - t0, stats["nfev"]


def cpu_baseline(gname: str, restarts: int, seed0: int, seed: int, n_sample: int):
    import multiprocessing as mp

    cores = min(os.cpu_count() or 1, 16)
    jobs = [(seed0, i, gname, restarts, seed) for i in range(n_sample)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(_cpu_one, jobs, chunksize=1)
    wall = time.perf_counter() - t0
    ok = sum(1 for r in res if r[0] < SUCCESS_LOSS)
    cpu_s = sum(r[2] for r in res)
    return {
        "value": ok / wall,
        "unit": "decompositions/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{n_sample} targets of the same workload (SciPy BFGS + finite differences on the NumPy oracle, "
        f"sequential restarts with early break), {cpu_s:.1f} core-seconds, {wall:.1f} s wall",
        "per_core": ok / cpu_s if cpu_s > 0 else None,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--targets", type=int, default=None, help="override targets per step per GPU")
    ap.add_argument("--restarts", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=8)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    dist = None
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from slam_decomposition_amd import _ffi

    gname, n_per_step, restarts, desc = WORKLOADS[args.workload]
    if args.targets:
        n_per_step = args.targets
    if args.restarts:
        restarts = args.restarts
    steps, warmup = args.steps, args.warmup
    total_steps = steps + warmup
    seed0 = 20260000 + rank * total_steps * n_per_step  # disjoint targets per rank (weak scaling)
    opt_seed = 20261003

    ctx = _ffi.Context(local_rank)
    dev_name, cus, clock_khz = ctx.device_info()
    table = gate_table(gname)
    ctx.set_gates(table)
    targets = make_targets(total_steps * n_per_step, seed0)
    ctx.set_targets(targets)  # every batch resident in HBM before the timed region
    gate_seqs = [[i % len(table) for i in range(k)] for k in (1, 2, 3)]
    prm = _ffi.OptParams(restarts=restarts, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=opt_seed, flags=_ffi.FLAG_EARLY_EXIT)
    threshold = 1e-10  # reference SUCCESS_THRESHOLD (optimizer.py:18); the metric counts loss < 1e-8

    merged = None
    if world > 1:
        merged = torch.full((world * n_per_step,), float("inf"), dtype=torch.float64, device="cuda")

    def one_step(s: int):
        first = s * n_per_step
        best_loss, best_x, best_cycles = ctx.decompose_range(first, n_per_step, 1, 3, gate_seqs, prm, threshold)
        if world > 1:
            # final best-loss all-reduce (min) over RCCL: every rank ends with the whole job's losses
            merged.fill_(float("inf"))
            merged[rank * n_per_step : (rank + 1) * n_per_step] = torch.from_numpy(best_loss).cuda()
            dist.all_reduce(merged, op=dist.ReduceOp.MIN)
        return best_loss, best_cycles

    def sync():
        ctx.synchronize()
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for s in range(warmup):
        one_step(s)
    sync()
    ctx.reset_stats()
    solved = 0
    cyc_hist = np.zeros(4, dtype=np.int64)
    worst = 0.0
    t0 = time.perf_counter()
    for s in range(warmup, total_steps):
        bl, bc = one_step(s)
        ok = bl < SUCCESS_LOSS
        solved += int(ok.sum())
        worst = max(worst, float(bl.max()))
        cyc_hist += np.bincount(np.clip(bc, 0, 3), minlength=4)
    sync()
    elapsed = time.perf_counter() - t0
    st = ctx.stats()

    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        cnt = torch.tensor([solved], dtype=torch.float64, device="cuda")
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        solved_all = int(cnt.item())
    else:
        solved_all = solved

    if rank == 0:
        flops = sum(st["evals"][k] * f_eval(k) for k in (1, 2, 3))
        kernel_s = st["kernel_ms"] * 1e-3
        achieved = flops / kernel_s / 1e12 if kernel_s > 0 else 0.0
        out = {
            "metric": "Haar 2-qubit decompositions/sec (span<=3, loss<1e-8)",
            "value": solved_all / elapsed,
            "unit": "decompositions/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": desc,
                "basis": gname,
                "targets_per_step_per_gpu": n_per_step,
                "restarts": restarts,
                "span_max": 3,
                "success_threshold": threshold,
                "parallelism": f"targets sharded over {world} GPU(s), no data-path collective",
                "device": dev_name,
                "compute_units": cus,
            },
            "solved_fraction": solved_all / (world * steps * n_per_step),
            "best_cycles_hist_rank0": {str(k): int(cyc_hist[k]) for k in range(4)},
            "worst_loss_rank0": worst,
            "roofline": {
                "bound": "valu_fp64",
                "achieved": achieved,
                "peak": PEAK_FP64_VALU_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP64_VALU_TFLOPS,
                "traffic": None,
                "kernel": "minimize_kernel<K> (k=1..3)",
                "kernel_ms_total": st["kernel_ms"],
                "kernel_launches": st["kernel_launches"],
                "avg_launch_ms": st["kernel_ms"] / max(1, st["kernel_launches"]),
                "evals_per_span": {str(k): st["evals"][k] for k in (1, 2, 3)},
                "items_per_span": {str(k): st["items"][k] for k in (1, 2, 3)},
                "flops_per_eval": {str(k): f_eval(k) for k in (1, 2, 3)},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(gname, restarts, 20260000, opt_seed, args.cpu_sample)
        print(json.dumps(out), flush=True)

    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
