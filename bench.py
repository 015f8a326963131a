#!/usr/bin/env python3
"""bench.py -- Haar 2-qubit decompositions/sec on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (TemplateOptimizer._run span loop k = 1..3, R restarts per
span, BasicCost + analytic gradient + in-kernel BFGS) over one batch of synthetic Haar targets.
Default workload = BASELINE.json configs[1]: CNOT basis, span <= 3, 1024 targets x 16 restarts,
fp64, per GPU (weak scaling: every rank gets its own 1024-target batches).  All target batches
are uploaded before the timed region; each step ends with the per-target results on the host.
Several batches are kept in flight per GPU (host threads, one context + HIP stream each) so that
the straggler tail of one batch overlaps the next.  For N > 1 the job ends with ONE collective:
the min-all-reduce of the best-loss vector over RCCL.

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
# one hardware queue per batch in flight (the HIP runtime maps streams onto 4 hardware queues by default;
# streams that share a queue serialise).  Must be set before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
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
    # one step = one basis gate of this GPU's 16 (of 128) against the 4096 shared targets
    "cfg5": ("cgsweep", 4096, 16, "BASELINE configs[4] per-GPU shard: 16 of 128 ConversionGain(0,0,gc,gg,1) bases x 4096 shared Haar targets x 16 restarts"),
}
SWEEP_BASES_PER_GPU = 16
SWEEP_CPU_BASIS = 64  # m = 9/32, p = 0: the basis the CPU baseline of cfg5 runs


def sweep_gate(b: int) -> np.ndarray:
    """Basis b of the 128-gate parametric-Hamiltonian sweep (SURVEY.md §8(d) cfg 5, shaped like build_gates(),
    utils/gates/bare_candidates.py:47-69): gc = p m pi, gg = (1 - p) m pi, 16 values of m in (0, 0.5] x 8 of p in [0, 1]."""
    from slam_decomposition_amd import gates as G

    m = 0.5 * (b // 8 + 1) / 16
    pfrac = (b % 8) / 7
    return G.ConversionGainGate(0.0, 0.0, pfrac * m * np.pi, (1 - pfrac) * m * np.pi, 1.0).to_matrix()


def gate_table(name: str) -> np.ndarray:
    from slam_decomposition_amd import gates as G

    if name == "cx":
        return np.stack([G.CXGate().to_matrix()])
    if name == "sqiswap":
        return np.stack([G.RiSwapGate(0.5).to_matrix()])
    if name == "iswap+b":
        return np.stack([G.RiSwapGate(1.0).to_matrix(), G.BerkeleyGate().to_matrix()])
    if name == "cgsweep":
        return np.stack([sweep_gate(0)])
    raise ValueError(name)


def make_targets(n: int, seed0: int) -> np.ndarray:
    """T_i = unitary_group.rvs(4, default_rng(seed0 + i)); generated with a small
    process pool because SciPy draws them one at a time."""
    from slam_decomposition_amd.sampler import random_unitary

    return np.stack([random_unitary(4, seed=seed0 + i) for i in range(n)])


def _cpu_one(args):
    """One target through the reference path on the CPU oracle (SciPy BFGS, finite differences:
    src/slam/optimizer.py:270-278), restarts sequential with early break like the reference."""
    seed0, idx, gname, restarts, seed, host_targets, analytic = args
    from oracle import slam_oracle as o

    if gname == "cgsweep":
        gates = [sweep_gate(SWEEP_CPU_BASIS)]  # one representative basis of the sweep (sqrt(iSWAP)-like strength)
    else:
        gates = {"cx": [o.cx_matrix()], "sqiswap": [o.riswap_matrix(0.5)], "iswap+b": [o.riswap_matrix(1.0), o.berkeley_matrix()]}[gname]
    target = o.haar_unitary(seed0 + idx) if host_targets else o.haar_philox_port(seed0, idx)
    t0 = time.perf_counter()
    best, _, k, stats = o.run_reference(
        target, gates, range(1, 4), restarts, SUCCESS_LOSS, x0_fn=lambda kk, r: o.x0_philox(seed, idx, r, kk),
        analytic_jac=analytic,
    )
    return best, k, time.perf_counter() - t0, stats["nfev"]


def cpu_baseline(gname: str, restarts: int, seed0: int, seed: int, n_sample: int, host_targets: bool):
    import multiprocessing as mp

    cores = min(os.cpu_count() or 1, 16)
    with mp.get_context("spawn").Pool(cores) as pool:
        t0 = time.perf_counter()
        res = pool.map(_cpu_one, [(seed0, i, gname, restarts, seed, host_targets, False) for i in range(n_sample)], chunksize=1)
        wall = time.perf_counter() - t0
        # second, stronger CPU line: the same loop with the oracle's analytic gradient
        t1 = time.perf_counter()
        res_j = pool.map(_cpu_one, [(seed0, i, gname, restarts, seed, host_targets, True) for i in range(n_sample)], chunksize=1)
        wall_j = time.perf_counter() - t1
    ok = sum(1 for r in res if r[0] < SUCCESS_LOSS)
    cpu_s = sum(r[2] for r in res)
    ok_j = sum(1 for r in res_j if r[0] < SUCCESS_LOSS)
    cpu_sj = sum(r[2] for r in res_j)
    return {
        "value": ok / wall,
        "unit": "decompositions/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{n_sample} targets of the same workload{' (sweep basis %d only)' % SWEEP_CPU_BASIS if gname == 'cgsweep' else ''} "
        f"(SciPy BFGS + finite differences on the NumPy oracle, "
        f"sequential restarts with early break), {cpu_s:.1f} core-seconds, {wall:.1f} s wall",
        "per_core": ok / cpu_s if cpu_s > 0 else None,
        "analytic_jac": {"value": ok_j / wall_j, "per_core": ok_j / cpu_sj if cpu_sj > 0 else None,
                         "note": "same sample and loop, SciPy BFGS with the oracle's analytic gradient"},
    }


def traffic_per_launch(workload: str):
    """HBM bytes per optimizer-kernel launch (mean over the three spans) from the committed PMC passes
    (profiles/r1d_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE); None for workloads that were not profiled."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1d_traffic.json")
    try:
        t = json.load(open(path))[workload]
    except (OSError, KeyError, ValueError):
        return None
    return sum(t.values()) / len(t)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 320 for cfg2, 9 for the big workloads)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 32 for cfg2, 3 otherwise)")
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--targets", type=int, default=None, help="override targets per step per GPU")
    ap.add_argument("--restarts", type=int, default=None)
    ap.add_argument("--streams", type=int, default=None,
                    help="batches in flight per GPU (one host thread + context + HIP stream each); default 16 for cfg2-sized batches, 3 otherwise")
    ap.add_argument("--items-per-quad", type=int, default=int(os.environ.get("SLAM_BENCH_IPQ", "-1")),
                    help="launch shaping (slam_opt_params.items_per_quad); 0 = library default (one item per quad, lowest "
                         "latency); default here: 3 for small batches with several in flight (+4.7 %% measured), else 0")
    ap.add_argument("--host-targets", action="store_true", help="draw the Haar targets with SciPy on the host instead of on the device")
    ap.add_argument("--span-rules", action="store_true",
                    help="polytope mode (CircuitTemplate(use_polytopes=True)): each target is optimised only at the template "
                         "size the analytic span rules assign to it (device c1c2c3 + span_rules.py) instead of spans 1..3")
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

        # "nccl" is RCCL over xGMI; SLAM_BENCH_BACKEND=gloo rehearses the N > 1 path on a one-GPU box
        backend = os.environ.get("SLAM_BENCH_BACKEND", "nccl")
        ndev = torch.cuda.device_count()
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            tdev = "cuda"
        else:
            local_rank = local_rank % max(1, ndev)
            dist.init_process_group(backend)
            tdev = "cpu"

    from slam_decomposition_amd import _ffi

    gname, n_per_step, restarts, desc = WORKLOADS[args.workload]
    if args.targets:
        n_per_step = args.targets
    if args.restarts:
        restarts = args.restarts
    small = n_per_step * restarts <= 65536
    steps = args.steps if args.steps is not None else (320 if small else 9)
    warmup = args.warmup if args.warmup is not None else (32 if small else 3)
    total_steps = steps + warmup
    seed0 = 20260000 + rank * total_steps * n_per_step  # disjoint targets per rank (weak scaling)
    opt_seed = 20261003

    n_streams = args.streams if args.streams else (16 if n_per_step * restarts <= 65536 else 3)
    n_streams = max(1, min(n_streams, steps))
    ctxs = [_ffi.Context(local_rank) for _ in range(n_streams)]
    ctx = ctxs[0]
    dev_name, cus, clock_khz = ctx.device_info()
    table = gate_table(gname)
    # every batch resident in HBM before the timed region: Haar targets generated in place by the device
    # sampler (slam_sample_haar; --host-targets: SciPy's sampler on the host, ~55 us per target, then uploaded)
    targets = make_targets(total_steps * n_per_step, seed0) if args.host_targets else None
    sweep = gname == "cgsweep"
    n_resident = n_per_step if sweep else total_steps * n_per_step  # the sweep's targets are shared by all bases
    if sweep and args.host_targets:
        targets = targets[:n_per_step]
    for c in ctxs:
        c.set_gates(table)
        if args.host_targets:
            c.set_targets(targets)
        else:
            c.sample_haar(seed0 if not sweep else 20260000, n_resident)
    gate_seqs = [[i % len(table) for i in range(k)] for k in (1, 2, 3)]
    ipq = args.items_per_quad if args.items_per_quad >= 0 else (3 if (small and n_streams > 1) else 0)
    prm = _ffi.OptParams(restarts=restarts, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=opt_seed, flags=_ffi.FLAG_EARLY_EXIT,
                          items_per_quad=ipq)
    threshold = 1e-10  # reference SUCCESS_THRESHOLD (optimizer.py:18); the metric counts loss < 1e-8

    def basis_of(s: int) -> int:
        # rank r takes column p = r of the (m, p) grid: all 16 strengths m, so every GPU has weak and strong gates
        return (s % SWEEP_BASES_PER_GPU) * 8 + rank % 8

    gate_coords = None
    if args.span_rules:
        from slam_decomposition_amd import span_rules
        from slam_decomposition_amd.weyl import c1c2c3 as host_c1c2c3

        if len(table) != 1:
            raise SystemExit("--span-rules needs a single basis gate (cfg2, cfg3)")
        gate_coords = host_c1c2c3(table[0])
        span_rules.family_of(gate_coords)

    def one_step(s: int, c):
        if args.span_rules:
            first = s * n_per_step
            spans = span_rules.minimal_span(c.targets_c1c2c3(first, n_per_step), gate_coords)
            for k in np.unique(spans):
                k = int(k)
                if k < 1:
                    continue  # local targets need no 2Q gate
                c.decompose_list(first + np.nonzero(spans == k)[0], k, k, [gate_seqs[k - 1]], prm, threshold, k_layout=3)
            best_loss, best_x, best_cycles = c.fetch_results_range(3, first, n_per_step)
            return best_loss, best_cycles
        if sweep:
            c.set_gates(np.stack([sweep_gate(basis_of(s))]))
            best_loss, best_x, best_cycles = c.decompose_range(0, n_per_step, 1, 3, gate_seqs, prm, threshold)
        else:
            best_loss, best_x, best_cycles = c.decompose_range(s * n_per_step, n_per_step, 1, 3, gate_seqs, prm, threshold)
        return best_loss, best_cycles

    def sync():
        for c in ctxs:
            c.synchronize()
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    import threading

    def run_steps(step_ids, results):
        # steps are dealt round-robin to n_streams host threads, each with its own context / HIP stream,
        # so the tail of one batch (a stage lasts as long as its slowest work item) overlaps the next batch
        def worker(w):
            for s in step_ids[w::n_streams]:
                results[s] = one_step(s, ctxs[w])

        if n_streams == 1:
            worker(0)
        else:
            threads = [threading.Thread(target=worker, args=(w,)) for w in range(n_streams)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        if world > 1:
            # the job's one collective: final best-loss all-reduce (min) over RCCL / xGMI.  Every rank
            # contributes +inf outside its shard and ends with the whole job's per-target losses.
            n_loc = len(step_ids) * n_per_step
            merged = torch.full((world * n_loc,), float("inf"), dtype=torch.float64, device=tdev)
            mine = np.concatenate([results[s][0] for s in step_ids])
            merged[rank * n_loc : (rank + 1) * n_loc] = torch.from_numpy(mine).to(tdev)
            dist.all_reduce(merged, op=dist.ReduceOp.MIN)
            results["merged_solved"] = int((merged < SUCCESS_LOSS).sum().item())

    # set-up, not a step: every context runs one batch once so that its device buffers exist and its
    # kernels are loaded (with 8 contexts, W < 8 warm-up steps would leave some of them cold)
    if n_streams > 1:
        prime = [threading.Thread(target=one_step, args=(0, c)) for c in ctxs]
        for t in prime:
            t.start()
        for t in prime:
            t.join()
    res = {}
    run_steps(list(range(warmup)), res)
    sync()
    for c in ctxs:
        c.reset_stats()
    solved = 0
    cyc_hist = np.zeros(4, dtype=np.int64)
    worst = 0.0
    res = {}
    t0 = time.perf_counter()
    run_steps(list(range(warmup, total_steps)), res)
    sync()
    elapsed = time.perf_counter() - t0
    for s in range(warmup, total_steps):
        bl, bc = res[s]
        ok = bl < SUCCESS_LOSS
        solved += int(ok.sum())
        worst = max(worst, float(bl.max()))
        cyc_hist += np.bincount(np.clip(bc, 0, 3), minlength=4)
    per_basis = None
    if sweep:
        # SURVEY.md §8(d) cfg 5 output: per-basis success fraction and mean best_cycles (rank 0's bases)
        per_basis = {}
        for s in range(warmup, total_steps):
            b = basis_of(s)
            if b in per_basis:
                continue
            bl, bc = res[s]
            ok = bl < SUCCESS_LOSS
            per_basis[b] = {"solved_fraction": float(ok.mean()), "mean_cycles": float(bc[ok].mean()) if ok.any() else None}
    sts = [c.stats() for c in ctxs]
    st = {
        "kernel_ms": sum(x["kernel_ms"] for x in sts),
        "kernel_launches": sum(x["kernel_launches"] for x in sts),
        "evals": [sum(x["evals"][k] for x in sts) for k in range(6)],
        "items": [sum(x["items"][k] for x in sts) for k in range(6)],
    }
    streams_used = n_streams

    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        solved_all = res["merged_solved"]  # counted on the all-reduced loss vector (same on every rank)
        cnt = torch.tensor([solved], dtype=torch.float64, device=tdev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        assert int(cnt.item()) == solved_all, "merged best-loss vector disagrees with the per-rank counts"
    else:
        solved_all = solved

    if rank == 0:
        flops = sum(st["evals"][k] * f_eval(k) for k in (1, 2, 3))
        # one batch in flight: launches do not overlap, achieved = flops / sum of HIP-event launch durations.
        # several batches in flight: launches of different streams share the chip and their event
        # durations overlap, so the denominator is the wall time of the timed region instead.
        kernel_s = st["kernel_ms"] * 1e-3 if streams_used == 1 else elapsed
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
                "span_selection": "analytic span rules (use_polytopes mode)" if args.span_rules else "brute force 1..3 (reference default)",
                "success_threshold": threshold,
                "parallelism": (f"bases sharded over {world} GPU(s) ({SWEEP_BASES_PER_GPU} each), targets replicated, no data-path collective"
                                if sweep else f"targets sharded over {world} GPU(s), no data-path collective"),
                "batches_in_flight_per_gpu": streams_used,
                "items_per_quad": ipq,
                "device": dev_name,
                "compute_units": cus,
            },
            "solved_fraction": solved_all / (world * steps * n_per_step),
            "best_cycles_hist_rank0": {str(k): int(cyc_hist[k]) for k in range(4)},
            "worst_loss_rank0": worst,
            **({"per_basis_rank0": {str(b): v for b, v in sorted(per_basis.items())}} if per_basis is not None else {}),
            "roofline": {
                "bound": "valu_fp64",
                "achieved": achieved,
                "peak": PEAK_FP64_VALU_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP64_VALU_TFLOPS,
                "traffic": traffic_per_launch(args.workload),
                "kernel": "minimize_kernel<K> (k=1..3)",
                "time_basis": "hip_events" if streams_used == 1 else "wall_clock_of_timed_region",
                "kernel_ms_total": st["kernel_ms"],
                "kernel_launches": st["kernel_launches"],
                "avg_launch_ms": st["kernel_ms"] / max(1, st["kernel_launches"]),
                "evals_per_span": {str(k): st["evals"][k] for k in (1, 2, 3)},
                "items_per_span": {str(k): st["items"][k] for k in (1, 2, 3)},
                "flops_per_eval": {str(k): f_eval(k) for k in (1, 2, 3)},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(gname, restarts, 20260000, opt_seed, args.cpu_sample, args.host_targets)
        print(json.dumps(out), flush=True)

    for c in ctxs:
        c.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
