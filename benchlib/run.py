"""bench: one workload's timed region (run_workload) and the command line (main)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

import numpy as np

from .cpu_baseline import cpu_baseline, parity_sample_size, usable_cores
from .launcher import gather_strings, launch_ranks, make_comm
from .profiles import pmc_figures, traffic_per_launch
from .secondary import run_api, run_long, run_medium_call, run_v2
from .workloads import (OPT_SEED, PEAK_FP64_VALU_TFLOPS, PER_SPAN_WARM_STEPS, SUCCESS_LOSS, SWEEP_BASES_PER_GPU, SWEEP_CPU_BASIS, TARGET_SEED0,
                        WORKLOADS, _batches_in_flight, f_eval, f_forward, gate_table, make_targets, sweep_gate)


def run_workload(args, workload, rank, world, local_rank, comm, steps, warmup, n_streams_arg, main: bool, group_arg: int = 0):
    """Run `warmup` untimed + `steps` timed steps of one workload; returns the dict of measurements."""
    from slam_decomposition_amd import _ffi, parallel

    gname, n_per_step, restarts, desc = WORKLOADS[workload]
    if main and args.targets:
        n_per_step = args.targets
    if main and args.restarts:
        restarts = args.restarts
    strong = main and args.scaling == "strong" and world > 1
    if strong:
        # strong scaling: the batch of ONE GPU's step is split over the ranks (65 536 x 32 over N for the default workload)
        if n_per_step % world:
            raise SystemExit(f"--scaling strong: {n_per_step} targets per step do not divide over {world} ranks")
        n_per_step //= world
        desc += f" -- STRONG scaling: one such batch per step split over {world} GPUs ({n_per_step} targets per GPU)"
    small = n_per_step * restarts <= 65536
    total_steps = steps + warmup
    seed0 = TARGET_SEED0 + rank * total_steps * n_per_step  # disjoint targets per rank (weak scaling)
    # one GPU per rank: LOCAL_RANK; modulo the visible devices, so that a launcher which restricts every rank's
    # visibility to its own GPU (device 0 everywhere) and the shared-GPU rehearsal (SLAM_BENCH_COMM=file) both work
    ndev = _ffi.device_count()
    device = local_rank % max(1, ndev)

    # Small batches are not given a host thread + stream each any more (16 in flight in round 2): `group` consecutive
    # steps -- windows of the same resident array, same basis and seed -- go to the library as ONE call, i.e. one
    # device-side work queue per span over all their (target, restart) items, and come back as per-step slices.  With
    # the ordered early exit every step's results are bit for bit those of its own call (tests/test_gpu_round2.py).
    group = 1
    if (small or (main and args.group)) and not (gname == "cgsweep") and not (main and args.span_rules):
        group = args.group if (main and args.group) else (group_arg or 20)  # measured (320 steps, 4 streams): 10 -> 0.40, 16 -> 0.44, 20 -> 0.445, 32 -> 0.44 of peak
    # the configs[3] shard (32 768 x 16): two steps per call, eight calls in flight -- measured on 80 steps (gpurun_out/r5_cfg4_sweep2.txt): one
    # step per call x 12 in flight 1.24e7 decompositions/s / 0.410 of peak, 2 x 8 1.36e7 / 0.429, 5 x 4 1.33e7 / 0.404, 4 x 6 1.20e7, 8 x 3 1.09e7
    cfg4_shape = main and workload == "cfg4" and not args.group and not n_streams_arg and not args.span_rules and not args.targets and not strong
    if cfg4_shape:
        group = 2
    group = max(1, min(group, steps))
    # (the basis sweep cannot group its steps -- every step has its own gate -- so it keeps more of them in flight; measured on
    # MI355X, 160 steps: 4 in flight 1.42e6 decompositions/s / 0.289 of peak, 8: 1.68e6 / 0.338, 16: 1.82e6 / 0.363)
    sweep = gname == "cgsweep"
    # the basis sweep as ONE chain of kernels per 8 bases (round 4, slam_decompose_multi: per span one multi-queue optimizer launch
    # over the bases' work queues -- a wavefront works on one basis at a time, gates stay scalar operands -- and one bookkeeping
    # launch), 4 such calls in flight; --no-multi: one call per basis, 16 in flight (round 3).  Measured on MI355X (160 steps,
    # tools/r4_cfg5b.sh): bases per call x calls in flight 16 x 2 -> 0.325 of peak, 16 x 4 0.324, 8 x 4 0.385, 8 x 6 0.381,
    # 4 x 8 0.381, 2 x 12 0.375; one call per basis x 16 in flight 0.346.  (A launch over 16 queues runs exactly as fast as one queue
    # of the same total size -- tools/r4_mq_probe.py --; what separates the rows is how well the calls' stage tails overlap.)
    mq = sweep and not (main and args.span_rules) and not (main and args.no_multi)
    if mq:
        group = min(args.group if (main and args.group) else 8, SWEEP_BASES_PER_GPU, steps)
    n_streams = n_streams_arg if n_streams_arg else ((4 if mq else 16) if sweep else (4 if small else _batches_in_flight(n_per_step * restarts, main and args.span_rules)))
    if cfg4_shape:
        n_streams = 8
    n_streams = max(1, min(n_streams, (steps + group - 1) // group))
    ctxs = [_ffi.Context(device) for _ in range(n_streams * (group if mq else 1))]
    dev_name, cus, _ = ctxs[0].device_info()
    table = gate_table(gname)
    host_targets = main and args.host_targets
    span_rules_mode = main and args.span_rules
    stub_mode = bool(os.environ.get("SLAM_BENCH_TEST_STUB"))
    # every batch resident in HBM before the timed region: Haar targets generated in place by the device
    # sampler (slam_sample_haar; --host-targets: SciPy's sampler on the host, ~55 us per target, then uploaded)
    n_resident = n_per_step if sweep else total_steps * n_per_step  # the sweep's targets are shared by all bases
    targets = make_targets(n_resident, seed0 if not sweep else TARGET_SEED0) if host_targets else None
    def basis_of(s: int) -> int:
        # rank r takes column p = r of the (m, p) grid: all 16 strengths m, so every GPU has weak and strong gates
        return (s % SWEEP_BASES_PER_GPU) * 8 + rank % 8

    ctx_basis = {}
    for i, c in enumerate(ctxs):
        # multi-queue sweep: context i of a call's group serves the steps with s mod group == i mod group -- with 16 steps per call
        # that is ONE basis for the whole run (its gate is set once, here)
        if mq:
            ctx_basis[id(c)] = basis_of(i % group)
        c.set_gates(np.stack([sweep_gate(basis_of(i % group))]) if mq else table)
        if host_targets:
            c.set_targets(targets)
        else:
            c.sample_haar(seed0 if not sweep else TARGET_SEED0, n_resident)
    gate_seqs = [[i % len(table) for i in range(k)] for k in (1, 2, 3)]
    ipq = args.items_per_quad if (main and args.items_per_quad >= 0) else (3 if (small and group == 1 and n_streams > 1) else 0)
    flags = _ffi.FLAG_EARLY_EXIT | (0 if args.fast_exit else _ffi.FLAG_ORDERED)
    if n_streams > 1:
        flags |= _ffi.FLAG_NO_OVERLAP  # several calls in flight fill the chip: no speculative stages beside them
    elif not args.fast_exit and not sweep and not span_rules_mode and not args.staged:
        # ONE call at a time alone on the device: TemplateOptimizer's rule (optimizer.py:_overlap_pays) -- the spans side by side
        # for all targets of the call (SLAM_FLAG_OVERLAP, same results bit for bit) when the exact coverage regions say that a good
        # share of the batch needs the last span anyway (CNOT: every Haar target; sqrt(iSWAP): a fifth; B: none).  Measured:
        # `--workload cfg2 --steps 20` 0.327 -> 0.347, `--streams 1` on the default workload 18.7 -> 17.6 ms per step.
        from slam_decomposition_amd.optimizer import TemplateOptimizer
        from slam_decomposition_amd.weyl import c1c2c3 as _c1c2c3

        try:
            spans3 = ctxs[0].predict_spans([_c1c2c3(table[i]) for i in gate_seqs[2]], 3, 0, min(n_per_step, 65536))
            if float(np.mean(spans3 >= 3)) >= TemplateOptimizer.OVERLAP_MIN_TOP_SHARE:
                flags |= _ffi.FLAG_OVERLAP
        except (NotImplementedError, ValueError):
            pass
    prm = _ffi.OptParams(restarts=restarts, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=OPT_SEED, flags=flags, items_per_quad=ipq)
    threshold = 1e-10  # reference SUCCESS_THRESHOLD (optimizer.py:18); the metric counts loss < 1e-8

    if span_rules_mode:
        from slam_decomposition_amd.weyl import c1c2c3 as host_c1c2c3

    def one_step(s: int, c):
        if span_rules_mode:
            # use_polytopes=True (basis.py:95-100): every target starts at the template size its coverage set assigns -- exact for every
            # gate sequence (coverage.py: the monodromy inequalities; the half-spaces of the three prefixes go to the device, which
            # evaluates them on the resident targets: slam_predict_spans) -- and runs on from there like the span loop does.
            # tol: the metric accepts loss < 1e-8, i.e. targets up to ~1e-4 in coordinates outside the exact reachable set
            first = 0 if sweep else s * n_per_step
            if sweep:
                g = sweep_gate(basis_of(s))
                c.set_gates(np.stack([g]))
                seq_coords = [host_c1c2c3(g)] * 3
            else:
                seq_coords = [host_c1c2c3(table[i]) for i in gate_seqs[2]]
            # round 5: lookup, per-size lists and the span loop in ONE chain of kernels (slam_decompose_predicted, carry: a target that
            # misses the threshold at its size goes on to the next) -- round 4 built the lists on the host, one call per size.
            # Local targets come back as (0, 0), targets beyond the whole template's reach as (+inf, -1).
            c.decompose_predicted(seq_coords, 3, gate_seqs, prm, threshold, first, n_per_step, carry=True, tol=5e-4)
            best_loss, best_x, best_cycles = c.fetch_results_range(3, first, n_per_step)
            return best_loss, best_cycles
        if sweep:
            c.set_gates(np.stack([sweep_gate(basis_of(s))]))
            best_loss, best_x, best_cycles = c.decompose_range(0, n_per_step, 1, 3, gate_seqs, prm, threshold)
        else:
            best_loss, best_x, best_cycles = c.decompose_range(s * n_per_step, n_per_step, 1, 3, gate_seqs, prm, threshold)
        return best_loss, best_cycles

    def sync():
        # barrier + device synchronisation on both sides of the timed region (all streams of this rank drained,
        # then all ranks arrived, then drained again)
        for c in ctxs:
            c.synchronize()
        comm.barrier()
        for c in ctxs:
            c.synchronize()

    resident_merge = world > 1 and hasattr(comm, "raw") and not sweep and not span_rules_mode

    def run_steps(step_ids, results, first_step):
        # steps are dealt round-robin to n_streams host threads, each with its own context / HIP stream,
        # so the tail of one batch (a stage lasts as long as its slowest work item) overlaps the next batch
        groups = parallel.step_groups(step_ids, group)

        def worker(w):
            for g in groups[w::n_streams]:
                if mq:
                    # one library call for up to 16 consecutive steps = 16 different bases (step s -> basis slot s mod 16)
                    cg = ctxs[w * group : (w + 1) * group]
                    sub = [cg[s % group] for s in g]
                    for s, c in zip(g, sub):
                        if ctx_basis[id(c)] != basis_of(s):  # (fewer than 16 steps per call: the slot changes its basis)
                            c.set_gates(np.stack([sweep_gate(basis_of(s))]))
                            ctx_basis[id(c)] = basis_of(s)
                    _ffi.decompose_multi(sub, 0, n_per_step, 1, 3, gate_seqs, prm, threshold)
                    for s, c in zip(g, sub):
                        bl, _, bc = c.fetch_results_range(3, 0, n_per_step)
                        results[s] = (bl, bc)
                    continue
                if len(g) == 1 or g != list(range(g[0], g[0] + len(g))):
                    for s in g:
                        results[s] = one_step(s, ctxs[w])
                    continue
                # one library call for the whole group of consecutive steps, results handed back per step
                bl, _, bc = ctxs[w].decompose_range(g[0] * n_per_step, len(g) * n_per_step, 1, 3, gate_seqs, prm, threshold)
                for i, s in enumerate(g):
                    results[s] = (bl[i * n_per_step : (i + 1) * n_per_step], bc[i * n_per_step : (i + 1) * n_per_step])

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
            if resident_merge:
                # device to device: each context's resident best_loss windows -> this rank's slice of the job vector
                comm.raw.merge_begin(world * n_loc)
                for w, local_first, cnt, global_first in parallel.merge_slices(step_ids, first_step, n_per_step, rank, world, n_streams, group):
                    comm.raw.merge_add(ctxs[w], local_first, cnt, global_first)
                t_c = time.perf_counter()
                results["merged_solved"], _ = comm.raw.allreduce_min_merged(SUCCESS_LOSS)
                results["collective_ms"] = 1e3 * (time.perf_counter() - t_c)
            else:
                merged = np.full(world * n_loc, np.inf)
                merged[rank * n_loc : (rank + 1) * n_loc] = np.concatenate([results[s][0] for s in step_ids])
                t_c = time.perf_counter()
                comm.allreduce_min(merged)
                results["collective_ms"] = 1e3 * (time.perf_counter() - t_c)
                results["merged_solved"] = int((merged < SUCCESS_LOSS).sum())

    # set-up, not a step: every context runs one batch once so that its device buffers exist and its
    # kernels are loaded (with 8 contexts, W < 8 warm-up steps would leave some of them cold)
    def prime_one(c):
        if mq:
            w = ctxs.index(c) // group
            _ffi.decompose_multi(ctxs[w * group : (w + 1) * group], 0, n_per_step, 1, 3, gate_seqs, prm, threshold)
        elif group > 1:
            # the grouped call's work buffers are sized by its item count: allocate them here, not inside the timed region
            # (W < group warm-up steps would make a smaller call), and bring the device out of its idle clock state
            c.decompose_range(0, min(group, total_steps) * n_per_step, 1, 3, gate_seqs, prm, threshold)
        else:
            one_step(0, c)

    if n_streams > 1 or group > 1:
        prime = [threading.Thread(target=prime_one, args=(c,)) for c in (ctxs[::group] if mq else ctxs)]
        for t in prime:
            t.start()
        for t in prime:
            t.join()
    def sum_stats():
        sts = [c.stats() for c in ctxs]
        out = {"kernel_ms": sum(x["kernel_ms"] for x in sts), "kernel_launches": sum(x["kernel_launches"] for x in sts)}
        for key in ("evals", "items", "evals_accepted", "evals_preempted", "kernel_ms_span", "wave_rounds"):
            out[key] = [sum(x[key][k] for x in sts) for k in range(6)]
        return out

    res = {}
    if warmup:
        run_steps(list(range(warmup)), res, 0)
    # The timed region -- exactly `steps` steps between barrier + drained streams on both sides, MAX over ranks -- is
    # repeated `reps` times on the same resident batches (identical work every time); the line reports the MEDIAN
    # repetition and the spread, so that box noise shows in a single run of the command.
    reps = max(1, args.repeats)
    rep_runs = []
    for _ in range(reps):
        sync()
        for c in ctxs:
            c.reset_stats()
        res = {}
        t0 = time.perf_counter()
        run_steps(list(range(warmup, total_steps)), res, warmup)
        res["own_ms"] = 1e3 * (time.perf_counter() - t0)  # this rank's own steps + the collective, before the closing barrier
        sync()
        tt = np.array([time.perf_counter() - t0])
        comm.allreduce_max(tt)  # max over ranks of the time
        rep_runs.append((float(tt[0]), res, sum_stats()))
    order = sorted(range(reps), key=lambda i: rep_runs[i][0])
    elapsed, res, st = rep_runs[order[(reps - 1) // 2]]
    rep_ms = [1e3 * r[0] / steps for r in rep_runs]

    solved = 0
    cyc_hist = np.zeros(4, dtype=np.int64)
    worst = 0.0
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

    # solved targets counted on the all-reduced vector (same on every rank)
    rank_diag = None
    if world > 1:
        solved_all = res["merged_solved"]
        cnt = np.array([float(solved)])
        comm.allreduce_sum(cnt)
        assert int(cnt[0]) == solved_all, "merged best-loss vector disagrees with the per-rank counts"
        # what a first N > 1 run needs to diagnose itself: every rank's own time for the median repetition's timed region
        # (steps + collective, before the closing barrier), its solved count, the collective's duration, its evaluations
        cores, quota = usable_cores()
        flat = np.zeros(world * 6)
        flat[6 * rank : 6 * rank + 6] = [res["own_ms"], float(solved), res.get("collective_ms", 0.0), float(sum(st["evals"][k] for k in (1, 2, 3))),
                                         float(threading.active_count()), float(quota if quota is not None else cores)]
        comm.allreduce_sum(flat)
        diag = flat.reshape(world, 6)
        rank_diag = {"own_ms": [round(float(v), 3) for v in diag[:, 0]], "solved": [int(v) for v in diag[:, 1]],
                     "collective_ms": [round(float(v), 3) for v in diag[:, 2]], "evals": [int(v) for v in diag[:, 3]],
                     "host_threads": [len(ctxs) // (group if mq else 1) + 1] * world, "live_threads_at_report": [int(v) for v in diag[:, 4]],
                     "cpu_share": [round(float(v), 2) for v in diag[:, 5]],
                     "note": "per rank, median repetition: wall time of its own steps + the final collective (before the closing barrier), "
                             "targets it solved, duration of the collective as it saw it, loss+gradient evaluations; host_threads = worker "
                             "threads (one per call in flight) + the main one, cpu_share = the cores this rank's cgroup / affinity grants: "
                             "N ranks x host_threads on one node must fit the node's cores or the calls in flight starve each other"}
    else:
        solved_all = solved

    # per-span pass, ONE batch in flight: launches do not overlap, so every frac below is evals x F_eval(k) / the HIP
    # events around that launch -- the figure `rocprofv3 --kernel-trace --stats` reports for the same launches
    # (tools/r4_trace_summary.py picks them out of the driver command's trace).  Not part of `value`.
    # The steps are enqueued BACK TO BACK on one stream (no result fetch in between: the 13 MB copy into pageable memory
    # leaves the chip idle for ~1 ms, and the launch after an idle gap runs 2-10 % slower while the clock ramps:
    # profiles/r4_solo_probe.txt), after one untimed step of the same kind.
    per_span = None
    if main and rank == 0 and not span_rules_mode and args.per_span_steps > 0:
        c = ctxs[0]
        # (span by span whatever the timed region asked for: every launch alone on the chip)
        prm_solo = _ffi.OptParams(restarts=restarts, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=OPT_SEED,
                                  flags=(flags & ~_ffi.FLAG_OVERLAP) | _ffi.FLAG_NO_OVERLAP, items_per_quad=ipq)

        def solo_step(s):
            if sweep:
                c.set_gates(np.stack([sweep_gate(basis_of(s))]))
                c.decompose_range(0, n_per_step, 1, 3, gate_seqs, prm_solo, threshold, fetch=False)
            else:
                c.decompose_range(s * n_per_step, n_per_step, 1, 3, gate_seqs, prm_solo, threshold, fetch=False)

        for s in range(PER_SPAN_WARM_STEPS):  # untimed: the chip's clocks settle under this load (the first launches after an idle
            solo_step(s % total_steps)        # period run 2-3 % slower: profiles/r4_solo_probe.txt)
        rows = []
        for s in range(args.per_span_steps):
            c.reset_stats()
            solo_step(s % total_steps)
            rows.append(c.stats())
        per_span = {}
        for k in (1, 2, 3):
            rk = [r for r in rows if r["items"][k] and r["kernel_ms_span"][k] > 0]
            if not rk:
                continue
            ms = sum(r["kernel_ms_span"][k] for r in rk)
            ev = sum(r["evals"][k] for r in rk)
            per_span[str(k)] = {
                "launches": len(rk),
                "evals_per_launch": ev / len(rk),
                "hip_event_ms": ms / len(rk),
                "hip_event_ms_all": [round(r["kernel_ms_span"][k], 4) for r in rk],
                "achieved": ev * f_eval(k) / (ms * 1e-3) / 1e12,
                "frac": ev * f_eval(k) / (ms * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS,
                "quad_occupancy": ev / 16 / max(1, sum(r["wave_rounds"][k] for r in rk)),
            }
        tot_ms = sum(r["kernel_ms_span"][k] for r in rows for k in (1, 2, 3))
        tot_fl = sum(r["evals"][k] * f_eval(k) for r in rows for k in (1, 2, 3))
        per_span["all"] = {"hip_event_ms_per_step": tot_ms / len(rows),
                           "frac": tot_fl / (tot_ms * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS if tot_ms > 0 else None,
                           "warm_steps": PER_SPAN_WARM_STEPS,
                           "mode": "one batch in flight, steps enqueued back to back on one stream, no result fetch in between"}

    # parity sample: the HIP path's answers for the targets the CPU baseline solves (same indices of the resident array, same
    # Philox start points); with the ordered early exit a window's results do not depend on what else is in the call
    gpu_sample = None
    if main and rank == 0 and world == 1 and not args.no_cpu_baseline and not span_rules_mode and not host_targets and not stub_mode:
        n_s = min(parity_sample_size(args.cpu_sample), n_per_step)
        c = ctxs[0]
        if sweep:
            c.set_gates(np.stack([sweep_gate(SWEEP_CPU_BASIS)]))
        g_loss, g_x, g_cyc = c.decompose_range(0, n_s, 1, 3, gate_seqs, prm, threshold)
        g_coords = np.full((n_s, 3), np.nan)
        for k in np.unique(g_cyc):
            k = int(k)
            if k < 1:
                continue
            sel = np.nonzero(g_cyc == k)[0]
            g_coords[sel] = c.eval_c1c2c3(gate_seqs[k - 1], np.ascontiguousarray(g_x[sel, : 6 * (k + 1)]), ndigits=-1)
        gpu_sample = (g_loss, g_cyc, g_coords, c.targets_c1c2c3(0, n_s, ndigits=-1), c.fetch_span_losses(0, n_s))

    for c in ctxs:
        c.close()

    flops = sum(st["evals"][k] * f_eval(k) for k in (1, 2, 3))
    rejected = [st["evals"][k] - st["evals_accepted"][k] - st["evals_preempted"][k] for k in range(6)]
    flops_accepted = sum(st["evals_accepted"][k] * f_eval(k) for k in (1, 2, 3))
    flops_strict = flops_accepted + sum(rejected[k] * f_forward(k) for k in (1, 2, 3))
    # one batch in flight: launches do not overlap, achieved = flops / sum of HIP-event launch durations.
    # several batches in flight: launches of different streams share the chip and their event
    # durations overlap, so the denominator is the wall time of the timed region instead.
    kernel_s = st["kernel_ms"] * 1e-3 if n_streams == 1 else elapsed
    achieved = flops / kernel_s / 1e12 if kernel_s > 0 else 0.0
    return {
        "desc": desc, "gname": gname, "n_per_step": n_per_step, "restarts": restarts, "threshold": threshold, "sweep": sweep,
        "n_streams": n_streams, "ipq": ipq, "dev_name": dev_name, "cus": cus, "elapsed": elapsed, "solved_all": solved_all,
        "cyc_hist": cyc_hist, "worst": worst, "per_basis": per_basis, "st": st, "achieved": achieved, "kernel_s": kernel_s,
        "rejected": rejected, "flops_accepted": flops_accepted, "flops_strict": flops_strict, "per_span": per_span,
        "resident_merge": resident_merge, "rep_ms": rep_ms, "strong": strong, "group": group, "gpu_sample": gpu_sample,
        "rank_diag": rank_diag, "mq": mq,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; 320 for cfg2-sized batches)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 5; 32 for cfg2-sized batches)")
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS),
                    help="cfg3 = BASELINE configs[2] (default: the largest single-GPU configuration); cfg2 = configs[1]; "
                         "cfg4 / cfg5 = one GPU's shard of configs[3] / configs[4] (targets / bases sharded over --gpus ranks)")
    ap.add_argument("--targets", type=int, default=None, help="override targets per step per GPU")
    ap.add_argument("--restarts", type=int, default=None)
    ap.add_argument("--streams", type=int, default=None,
                    help="batches in flight per GPU (one host thread + context + HIP stream each); default 16 for cfg2-sized batches, 5 otherwise "
                         "(measured on cfg3: 3 -> 2.95e6, 4 -> 3.0e6, 5 -> 3.19e6, 6 -> 3.20e6 decompositions/s; batches below 2^20 items per span, "
                         "e.g. the cfg4 shard: 8 -- round 4, tools/r4_cfg4_sweep.sh: 5 -> 1.03e7, 8 -> 1.16e7; by batch size up to 16: _batches_in_flight)")
    ap.add_argument("--group", type=int, default=0,
                    help="small batches: consecutive steps handed to the library as one call = one device-side work queue per span "
                         "(default 20 for cfg2-sized batches, 1 otherwise)")
    ap.add_argument("--items-per-quad", type=int, default=int(os.environ.get("SLAM_BENCH_IPQ", "-1")),
                    help="launch shaping (slam_opt_params.items_per_quad); 0 = library default (one item per quad, lowest "
                         "latency); default here: 3 for small batches with several in flight (+4.7 %% measured), else 0")
    ap.add_argument("--host-targets", action="store_true", help="draw the Haar targets with SciPy on the host instead of on the device")
    ap.add_argument("--span-rules", action="store_true",
                    help="polytope mode (CircuitTemplate(use_polytopes=True)): each target is optimised only at the template "
                         "size the analytic span rules assign to it (device c1c2c3 + span_rules.py) instead of spans 1..3")
    ap.add_argument("--fast-exit", action="store_true",
                    help="drop SLAM_FLAG_ORDERED: the first restart to FINISH below stop_loss wins (timing-dependent winner) instead of "
                         "the lowest-index successful restart (the reference's sequential semantics, bitwise reproducible; default)")
    ap.add_argument("--repeats", type=int, default=3,
                    help="repetitions of the timed region (each exactly --steps steps between barriers); the line reports the median one and min / max")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every GPU gets its own full-size batches; strong: one GPU's batch per step is split over the --gpus ranks")
    ap.add_argument("--no-multi", action="store_true", help="cfg5: one library call per basis, 16 in flight (round 3) instead of slam_decompose_multi")
    ap.add_argument("--staged", action="store_true", help="one call in flight: never ask for the spans side by side (SLAM_FLAG_OVERLAP is asked for where the batch's coverage says it pays)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="targets of the CPU baseline sample (default 4 x host cores)")
    ap.add_argument("--per-span-steps", type=int, default=5, help="steps of the single-stream per-span roofline pass after the timed region (0 = skip)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary cfg2 / v2 measurements")
    ap.add_argument("--v2-only", action="store_true", help="dev: run only the secondary.v2 measurement (CircuitTemplateV2) and print it")
    ap.add_argument("--long-only", action="store_true", help="dev: run only the secondary.long measurement (templates of 6..16 gates) and print it")
    ap.add_argument("--api-only", action="store_true", help="dev: run only the secondary.api measurement (TemplateOptimizer.approximate_from_distribution) and print it")
    args = ap.parse_args()

    stub = os.environ.get("SLAM_BENCH_TEST_STUB")
    if stub:
        # TEST HOOK (tests/test_bench_cpu.py): a stand-in for _ffi.Context so that the launcher / rank / communicator /
        # JSON plumbing of the N > 1 path can be exercised on a box without a GPU.  Its numbers mean nothing; the line
        # says so in "data".  Never set outside the tests.
        import importlib.util

        spec = importlib.util.spec_from_file_location("slam_bench_test_stub", stub)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.install()

    if args.v2_only:
        kw = {k: int(os.environ[e]) for k, e in (("steps", "SLAM_V2_STEPS"), ("group", "SLAM_V2_GROUP"), ("n_streams", "SLAM_V2_STREAMS"), ("n_targets", "SLAM_V2_TARGETS")) if e in os.environ}
        print(json.dumps(run_v2(0, 0, **kw)), flush=True)  # (dev: SLAM_V2_STEPS / _GROUP / _STREAMS / _TARGETS override the defaults)
        return
    if args.long_only:
        for k in (6, 8, 12, 16):
            print(json.dumps(run_long(0, k=k)), flush=True)
        return
    if args.api_only:
        r, big = run_api(0)
        print(json.dumps({"value": r["value"], "wall_ms_all": r["wall_ms_all"], "kernel_ms": r["kernel_ms"],
                          "approximate_target_U_ms": r["approximate_target_U_ms"]["median"], "use_polytopes": r["use_polytopes"], "api_large": big}), flush=True)
        return
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and (args.gpus > 1 or os.environ.get("SLAM_BENCH_FORCE_LAUNCH")):
        # plain `python bench.py --gpus N`: become the launcher (no GPU call has happened in this process)
        raise SystemExit(launch_ranks(args.gpus))
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch exactly one rank per GPU")

    _, n_default, r_default, _ = WORKLOADS[args.workload]
    small = (args.targets or n_default) * (args.restarts or r_default) <= 65536
    # default length of the timed region by workload: at least a quarter of a second of device time (the cfg4 shard's 20 steps were a 58 ms
    # region with calls of 35 ms latency in flight: a fifth of it was the pipeline filling and draining -- 1.13e7 against 1.24e7 on 80 steps)
    by_workload = {"cfg4": (80, 8), "cfg5": (160, 16)}
    d_steps, d_warm = by_workload.get(args.workload, (320, 32) if small else (20, 5))
    if small:
        d_steps, d_warm = 320, 32
    steps = args.steps if args.steps is not None else d_steps
    warmup = args.warmup if args.warmup is not None else d_warm

    comm = make_comm(rank, world, local_rank)
    m = run_workload(args, args.workload, rank, world, local_rank, comm, steps, warmup, args.streams, main=True)
    secondary = None
    if not args.no_secondary and args.workload == "cfg3" and not args.span_rules and not args.targets and not args.restarts:
        # BASELINE configs[1] (1024 x 16 CNOT, the latency-bound small-batch regime) beside the headline
        s2 = run_workload(args, "cfg2", rank, world, local_rank, comm, 320, 32, None, main=False)
        fl2 = sum(s2["st"]["evals"][k] * f_eval(k) for k in (1, 2, 3))
        secondary = {"cfg2": {
            "workload": s2["desc"], "value": s2["solved_all"] / s2["elapsed"], "unit": "decompositions/s", "steps": 320, "warmup": 32,
            "ms_per_step": 1e3 * s2["elapsed"] / 320, "batches_in_flight_per_gpu": s2["n_streams"], "steps_per_library_call": s2["group"],
            "items_per_quad": s2["ipq"],
            "solved_fraction": s2["solved_all"] / (world * 320 * s2["n_per_step"]),
            "roofline_frac": fl2 / s2["elapsed"] / 1e12 / PEAK_FP64_VALU_TFLOPS,
        }}
        # the same batches ONE per library call, one call in flight: the latency of a lone small batch (round 4: the whole span loop of
        # a target in one wavefront, one launch -- span_wave_kernel)
        s3 = run_workload(args, "cfg2", rank, world, local_rank, comm, 40, 8, 1, main=False, group_arg=1)
        fl3 = sum(s3["st"]["evals"][k] * f_eval(k) for k in (1, 2, 3))
        secondary["cfg2"]["one_batch_per_call"] = {"value": s3["solved_all"] / s3["elapsed"], "ms_per_step": 1e3 * s3["elapsed"] / 40,
                                                   "roofline_frac": fl3 / s3["elapsed"] / 1e12 / PEAK_FP64_VALU_TFLOPS,
                                                   "kernel_launches_per_step": s3["st"]["kernel_launches"] / 40}
        if rank == 0 and not os.environ.get("SLAM_BENCH_TEST_STUB"):
            secondary["v2"] = run_v2(rank, local_rank)
            secondary["api"], secondary["api_large"] = run_api(local_rank)
            secondary["medium_call"] = run_medium_call(local_rank)
            secondary["long"] = run_long(local_rank)

    rank_devices = gather_strings(comm, rank, world, f"{m['dev_name'].strip()} cu={m['cus']} dev={local_rank}")
    if rank == 0:
        st = m["st"]
        n_launch = max(1, st["kernel_launches"])
        pmc = pmc_figures(args.workload)
        rep_ms = sorted(m["rep_ms"])
        out = {
            "metric": "Haar 2-qubit decompositions/sec (span<=3, loss<1e-8)",
            "value": m["solved_all"] / m["elapsed"],
            "unit": "decompositions/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": 1e3 * m["elapsed"] / steps,
            # the timed region (exactly `steps` steps between barriers) ran `repetitions` times on the same batches;
            # value / ms_per_step / roofline are the MEDIAN repetition's, min / max show the box noise of this run
            "repetitions": len(rep_ms),
            "ms_per_step_min": rep_ms[0],
            "ms_per_step_max": rep_ms[-1],
            "ms_per_step_all": m["rep_ms"],
            "higher_is_better": True,
            "scaling": "strong" if m["strong"] else "weak",
            # what RCCL itself reports for the communicator the collective ran on (ncclCommCount); None: no RCCL in this run
            "rccl_world": getattr(comm, "rccl_world", None),
            "comm": type(comm).__name__,
            "rank_devices": rank_devices,
            "vs_baseline": None,
            "dtype": "f64",  # loss, gradient, parameters, steps: every counted flop; see config.metric_dtype
            "data": "synthetic" if not os.environ.get("SLAM_BENCH_TEST_STUB") else "STUB: test hook, no GPU work was done, numbers are meaningless",
            "config": {
                "workload": m["desc"],
                "basis": m["gname"],
                "metric_dtype": "f32 (inverse-Hessian preconditioner of the quasi-Newton iteration only; not in the flop count)",
                "targets_per_step_per_gpu": m["n_per_step"],
                "restarts": m["restarts"],
                "span_max": 3,
                "span_selection": ("use_polytopes mode: every target starts at the template size its exact coverage set assigns (coverage.py: "
                                   "the monodromy inequalities; looked up on the device, slam_predict_spans)") if args.span_rules else "brute force 1..3 (reference default)",
                "success_threshold": m["threshold"],
                "restart_early_exit": "first restart to finish below stop_loss wins (timing-dependent)" if args.fast_exit
                else "ordered: lowest-index successful restart wins (reference semantics, bitwise reproducible)",
                "parallelism": (f"bases sharded over {world} GPU(s) ({SWEEP_BASES_PER_GPU} each), targets replicated, no data-path collective"
                                if m["sweep"] else f"targets sharded over {world} GPU(s), one process per GPU, no data-path collective"),
                "final_collective": (None if world == 1 else
                                     ("slam_allreduce_min: ncclAllReduce(min) of the resident best-loss windows, device to device (RCCL via C ABI)"
                                      if m["resident_merge"] else f"min-all-reduce of the best-loss vector ({type(comm).__name__})")),
                "batches_in_flight_per_gpu": m["n_streams"],
                "steps_per_library_call": m["group"],
                **({"library_call": "slam_decompose_multi: one multi-queue optimizer launch per span over the call's bases"} if m["mq"] else {}),
                "items_per_quad": m["ipq"],
                "device": m["dev_name"],
                "compute_units": m["cus"],
            },
            "solved_fraction": m["solved_all"] / (world * steps * m["n_per_step"]),
            "best_cycles_hist_rank0": {str(k): int(m["cyc_hist"][k]) for k in range(4)},
            "worst_loss_rank0": m["worst"],
            **({"per_basis_rank0": {str(b): v for b, v in sorted(m["per_basis"].items())}} if m["per_basis"] is not None else {}),
            "roofline": {
                "bound": "valu_fp64",
                "achieved": m["achieved"],
                "peak": PEAK_FP64_VALU_TFLOPS,
                "unit": "TFLOP/s",
                "frac": m["achieved"] / PEAK_FP64_VALU_TFLOPS,
                # the DOMINANT kernel alone on the chip (minimize_kernel<1>: 64 % of the flops): algorithmic flops of a launch / the
                # HIP events around it, mean over the back-to-back single-stream launches of `per_span` (= rocprofv3's average for
                # those launches, profiles/r4_trace_summary.json).  `frac` above is the whole job on the wall clock.
                "frac_kernel": (m["per_span"] or {}).get("1", {}).get("frac"),
                "kernel_dominant": "minimize_kernel<1, *>",
                "traffic": traffic_per_launch(args.workload),
                # north_star's two evidence figures, per span, from the committed PMC passes of this workload
                "valu_busy": pmc["valu_busy"] if pmc else None,
                "hbm_gbps": pmc["hbm_gbps"] if pmc else None,
                "pmc_source": pmc["source"] if pmc else None,
                "kernel": "minimize_kernel<K> (k=1..3)",
                "scope": "rank 0's GPU",
                "time_basis": "hip_events" if m["n_streams"] == 1 else "wall_clock_of_timed_region",
                "kernel_ms_total": st["kernel_ms"],
                "kernel_ms_span": {str(k): st["kernel_ms_span"][k] for k in (1, 2, 3)},
                "kernel_launches": st["kernel_launches"],
                "avg_launch_ms": st["kernel_ms"] / n_launch,
                "evals_per_span": {str(k): st["evals"][k] for k in (1, 2, 3)},
                "evals_accepted": {str(k): st["evals_accepted"][k] for k in (1, 2, 3)},
                "evals_rejected": {str(k): m["rejected"][k] for k in (1, 2, 3)},
                "evals_preempted": {str(k): st["evals_preempted"][k] for k in (1, 2, 3)},
                "items_per_span": {str(k): st["items"][k] for k in (1, 2, 3)},
                "flops_per_eval": {str(k): f_eval(k) for k in (1, 2, 3)},
                # the same time, stricter numerators: accepted points only; accepted at F_eval + rejected line-search
                # trials at the forward-only count, pre-empted restarts not counted at all
                "frac_accepted": m["flops_accepted"] / m["kernel_s"] / 1e12 / PEAK_FP64_VALU_TFLOPS,
                "frac_accepted_plus_rejected_forward": m["flops_strict"] / m["kernel_s"] / 1e12 / PEAK_FP64_VALU_TFLOPS,
                "peak_note": "nominal: 256 CU x 4 SIMD x 16 fp64 FMA lanes/clk x 2 x 2.4 GHz; a pure v_fma_f64 loop sustains 58.8 TFLOP/s on this "
                             "chip (it holds ~1.65 GHz under that load: tools/ubench_valu.hip, profiles/r2_ubench_valu.txt)",
                "numerator_note": "dense flop accounting of SURVEY.md 8(d) for every lock-step evaluation; structured gates (CX = a swap) "
                                  "count at the dense 4x4 product's cost",
                **({"per_span": m["per_span"]} if m["per_span"] else {}),
            },
        }
        if secondary:
            out["secondary"] = secondary
        if m["rank_diag"]:
            out["rank_diag"] = m["rank_diag"]
        parity = None
        if world == 1 and not args.no_cpu_baseline:
            parity, out["cpu_baseline"] = cpu_baseline(m["gname"], m["restarts"], TARGET_SEED0, OPT_SEED, args.cpu_sample, args.host_targets,
                                                       gpu_sample=m["gpu_sample"])
            if parity is not None:
                out["parity_sample"] = parity
        print(json.dumps(out), flush=True)
        if parity is not None and not parity["pass"]:
            print(f"[bench] parity sample FAILED: {parity}", file=sys.stderr, flush=True)
            comm.close()
            raise SystemExit(4)
    comm.close()
