"""bench: the `secondary` probes on the line -- CircuitTemplateV2, the drop-in API end to end (one window, five windows, one target,
polytope mode), a medium call alone on the device."""
from __future__ import annotations

import threading
import time

import numpy as np

from .workloads import OPT_SEED, PEAK_FP64_VALU_TFLOPS, SUCCESS_LOSS, TARGET_SEED0, f_eval, f_eval_v2, gate_table


def run_v2(rank: int, local_rank: int, steps: int = 1024, warmup: int = 32, n_targets: int = 4096, restarts: int = 16, n_streams: int = 8, group: int = 32,
           base_gate=None, gate_desc: str = "RiSwapGate"):
    """secondary.v2: CircuitTemplateV2(base_gates=[RiSwapGate]) -- every gate instance with its own free alpha -- SquareCost,
    spans 1..3, `n_targets` Haar targets x `restarts` restarts per step.  The span loop is the one TemplateOptimizer runs for a
    V2 template (optimizer.py:_run_batch_v2 -> slam_v2_decompose_range): enqueued on the device as one chain of kernels per
    step.  Like the configs[1]-sized steps of the fixed-gate path, `group` consecutive steps (windows of one resident array) go
    to the library as ONE call -- one device-side work queue per span over all their items -- on `n_streams` host threads /
    contexts / streams (measured, MI355X: one step per call 2.5e6 decompositions/s / 0.19 of peak, 8 per call 6.4e6 / 0.31; round 4,
    tools/r4_v2_sweep.sh: 64 steps at 8 per call x 4 in flight 6.2e6 / 0.31, 128 steps at 16 x 4 7.6e6 / 0.36, 8 x 8
    6.8e6 / 0.33, 32 x 2 7.3e6 / 0.34, 256 steps at 32 x 4 7.7e6 / 0.345; tools/r4_v2_sweep2.sh: 512 steps at 16 x 8 8.1e6 / 0.37, at 32 x 8
    8.3e6 / 0.37 (a 0.25 s region with eight calls per stream instead of two); round 5, tools/r5_v2_shapes.sh, one box: 512 steps 0.347-0.357,
    1024 steps (the default now: four calls per stream, a 0.5 s region) 0.372; 64 per call x 8 0.353, 32 x 12 0.359, 2048 steps at 64 x 8 0.347)."""
    from slam_decomposition_amd import _ffi
    from slam_decomposition_amd.basisv2 import CircuitTemplateV2
    from slam_decomposition_amd.gates import RiSwapGate

    basis = CircuitTemplateV2(base_gates=[RiSwapGate if base_gate is None else base_gate], maximum_span_guess=3)
    total = steps + warmup
    group = max(1, min(group, steps))
    n_streams = max(1, min(n_streams, (steps + group - 1) // group))
    ctxs = [_ffi.Context(local_rank % max(1, _ffi.device_count())) for _ in range(n_streams)]
    for c in ctxs:
        c.sample_haar(TARGET_SEED0 + 7_000_000 + rank * total * n_targets, total * n_targets)
        c.v2_set_gates(basis._gate_maps)
        c.set_cost(_ffi.COST_SQUARE)
    prm = _ffi.OptParams(restarts=restarts, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=OPT_SEED, flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED)
    threshold = 1e-10
    layouts = {}
    for k in (1, 2, 3):
        basis.build(k)
        layouts[k] = basis.device_layout(k)

    def one_call(s0: int, n_steps: int, ctx):
        # the whole span loop on the device (slam_v2_decompose_range): optimizer kernel + epilogue per template size, no host
        # round trip in between; (best_loss, best_x, best_cycles) of the steps' targets come back at the end
        best, _, cyc = ctx.v2_decompose_range(s0 * n_targets, n_steps * n_targets, 1, 3, [[0] * k for k in (1, 2, 3)],
                                              [layouts[k][2:6] for k in (1, 2, 3)], prm, threshold)
        return best, cyc

    def run(step_ids):
        res = {}
        groups = [step_ids[i : i + group] for i in range(0, len(step_ids), group)]

        def worker(w):
            for g in groups[w::n_streams]:
                best, cyc = one_call(g[0], len(g), ctxs[w])
                for i, s in enumerate(g):
                    res[s] = (best[i * n_targets : (i + 1) * n_targets], cyc[i * n_targets : (i + 1) * n_targets])

        threads = [threading.Thread(target=worker, args=(w,)) for w in range(n_streams)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        return res

    # set-up, not a step: every context runs one call of the timed size (its work buffers are sized by the item count)
    prime = [threading.Thread(target=one_call, args=(0, min(group, total), c)) for c in ctxs]
    for t in prime:
        t.start()
    for t in prime:
        t.join()
    if warmup:
        # the warm-up steps are repeated until the device has been busy for 0.2 s: after a second of host-side set-up the chip
        # idles at a low clock, and a timed region of 40 ms that starts there measures the ramp (seen: half the rate)
        t_w = time.perf_counter()
        while True:
            run(list(range(warmup)))
            if time.perf_counter() - t_w > 0.2:
                break
    for c in ctxs:
        c.synchronize()
        c.reset_stats()
    t0 = time.perf_counter()
    res = run(list(range(warmup, total)))
    for c in ctxs:
        c.synchronize()
    elapsed = time.perf_counter() - t0
    solved = 0
    hist = np.zeros(4, dtype=np.int64)
    for s in range(warmup, total):
        best, cyc = res[s]
        solved += int((best < SUCCESS_LOSS).sum())
        hist += np.bincount(np.clip(cyc, 0, 3), minlength=4)
    sts = [c.stats() for c in ctxs]
    for c in ctxs:
        c.close()
    ev = {k: sum(x["evals"][k] for x in sts) for k in (1, 2, 3)}
    kms_span = {k: sum(x["kernel_ms_span"][k] for x in sts) for k in (1, 2, 3)}
    flops = sum(ev[k] * f_eval_v2(k) for k in (1, 2, 3))
    return {
        "workload": f"CircuitTemplateV2(base_gates=[{gate_desc}]) (free gate parameters per gate instance), SquareCost, spans 1..3, {n_targets} Haar targets x {restarts} restarts per step",
        "value": solved / elapsed, "unit": "decompositions/s", "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
        "batches_in_flight_per_gpu": n_streams, "steps_per_library_call": group,
        "solved_fraction": solved / (steps * n_targets), "best_cycles_hist": {str(k): int(hist[k]) for k in range(4)},
        "roofline_frac": flops / elapsed / 1e12 / PEAK_FP64_VALU_TFLOPS,
        "kernel_ms_per_step_alone_or_overlapped": {str(k): kms_span[k] / steps for k in (1, 2, 3)},
        "evals_per_span": {str(k): ev[k] for k in (1, 2, 3)},
        "flops_per_eval": {str(k): f_eval_v2(k) for k in (1, 2, 3)},
        "flops_note": "F_eval(k) + 488 k: the fixed-gate count plus the gate-angle derivatives (bench.py:f_eval_v2)",
        "span_loop": "on the device (slam_v2_decompose_range): one chain of kernels per library call, results fetched at its end",
    }


def run_api(local_rank: int, n_targets: int = 65536, restarts: int = 32, reps: int = 3, shards: int = 0):
    """secondary.api: the drop-in method north_star names, end to end --
    ``TemplateOptimizer(CircuitTemplate([RiSwapGate(1/2)], maximum_span_guess=3), BasicCost(), training_restarts=32)
    .approximate_from_distribution(DeviceHaarBatch(n_samples=65536))`` (src/slam/optimizer.py:180-186): targets generated on the
    device, ONE blocking call, results back as the reference's 3-tuple (training_loss, coordinate_list, [DataDictEntry]).  Wall
    time of the whole Python call, median of `reps` after one untimed call; a fresh sampler seed per call."""
    from slam_decomposition_amd import _ffi
    from slam_decomposition_amd.basis import CircuitTemplate
    from slam_decomposition_amd.cost_function import BasicCost
    from slam_decomposition_amd.gates import RiSwapGate
    from slam_decomposition_amd.optimizer import TemplateOptimizer
    from slam_decomposition_amd.sampler import DeviceHaarBatch

    device = local_rank % max(1, _ffi.device_count())
    basis = CircuitTemplate(base_gates=[RiSwapGate(0.5)], maximum_span_guess=3, device=device)
    times, solved = [], 0
    for r in range(reps + 1):
        opt = TemplateOptimizer(basis, BasicCost(), training_restarts=restarts, seed=OPT_SEED, override_fail=True,
                                **({"devices": [device] * shards} if shards else {}))
        t0 = time.perf_counter()
        loss, _, data = opt.approximate_from_distribution(DeviceHaarBatch(seed=TARGET_SEED0 + 9_000_000 + r, n_samples=n_targets, device=device))
        dt = time.perf_counter() - t0
        if r:
            times.append(dt)
            solved = int((np.asarray(loss) < SUCCESS_LOSS).sum())
            assert len(data) == n_targets and data[n_targets - 1].cycles in (2, 3)
    times.sort()
    med = times[(len(times) - 1) // 2]
    # the same method on a sampler of 327 680 targets: contiguous shares on helper contexts, each run as successive windows of at most
    # 65 536 targets (TemplateOptimizer._run_batch_windows) -- the drop-in method on the bench's own scheme of batches in flight
    big_n = 5 * n_targets
    btimes, bsolved = [], 0
    for r in range(reps + 1):
        bopt = TemplateOptimizer(basis, BasicCost(), training_restarts=restarts, seed=OPT_SEED, override_fail=True)
        t0 = time.perf_counter()
        bloss, _, bdata = bopt.approximate_from_distribution(DeviceHaarBatch(seed=TARGET_SEED0 + 9_500_000 + r, n_samples=big_n, device=device))
        dt = time.perf_counter() - t0
        if r:
            btimes.append(dt)
            bsolved = int((np.asarray(bloss) < SUCCESS_LOSS).sum())
            assert len(bdata) == big_n and bdata[big_n - 1].cycles in (2, 3)
    btimes.sort()
    bmed = btimes[(len(btimes) - 1) // 2]
    api_large = {"workload": f"TemplateOptimizer.approximate_from_distribution(DeviceHaarBatch(n_samples={big_n})), sqrt(iSWAP) span<=3, {restarts} restarts: "
                             f"one call, {bopt.windows_in_flight} helper contexts in flight, windows of at most {bopt.WINDOW_TARGETS} targets",
                 "value": bsolved / bmed, "unit": "decompositions/s", "wall_ms": 1e3 * bmed, "wall_ms_all": [round(1e3 * t, 3) for t in btimes],
                 "solved_fraction": bsolved / big_n, "windows": len(bopt.last_stats_per_device)}
    del bdata, bloss
    # the reference's atomic call (optimizer.py:65-119): ONE target, the reference's default 5 restarts, spans 1..3 -- latency
    from slam_decomposition_amd.sampler import random_unitary

    lat = []
    for i in range(24):
        one = TemplateOptimizer(basis, BasicCost(), seed=OPT_SEED + i, override_fail=True)
        U = random_unitary(4, seed=TARGET_SEED0 + i)
        t0 = time.perf_counter()
        d = one.approximate_target_U(U)
        lat.append(time.perf_counter() - t0)
    lat = sorted(lat[4:])
    # the same call with use_polytopes=True (basis.py:95-100): every target only at the template size its coverage set assigns -- the
    # lookup runs on the device too (slam_predict_spans), nothing but the results comes back
    pbasis = CircuitTemplate(base_gates=[RiSwapGate(0.5)], maximum_span_guess=3, use_polytopes=True, device=device)
    ptimes, psolved = [], 0
    for r in range(reps + 1):
        popt = TemplateOptimizer(pbasis, BasicCost(), training_restarts=restarts, seed=OPT_SEED, override_fail=True)
        t0 = time.perf_counter()
        ploss, _, pdata = popt.approximate_from_distribution(DeviceHaarBatch(seed=TARGET_SEED0 + 9_000_000 + r, n_samples=n_targets, device=device))
        dt = time.perf_counter() - t0
        if r:
            ptimes.append(dt)
            psolved = int((np.asarray(ploss) < SUCCESS_LOSS).sum())
    ptimes.sort()
    pmed = ptimes[(len(ptimes) - 1) // 2]
    return {"workload": f"TemplateOptimizer.approximate_from_distribution(DeviceHaarBatch(n_samples={n_targets})), sqrt(iSWAP) span<=3, {restarts} restarts, one blocking call",
            "value": solved / med, "unit": "decompositions/s", "wall_ms": 1e3 * med, "wall_ms_all": [round(1e3 * t, 3) for t in times],
            "solved_fraction": solved / n_targets, "kernel_ms": opt.last_stats["kernel_ms"],
            "approximate_target_U_ms": {"median": round(1e3 * lat[len(lat) // 2], 4), "min": round(1e3 * lat[0], 4), "restarts": 5,
                                        "what": "one Haar target per call, wall time of the Python call (speculative spans: all three template sizes side by side)",
                                        "last_loss": float(d.loss_result), "last_cycles": int(d.cycles)},
            "use_polytopes": {"value": psolved / pmed, "unit": "decompositions/s", "wall_ms": 1e3 * pmed, "solved_fraction": psolved / n_targets,
                              "what": "the same call with CircuitTemplate(use_polytopes=True): template sizes from the exact coverage sets, looked up on the device"}}, api_large


def run_medium_call(local_rank: int, n_targets: int = 4096, restarts: int = 16, reps: int = 9):
    """secondary.medium_call: ONE call of a medium batch (CNOT, 4096 x 16: beyond the wave kernels, far from filling the chip for long),
    alone on the device -- the spans of its loop side by side (overlapped spans, the library's own choice at this size) against the
    span-by-span launches (SLAM_FLAG_STAGED); same results bit for bit (tests/test_gpu_round4.py)."""
    from slam_decomposition_amd import _ffi

    ctx = _ffi.Context(local_rank % max(1, _ffi.device_count()))
    ctx.set_gates(gate_table("cx"))
    ctx.sample_haar(TARGET_SEED0 + 77, n_targets)
    seqs = [[0], [0, 0], [0, 0, 0]]
    out = {}
    for name, extra in (("overlapped_spans", 0), ("span_by_span", _ffi.FLAG_STAGED)):
        prm = _ffi.OptParams(restarts=restarts, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=OPT_SEED, flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED | extra)
        ts = []
        for r in range(reps + 2):
            ctx.reset_stats()
            t0 = time.perf_counter()
            loss, _, _ = ctx.decompose_range(0, n_targets, 1, 3, seqs, prm, 1e-10)
            ts.append(time.perf_counter() - t0)
        st = ctx.stats()
        med = sorted(ts[2:])[reps // 2]
        out[name] = {"wall_ms": round(1e3 * med, 4), "kernel_launches": st["kernel_launches"], "solved_fraction": float((loss < SUCCESS_LOSS).mean()),
                     "roofline_frac": sum(st["evals"][k] * f_eval(k) for k in (1, 2, 3)) / med / 1e12 / PEAK_FP64_VALU_TFLOPS}
    ctx.close()
    return {"workload": f"CNOT span<=3, {n_targets} Haar targets x {restarts} restarts, one blocking call alone on the device", **out}


def run_long(local_rank: int, k: int = 8, n_targets: int = 4096, restarts: int = 8, reps: int = 3):
    """secondary.long: one stage of a LONG template -- k = 8 applications of ConversionGainGate(gain pi/16), the kind of circuit the
    reference's MixedOrderBasisCircuitTemplate builds from weak gates (src/slam/basis.py:213-359) -- through the wavefront-per-item
    kernels (csrc/slam_long.hpp: layers over the quads, prefix / suffix scans, the quasi-Newton metric in device memory).  Throughput,
    the fraction of the fp64 peak, and the bytes the metric's pass moves (8 n^2 per evaluation: measured as HBM traffic at 12-16 gates,
    profiles/r5_long_pmc.txt) -- no roofline bar was asked for this family yet."""
    from slam_decomposition_amd import _ffi
    from slam_decomposition_amd.gates import ConversionGainGate

    ctx = _ffi.Context(local_rank % max(1, _ffi.device_count()))
    ctx.set_gates(np.stack([ConversionGainGate(0.0, 0.0, 0.0, np.pi / 16, 1.0).to_matrix()]))
    ctx.sample_haar(TARGET_SEED0 + 555, n_targets)
    prm = _ffi.OptParams(restarts=restarts, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=OPT_SEED, flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED)
    ts, st, out = [], None, None
    for r in range(reps + 1):
        ctx.reset_stats()
        t0 = time.perf_counter()
        out = ctx.minimize_stage([0] * k, prm, want_items=False)
        ts.append(time.perf_counter() - t0)
        st = ctx.stats()
    ctx.close()
    med = sorted(ts[1:])[(reps - 1) // 2]
    flops = st["evals"][k] * f_eval(k)
    return {"workload": f"one stage of a {k}-gate template (n = {6 * (k + 1)} parameters) of ConversionGainGate(gain pi/16), {n_targets} Haar targets x {restarts} "
                        "restarts, ordered early exit: wavefront-per-item kernels (slam_long.hpp)",
            "wall_ms": round(1e3 * med, 3), "kernel_ms": round(st["kernel_ms"], 3), "items": st["items"][k], "evals": st["evals"][k],
            "evals_per_s": st["evals"][k] / med, "targets_per_s": n_targets / med, "solved_fraction": float((out["best_loss"] < SUCCESS_LOSS).mean()),
            "roofline_frac": flops / med / 1e12 / PEAK_FP64_VALU_TFLOPS, "flops_per_eval": f_eval(k),
            "metric_pass_bytes_per_eval": 8 * (6 * (k + 1)) ** 2, "metric_pass_tbps": st["evals"][k] * 8 * (6 * (k + 1)) ** 2 / med / 1e12,
            "note": "throughput of the new kernel family, reported without a roofline bar (VERDICT r4 item 5); metric_pass_tbps = evaluations x 8 n^2 "
                    "bytes (the fp32 metric read and written once per accepted step) / wall time: all of it HBM traffic at 12-16 gates "
                    "(rocprofv3 FETCH_SIZE / WRITE_SIZE, profiles/r5_long_pmc.txt: 4.2 / 5.2 TB/s in steady state), L2 hits below"}
