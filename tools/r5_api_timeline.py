"""Dev tool (GPU): where the wall time of approximate_from_distribution on a 327 680-target sampler goes -- per window the times of
(sampler fill, span loop + fetch) relative to the call's start, with and without the result fetch."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slam_decomposition_amd import _ffi, runtime
from slam_decomposition_amd.basis import CircuitTemplate
from slam_decomposition_amd.cost_function import BasicCost
from slam_decomposition_amd.gates import RiSwapGate
from slam_decomposition_amd.optimizer import TemplateOptimizer
from slam_decomposition_amd.sampler import DeviceHaarBatch

N = 327680
FL = [int(a) for a in sys.argv[1:] if a.isdigit()] or [4]  # helpers per repetition, cycled
KEEP = "keep" in sys.argv  # hold on to every call's results (nothing is freed between the calls)
kept = []
basis = CircuitTemplate(base_gates=[RiSwapGate(0.5)], maximum_span_guess=3)
log = []
t0 = [0.0]
orig = _ffi.Context.decompose_range
mode = {"fetch": True}
def timed(self, first, count, k_min, k_max, gate_seqs, params, thr, fetch=True):
    a = time.perf_counter() - t0[0]
    if mode["fetch"]:
        out = orig(self, first, count, k_min, k_max, gate_seqs, params, thr, fetch=True)
    else:
        orig(self, first, count, k_min, k_max, gate_seqs, params, thr, fetch=False)
        out = (np.zeros(count), np.zeros((count, 6 * (k_max + 1))), np.full(count, k_max, dtype=np.int32))
    b = time.perf_counter() - t0[0]
    log.append((params.target_base, round(1e3 * a, 2), round(1e3 * b, 2)))
    return out
_ffi.Context.decompose_range = timed
marks = {}
orig_rbw = TemplateOptimizer._run_batch_windows
def rbw(self, *a, **k):
    marks["enter"] = round(1e3 * (time.perf_counter() - t0[0]), 2)
    out = orig_rbw(self, *a, **k)
    marks["exit"] = round(1e3 * (time.perf_counter() - t0[0]), 2)
    return out
TemplateOptimizer._run_batch_windows = rbw
orig_fill = DeviceHaarBatch.fill
fills = []
def fill(self, ctx, first, count):
    a = time.perf_counter() - t0[0]
    out = orig_fill(self, ctx, first, count)
    fills.append((first, round(1e3 * a, 2), round(1e3 * (time.perf_counter() - t0[0]), 2)))
    return out
DeviceHaarBatch.fill = fill
for fetch in (True,):
    mode["fetch"] = fetch
    for r in range(25):
        log.clear(); fills.clear()
        opt = TemplateOptimizer(basis, BasicCost(), training_restarts=32, seed=20261003, override_fail=True, windows_in_flight=FL[r % len(FL)])
        t0[0] = time.perf_counter()
        loss, _, data = opt.approximate_from_distribution(DeviceHaarBatch(seed=20260000 + 9_500_000 + r, n_samples=N))
        dt = time.perf_counter() - t0[0]
        if KEEP:
            kept.append((loss, data, opt))
        time.sleep(0.01)
        if r:
            print(f"helpers {FL[r % len(FL)]} keep {KEEP}: total {1e3 * dt:.2f} ms; fills done at {max(f[2] for f in fills):.2f}; last window ends {max(l[2] for l in log):.2f}", flush=True)
