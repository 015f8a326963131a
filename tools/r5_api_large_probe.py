"""Dev tool (GPU): approximate_from_distribution on a 327 680-target sampler over window size x helpers (windows in flight) x staggered first windows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slam_decomposition_amd.basis import CircuitTemplate
from slam_decomposition_amd.cost_function import BasicCost
from slam_decomposition_amd.gates import RiSwapGate
from slam_decomposition_amd.optimizer import TemplateOptimizer
from slam_decomposition_amd.sampler import DeviceHaarBatch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 327680
basis = CircuitTemplate(base_gates=[RiSwapGate(0.5)], maximum_span_guess=3)
cases = [(65536, 4, False), (65536, 5, False), (65536, 6, False), (65536, 4, True), (32768, 5, False), (1 << 30, 1, False)]
for W, F, S in cases:
    ts = []
    for r in range(10):
        opt = TemplateOptimizer(basis, BasicCost(), training_restarts=32, seed=20261003, override_fail=True, windows_in_flight=max(F, 2) if W < (1 << 30) else 1)
        opt.WINDOW_TARGETS, opt.window_stagger = W, S
        t0 = time.perf_counter()
        loss, _, data = opt.approximate_from_distribution(DeviceHaarBatch(seed=20260000 + 9_500_000 + r, n_samples=N))
        dt = time.perf_counter() - t0
        if r:
            ts.append(dt)
    ts.sort()
    print(f"window {W} in flight {F} stagger {S}: {1e3 * ts[len(ts) // 2]:.2f} ms  ({N / ts[len(ts) // 2]:.4g} /s)  all {[round(1e3 * t, 1) for t in ts]}  windows {len(opt.last_stats_per_device)}", flush=True)
