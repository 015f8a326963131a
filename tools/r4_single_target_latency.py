"""Dev tool (GPU): wall time of the reference's atomic call, TemplateOptimizer.approximate_target_U(U) (optimizer.py:65-119), for ONE Haar
target (default 5 restarts, spans 1..3): median of 20 calls after a warm-up, log lines off."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from slam_decomposition_amd.basis import CircuitTemplate
from slam_decomposition_amd.cost_function import BasicCost
from slam_decomposition_amd.gates import RiSwapGate
from slam_decomposition_amd.optimizer import TemplateOptimizer
from slam_decomposition_amd.sampler import random_unitary

basis = CircuitTemplate(base_gates=[RiSwapGate(0.5)], maximum_span_guess=3)
for R in (5, 16):
    ts = []
    for i in range(25):
        U = random_unitary(4, seed=100 + i)
        opt = TemplateOptimizer(basis, BasicCost(), training_restarts=R, seed=i, override_fail=True)
        t0 = time.perf_counter()
        d = opt.approximate_target_U(U)
        ts.append(time.perf_counter() - t0)
    ts = sorted(ts[5:])
    print(f"approximate_target_U, {R} restarts: median {1e3 * ts[len(ts) // 2]:.3f} ms, min {1e3 * ts[0]:.3f} ms (loss {d.loss_result:.2e}, cycles {d.cycles}, kernel ms {opt.last_stats['kernel_ms']:.3f})")
