"""Dev tool (GPU): wall time of the reference-shaped Python API on a big batch, next to the C-ABI span loop itself."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cProfile, pstats
import numpy as np
from slam_decomposition_amd.basis import CircuitTemplate
from slam_decomposition_amd.cost_function import BasicCost
from slam_decomposition_amd.gates import RiSwapGate
from slam_decomposition_amd.optimizer import TemplateOptimizer
from slam_decomposition_amd.sampler import DeviceHaarBatch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
basis = CircuitTemplate(base_gates=[RiSwapGate(0.5)], maximum_span_guess=3)
for rep in range(2):
    opt = TemplateOptimizer(basis, BasicCost(), training_restarts=32, seed=1)
    t0 = time.perf_counter()
    pr = cProfile.Profile(); pr.enable()
    loss, coords, data = opt.approximate_from_distribution(DeviceHaarBatch(seed=7, n_samples=N))
    pr.disable()
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {N} targets in {dt:.3f} s = {N / dt:.3g} decompositions/s through TemplateOptimizer; kernel ms {opt.last_stats['kernel_ms']:.1f}")
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
