"""Dev tool (GPU): batches in the wave kernels' range through the overlapped spans instead (run with SLAM_WAVE_LOOP=0) -- wall time per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from slam_decomposition_amd import _ffi
ctx = _ffi.Context(0)
ctx.sample_haar(7, 4096)
seqs = [[0], [0, 0], [0, 0, 0]]
for g in ("cx", "sqiswap"):
    ctx.set_gates(bench.gate_table(g))
    for N, R in ((256, 32), (512, 16), (512, 32), (768, 16), (1024, 16), (1024, 32), (1024, 64)):
        prm = _ffi.OptParams(restarts=R, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=3, flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED)
        ts = []
        for rep in range(8):
            ctx.reset_stats()
            t0 = time.perf_counter(); ctx.decompose_range(0, N, 1, 3, seqs, prm, 1e-10); ts.append(time.perf_counter() - t0)
        print(f"{g} {N} x {R}: {1e3 * sorted(ts[1:])[3]:.3f} ms, {ctx.stats()['kernel_launches']} launches", flush=True)
