"""Dev tool (GPU): a LONE small batch through the one-wavefront-per-target span loop against the per-span launches (SLAM_FLAG_STAGED):
wall time per call (median of several), kernel time, evaluations.  usage: tools/r4_wave_probe.py [gate] [R]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from slam_decomposition_amd import _ffi

gname = sys.argv[1] if len(sys.argv) > 1 else "cx"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ctx = _ffi.Context(0)
ctx.set_gates(bench.gate_table(gname))
ctx.sample_haar(7, 8192)
seqs = [[0], [0, 0], [0, 0, 0]]
for N in (1, 16, 256, 512, 1024, 2048, 4096):
    row = []
    for name, extra in (("wave", 0), ("staged", _ffi.FLAG_STAGED)):
        prm = _ffi.OptParams(restarts=R, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=3, flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED | extra)
        ts = []
        for rep in range(7):
            ctx.reset_stats()
            t0 = time.perf_counter()
            l, x, c = ctx.decompose_range(0, N, 1, 3, seqs, prm, 1e-10)
            ts.append(time.perf_counter() - t0)
        st = ctx.stats()
        fl = sum(st["evals"][k] * bench.f_eval(k) for k in (1, 2, 3))
        med = sorted(ts[1:])[3]
        row.append(f"{name}: {1e3 * med:.3f} ms wall, {st['kernel_ms']:.3f} ms kernels, {st['kernel_launches']} launches, frac {fl / med / 78.6e12:.3f}, solved {(l < 1e-8).mean():.3f}")
    print(f"N = {N:5d} x {R}: " + " | ".join(row))
