"""Dev tool (GPU): 300 calls of mixed sizes / bases / paths through one context and through TemplateOptimizer; device memory in use
before and after (helper contexts, staging buffers and streams are created once and reused: no growth after the first rounds)."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from slam_decomposition_amd import _ffi
from slam_decomposition_amd import gates as G

hip = ctypes.CDLL("libamdhip64.so")


def used_mb():
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
    return (total.value - free.value) / 2**20


rng = np.random.default_rng(1)
ctx = _ffi.Context(0)
ctx.sample_haar(5, 20000)
tables = [G.CXGate().to_matrix()[None], G.RiSwapGate(0.5).to_matrix()[None], np.stack([G.RiSwapGate(1.0).to_matrix(), G.BerkeleyGate().to_matrix()])]
seqs_of = lambda t: [[i % len(t) for i in range(k)] for k in (1, 2, 3)]
marks = []
t0 = time.time()
for it in range(300):
    t = tables[it % 3]
    ctx.set_gates(t)
    N = int(rng.choice([1, 7, 300, 600, 1024, 3000, 9000, 20000]))
    R = int(rng.choice([3, 16, 32]))
    if N * R > 400000:
        R = 16
    prm = _ffi.OptParams(restarts=R, maxiter=300, seed=it, flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED | int(rng.choice([0, _ffi.FLAG_OVERLAP, _ffi.FLAG_STAGED])))
    l, x, c = ctx.decompose_range(0, N, 1, 3, seqs_of(t), prm, 1e-10)
    assert np.isfinite(l).all()
    if it in (30, 100, 299):
        marks.append(used_mb())
print(f"300 calls in {time.time() - t0:.1f} s; device memory in use after 31 / 101 / 300 calls: {[round(m) for m in marks]} MB")
assert marks[2] - marks[1] < 64, marks
