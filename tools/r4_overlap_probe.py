"""Dev tool (GPU): would running the spans of ONE call side by side pay?  Three contexts with the same targets, one span each on
its own stream / host thread, against the span loop of one context (CNOT: no Haar target is solved before k = 3, so nothing is wasted).
usage: tools/r4_overlap_probe.py [gate] [N] [R]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from slam_decomposition_amd import _ffi

gname = sys.argv[1] if len(sys.argv) > 1 else "cx"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20480
R = int(sys.argv[3]) if len(sys.argv) > 3 else 16
ctxs = [_ffi.Context(0) for _ in range(4)]
for c in ctxs:
    c.set_gates(bench.gate_table(gname))
    c.sample_haar(7, N)
seqs = [[0], [0, 0], [0, 0, 0]]
prm = _ffi.OptParams(restarts=R, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=3, flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED | _ffi.FLAG_STAGED)


def loop():
    t0 = time.perf_counter()
    res = ctxs[3].decompose_range(0, N, 1, 3, seqs, prm, 1e-10)
    return time.perf_counter() - t0, res


def side_by_side(ks=(1, 2, 3)):
    out = {}

    def work(k):
        out[k] = ctxs[k - 1].decompose_range(0, N, k, k, [seqs[k - 1]], prm, 1e-10, fetch=False)

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in ks]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return time.perf_counter() - t0


for rep in range(2):
    loop(); side_by_side()
def native(flags_extra):
    p2 = _ffi.OptParams(restarts=R, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=3, flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED | flags_extra)
    t0 = time.perf_counter()
    ctxs[3].decompose_range(0, N, 1, 3, seqs, p2, 1e-10)
    return time.perf_counter() - t0


native(_ffi.FLAG_OVERLAP); native(0)
tn = sorted(native(_ffi.FLAG_OVERLAP) for _ in range(7))[3]
td = sorted(native(0) for _ in range(7))[3]
print(f"{gname} {N} x {R}: the library's overlapped spans (SLAM_FLAG_OVERLAP) {1e3 * tn:.3f} ms | its default path {1e3 * td:.3f} ms")
tl = sorted(loop()[0] for _ in range(7))[3]
ts = sorted(side_by_side() for _ in range(7))[3]
t12 = sorted(side_by_side((1, 2)) for _ in range(7))[3]
singles = [sorted(side_by_side((k,)) for _ in range(5))[2] for k in (1, 2, 3)]
print(f"{gname} {N} x {R}: span loop {1e3 * tl:.3f} ms | three spans side by side {1e3 * ts:.3f} ms | k = 1, 2 side by side {1e3 * t12:.3f} ms | "
      f"each span alone (all targets) {[round(1e3 * t, 3) for t in singles]} ms")
