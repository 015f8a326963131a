"""Dev tool (GPU): what a multi-queue launch costs against a single queue of the same total size.  One conversion-gain gate
(configs[4] sweep basis 64, class XRI), 16 restarts: (a) ONE context with 16 x 4096 targets (one queue of 1 M items per span);
(b) slam_decompose_multi over 16 contexts of 4096 targets each, all with that gate (16 queues behind one launch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from slam_decomposition_amd import _ffi

B, N, R = 16, 4096, 16
g = bench.sweep_gate(int(sys.argv[1]) if len(sys.argv) > 1 else 64)
prm = _ffi.OptParams(restarts=R, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=5, flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED)
seqs = [[0], [0, 0], [0, 0, 0]]


def show(tag, sts):
    ev = [sum(s["evals"][k] for s in sts) for k in range(4)]
    ms = [sum(s["kernel_ms_span"][k] for s in sts) for k in range(4)]
    print(tag, {k: (round(ms[k], 3), round(ev[k] * bench.f_eval(k) / (ms[k] * 1e-3) / 78.6e12, 3)) for k in (1, 2, 3)}, "total ms", round(sum(ms), 2), "evals", ev[1:])


one = _ffi.Context(0)
one.set_gates(g[None])
one.sample_haar(99, B * N)
for rep in range(3):
    one.reset_stats()
    one.decompose_range(0, B * N, 1, 3, seqs, prm, 1e-10, fetch=False)
    if rep:
        show("single queue 65536 x 16:", [one.stats()])
ctxs = [_ffi.Context(0) for _ in range(B)]
for i, c in enumerate(ctxs):
    c.set_gates(g[None])
    c.sample_haar(99, N, first_index=i * N)  # the same 65536 targets, 4096 per context
for rep in range(3):
    for c in ctxs:
        c.reset_stats()
    _ffi.decompose_multi(ctxs, 0, N, 1, 3, seqs, prm, 1e-10)
    if rep:
        show("16 queues, one launch    :", [c.stats() for c in ctxs])
