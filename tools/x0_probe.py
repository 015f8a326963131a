"""Dev tool (GPU): what the in-kernel Philox start points cost -- one stage with explicit x0 (read from HBM) against the same
stage with in-kernel seeds.  usage: tools/x0_probe.py [gate] [N] [R]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_decomposition_amd import _ffi
from bench import gate_table, make_targets

gname = sys.argv[1] if len(sys.argv) > 1 else "sqiswap"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
R = int(sys.argv[3]) if len(sys.argv) > 3 else 32
ctx = _ffi.Context(0)
table = gate_table(gname)
ctx.set_gates(table)
ctx.set_targets(make_targets(N, 20260000))
rng = np.random.default_rng(1)
for k in (1, 2):
    seq = [i % len(table) for i in range(k)]
    n = 6 * (k + 1)
    x0 = rng.uniform(0, 2 * np.pi, size=(N, R, n))
    for label, xx in (("philox", None), ("x0 from HBM", x0), ("philox", None), ("x0 from HBM", x0)):
        prm = _ffi.OptParams(restarts=R, seed=7, flags=0)
        best = None
        for rep in range(3):
            ctx.reset_stats()
            ctx.minimize_stage(seq, prm, x0=xx, want_items=False)
            st = ctx.stats()
            if rep and (best is None or st["kernel_ms"] < best["kernel_ms"]):
                best = st
        ev, wr, ms = best["evals"][k], best["wave_rounds"][k], best["kernel_ms"]
        print(f"{gname} k={k} {label:12s}: {ms:7.2f} ms, evals {ev}, occupancy {ev/16/wr:.3f}, rounds {wr}, {ev/ms/1e6:.3f} G evals/s", flush=True)
