"""Dev tool: pure cost of one optimizer round (evaluation + quasi-Newton update) per span, with the
profiling flag 0x100 (every step accepted, nothing converges, maxiter rounds per item)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slam_decomposition_amd import _ffi
from bench import gate_table, make_targets, f_eval

gname = sys.argv[1] if len(sys.argv) > 1 else "sqiswap"
waves = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 400
ctx = _ffi.Context(0)
table = gate_table(gname)
ctx.set_gates(table)
N = waves  # 16 restarts per target -> one wave per target
ctx.set_targets(make_targets(N, 20260000))
for k in (1, 2, 3):
    prm = _ffi.OptParams(restarts=16, maxiter=rounds, seed=1, flags=0x100)
    best = 1e9
    for rep in range(3):
        ctx.reset_stats()
        ctx.minimize_stage([i % len(table) for i in range(k)], prm, want_items=False)
        best = min(best, ctx.stats()["kernel_ms"])
    per_round_us = best * 1e3 / (rounds + 1)
    ev = N * 16 * (rounds + 1)
    tf = ev * f_eval(k) / (best * 1e-3) / 1e12
    print(f"k={k}: {best:.2f} ms for {rounds + 1} rounds x {waves} waves -> {per_round_us:.3f} us/round, {tf:.2f} TF/s = {100 * tf / 78.6:.1f}% of fp64 peak")
