"""Dev tool (GPU): steady-state workload for PMC runs -- uniform items (fixed iteration count), one span."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_decomposition_amd import _ffi
from bench import gate_table, make_targets

gname, k, ipq = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
N = 2048 * 16 * 8
ctx = _ffi.Context(0)
table = gate_table(gname)
ctx.set_gates(table)
ctx.set_targets(make_targets(N, 20260000))
prm = _ffi.OptParams(restarts=1, maxiter=20, gtol=0.0, gtol_far=0.0, stop_loss=-1.0, seed=7, flags=0, items_per_quad=ipq)
for rep in range(3):
    ctx.reset_stats()
    ctx.minimize_stage([i % len(table) for i in range(k)], prm, want_items=False)
st = ctx.stats()
print(gname, k, ipq, "kernel ms", st["kernel_ms"], "evals", st["evals"][k], "wave_rounds", st["wave_rounds"][k])
