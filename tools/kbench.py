"""Dev tool: per-span kernel time of one resident batch, best of several repeats (GPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from slam_decomposition_amd import _ffi
from bench import gate_table, make_targets, f_eval

gname = sys.argv[1] if len(sys.argv) > 1 else "sqiswap"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
R = int(sys.argv[3]) if len(sys.argv) > 3 else 32
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
ctx = _ffi.Context(0)
table = gate_table(gname)
ctx.set_gates(table)
ctx.set_targets(make_targets(N, 20260000))
seqs = [[i % len(table) for i in range(k)] for k in (1, 2, 3)]
prm = _ffi.OptParams(restarts=R, seed=20261003, flags=int(os.environ.get("KB_FLAGS", str(_ffi.FLAG_EARLY_EXIT))), items_per_quad=int(os.environ.get("KB_IPQ", "0")))
best = None
for rep in range(reps + 1):
    ctx.reset_stats()
    ctx.decompose(1, 3, seqs, prm, 1e-10, fetch=False)
    st = ctx.stats()
    if rep == 0:
        continue  # warm-up
    if best is None or st["kernel_ms"] < best["kernel_ms"]:
        best = st
tf = sum(best["evals"][k] * f_eval(k) for k in (1, 2, 3)) / (best["kernel_ms"] * 1e-3) / 1e12
print(f"{gname} N={N} R={R}: kernel ms per span {[round(best['kernel_ms_span'][k], 2) for k in (1, 2, 3)]} total {best['kernel_ms']:.2f} ms; "
      f"evals {[best['evals'][k] for k in (1, 2, 3)]}; {tf:.2f} TF/s = {100 * tf / 78.6:.1f}% ; "
      f"quad occupancy {[round(best['evals'][k] / 16 / max(best['wave_rounds'][k], 1), 3) for k in (1, 2, 3)]}; "
      f"wave rounds {[best['wave_rounds'][k] for k in (1, 2, 3)]}; "
      f"G evals/s per span {[round(best['evals'][k] / best['kernel_ms_span'][k] / 1e6, 3) if best['kernel_ms_span'][k] else 0 for k in (1, 2, 3)]}")
