"""Dev tool (GPU): distribution of per-item evaluation counts / statuses of one stage."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slam_decomposition_amd import _ffi
from bench import gate_table, make_targets

gname = sys.argv[1] if len(sys.argv) > 1 else "cx"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
R = int(sys.argv[3]) if len(sys.argv) > 3 else 16
ctx = _ffi.Context(0)
table = gate_table(gname)
ctx.set_gates(table)
ctx.set_targets(make_targets(N, 20260000))
prm = _ffi.OptParams(restarts=R, seed=20261003, flags=_ffi.FLAG_EARLY_EXIT)
for k in (1, 2, 3):
    seq = [i % len(table) for i in range(k)]
    ctx.reset_stats()
    out = ctx.minimize_stage(seq, prm)
    st = ctx.stats()
    ev = out["item_evals"].ravel()
    ev = ev[ev > 0]
    q = np.percentile(ev, [50, 90, 99, 99.9, 99.99, 100])
    print(f"{gname} k={k}: items run {ev.size}, evals mean {ev.mean():.1f}, pct50/90/99/99.9/99.99/max {q.astype(int).tolist()}, "
          f"status counts {np.bincount(out['item_status'].ravel(), minlength=6).tolist()}, kernel {st['kernel_ms']:.2f} ms, "
          f"sum evals {ev.sum()}, wave rounds {st['wave_rounds'][k]}")
