"""Dev tool: distribution of per-item iteration counts / statuses per span stage (GPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slam_decomposition_amd import _ffi
from bench import gate_table, make_targets

gname = sys.argv[1] if len(sys.argv) > 1 else "cx"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
R = int(sys.argv[3]) if len(sys.argv) > 3 else 16
ctx = _ffi.Context(0)
table = gate_table(gname)
ctx.set_gates(table)
ctx.set_targets(make_targets(N, 20260000))
for k in (1, 2, 3):
    prm = _ffi.OptParams(restarts=R, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=20261003, flags=0)
    out = ctx.minimize_stage([i % len(table) for i in range(k)], prm)
    it = out["item_iters"].ravel(); ev = out["item_evals"].ravel(); st = out["item_status"].ravel()
    wave_max = out["item_evals"].reshape(-1, 16 // R if R < 16 else 1, R)[:, :, :].max(axis=(1, 2)) if R >= 16 else None
    print(f"k={k} iters pct [50,90,99,99.9,max] = {np.percentile(it,[50,90,99,99.9,100])}  evals/iter {ev.sum()/max(1,it.sum()):.3f} status {np.bincount(st,minlength=6)}")
    print(f"     evals mean {ev.mean():.1f}; per-target max evals: mean {out['item_evals'].max(axis=1).mean():.1f} max {out['item_evals'].max()}; solved frac {(out['best_loss']<1e-10).mean():.3f}")
    big = np.argsort(it)[-5:]
    print("     slowest items: iters", it[big], "loss", out["item_loss"].ravel()[big], "status", st[big])
    print("     stage ms", ctx.stats()["total_ms"])
