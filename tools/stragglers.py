"""Dev tool (GPU): list the longest-running (target, restart) items of one stage, with x0 seeds and targets,
so that they can be replayed with the CPU port (fp32 vs fp64 inverse Hessian)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slam_decomposition_amd import _ffi
from bench import gate_table, make_targets

gname, k, N, R = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ctx = _ffi.Context(0)
table = gate_table(gname)
ctx.set_gates(table)
T = make_targets(N, 20260000)
ctx.set_targets(T)
prm = _ffi.OptParams(restarts=R, seed=20261003, flags=0)
out = ctx.minimize_stage([i % len(table) for i in range(k)], prm)
ev = out["item_evals"]
idx = np.argsort(ev.ravel())[::-1][:24]
t, r = np.unravel_index(idx, ev.shape)
np.savez("gpurun_out/stragglers_%s_%d.npz" % (gname, k), targets=T[t], t=t, r=r, evals=ev[t, r], iters=out["item_iters"][t, r],
         status=out["item_status"][t, r], loss=out["item_loss"][t, r], seed=20261003, k=k, gate=table)
print("evals", ev[t, r].tolist()); print("status", out["item_status"][t, r].tolist()); print("loss", out["item_loss"][t, r].tolist())
