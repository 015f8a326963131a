import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import slam_oracle as o
from oracle.bfgs_port import minimize_port
from slam_decomposition_amd import _ffi
ctx = _ffi.Context(0)
g = o.cx_matrix(); k = 1
T = o.haar_batch(2, seed0=2024)
ctx.set_targets(T); ctx.set_gates(g[None])
x0 = np.stack([[o.x0_philox(3, t, r, k) for r in range(1)] for t in range(2)])
for maxiter in (0, 1, 2):
    out = ctx.minimize_stage([0]*k, _ffi.OptParams(restarts=1, maxiter=maxiter), x0=x0)
    f, x, it, st, nev = minimize_port(x0[0,0], [g]*k, T[0], maxiter=maxiter)
    print("maxiter", maxiter, "gpu loss", out["item_loss"][0], "port", f, "iters", out["item_iters"][0], it, "evals", out["item_evals"][0], nev)
    print("  dx gpu ", (out["best_x"][0]-x0[0,0])[:6])
    print("  dx port", (x-x0[0,0])[:6])
l, gr = o.loss_and_grad(x0[0,0], [g]*k, T[0])
print("g", gr[:6], "|g|", np.linalg.norm(gr))
