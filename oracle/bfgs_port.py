"""CPU port of the HIP kernel's per-(target, seed) optimizer loop.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  The reference runs
``scipy.optimize.minimize(method="BFGS")`` with finite-difference gradients
(src/slam/optimizer.py:270-278).  The HIP path keeps the quasi-Newton (BFGS,
inverse-Hessian form) iteration but uses the analytic gradient and a
safeguarded Armijo backtracking line search that is cheap to run in lock-step
over a wavefront.  This file restates that exact iteration in NumPy so tests
can compare the kernel step by step on a CPU, and so its convergence can be
compared with SciPy's BFGS on the same targets and seeds.

Status codes mirror the C-ABI (include/slam_hip.h):
0 converged (loss < stop_loss or |g|_inf < gtol), 1 maxiter, 2 line-search
failure, 3 non-finite, 4 stalled at the fp64 noise floor (no representable
decrease of the loss; happens at non-zero local minima where |g|_inf ~ 1e-9
while loss differences fall under one ulp).
"""
from __future__ import annotations

import numpy as np

from . import slam_oracle as o

ARMIJO_C1 = 1e-4
MAX_BACKTRACK = 20
STEP_MAX = 2.0  # cap on |alpha p|_2 of the first trial step (parameters are angles)
CURV_EPS = 1e-10
STALL_DF = 1e-15
STALL_GNORM = 1e-5


def minimize_port(x0, gate_seq, target, maxiter=2500, gtol=1e-9, stop_loss=1e-13, trace=None):
    """Returns (loss, x, iters, status, n_evals)."""
    n = len(x0)
    x = np.array(x0, dtype=np.float64)
    f, g = o.loss_and_grad(x, gate_seq, target)
    nev = 1
    H = np.eye(n)
    p = -g
    gnorm = np.abs(g).max()
    alpha = min(1.0, 1.0 / max(np.sqrt(g @ g), 1e-300))
    it = 0
    nback = 0
    nstall = 0
    scaled = False
    status = 1
    if not np.isfinite(f):
        return f, x, 0, 3, nev
    if f < stop_loss or gnorm < gtol:
        return f, x, 0, 0, nev
    while it < maxiter:
        gp = g @ p
        if not (gp < 0):
            # not a descent direction (H lost positive-definiteness numerically): reset
            H = np.eye(n)
            p = -g
            gp = g @ p
        xt = x + alpha * p
        ft, gt = o.loss_and_grad(xt, gate_seq, target)
        nev += 1
        if not np.isfinite(ft):
            ft = np.inf
        if ft <= f + ARMIJO_C1 * alpha * gp:
            s = xt - x
            y = gt - g
            sy = s @ y
            if sy > CURV_EPS * np.sqrt((s @ s) * (y @ y)):
                rho = 1.0 / sy
                if not scaled:
                    # scale the initial inverse Hessian before its first update
                    # (Nocedal & Wright eq. 6.20)
                    H = H * (sy / (y @ y))
                    scaled = True
                u = H @ y
                c = rho * (1.0 + rho * (y @ u))
                H = H + c * np.outer(s, s) - rho * (np.outer(s, u) + np.outer(u, s))
            nstall = nstall + 1 if (f - ft) <= STALL_DF else 0
            x, f, g = xt, ft, gt
            it += 1
            nback = 0
            if trace is not None:
                trace.append(f)
            gnorm = np.abs(g).max()
            if f < stop_loss or gnorm < gtol:
                status = 0
                break
            if nstall >= 2:
                status = 4
                break
            p = -(H @ g)
            alpha = min(1.0, STEP_MAX / max(np.sqrt(p @ p), 1e-300))
        else:
            # safeguarded quadratic interpolation backtrack
            denom = 2.0 * (ft - f - gp * alpha)
            a_new = -gp * alpha * alpha / denom if denom > 0 and np.isfinite(denom) else 0.5 * alpha
            alpha = min(max(a_new, 0.1 * alpha), 0.5 * alpha)
            nback += 1
            if nback > MAX_BACKTRACK:
                status = 4 if gnorm < STALL_GNORM else 2
                break
    return f, x, it, status, nev
