"""CPU port of the HIP kernel's per-(target, seed) optimizer loop.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  The reference runs
``scipy.optimize.minimize(method="BFGS")`` with finite-difference gradients
(src/slam/optimizer.py:270-278).  The HIP path keeps the quasi-Newton (BFGS,
inverse-Hessian form) iteration but uses the analytic gradient and a
safeguarded Armijo backtracking line search that is cheap to run in lock-step
over a wavefront.  This file restates that exact iteration in NumPy so tests
can compare the kernel step by step on a CPU, and so its convergence can be
compared with SciPy's BFGS on the same targets and seeds.

Like the kernel, the inverse-Hessian approximation H (a preconditioner) is
stored and applied in float32; loss, gradient, parameters, steps and all
scalars are float64.

Status codes mirror the C-ABI (include/slam_hip.h):
0 converged (loss < stop_loss, or |g|_inf < gtol, or |g|_inf < gtol_far at a
loss > far_loss), 1 maxiter, 2 line-search failure, 3 non-finite, 4 stalled at
the fp64 noise floor (no representable decrease of the loss).
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
WOLFE_C2 = 0.7     # an accepted step whose slope along p fell by less than (1 - c2) was too short: no metric update from it ...
GROW_FACTOR = 8.0  # ... the next first trial step is this much longer (compounding while it keeps happening)
GROW_MAX = 1048576.0
RESTART_PERIOD = 128  # every so many accepted iterations the quasi-Newton metric starts over from the identity (see below)


def minimize_port(
    x0, gate_seq, target, maxiter=2500, gtol=1e-9, stop_loss=1e-13, gtol_far=1e-5, far_loss=1e-6, trace=None,
    h_dtype=np.float32,
):
    """Returns (loss, x, iters, status, n_evals)."""
    n = len(x0)
    x = np.array(x0, dtype=np.float64)
    f, g = o.loss_and_grad(x, gate_seq, target)
    nev = 1
    if not np.isfinite(f):
        return f, x, 0, 3, nev
    H = np.eye(n, dtype=h_dtype)
    hs1 = 0.0  # the effective inverse Hessian is H + hs1 I: the one-off scaling of the initial one is a scalar (as in the kernel)
    p = -(H @ g.astype(h_dtype)).astype(np.float64)  # the kernel's first direction is -H g in float32 too
    gnorm = np.abs(g).max()

    def converged():
        return f < stop_loss or gnorm < gtol or (gnorm < gtol_far and f > far_loss)

    if converged():
        return f, x, 0, 0, nev
    if maxiter <= 0:
        return f, x, 0, 1, nev
    alpha = min(1.0, STEP_MAX / max(np.sqrt(p @ p), 1e-300))
    grow = 1.0
    it = 0
    nback = 0
    nstall = 0
    scaled = False
    status = 1
    while it < maxiter:
        gp = g @ p
        if not (gp < 0):
            # not a descent direction (H lost positive-definiteness numerically): reset
            H = np.eye(n, dtype=h_dtype)
            hs1 = 0.0
            p = -g
            gp = g @ p
        s = alpha * p
        xt = x + s
        ft, gt = o.loss_and_grad(xt, gate_seq, target)
        nev += 1
        if not np.isfinite(ft):
            ft = np.inf
        if ft <= f + ARMIJO_C1 * alpha * gp:
            y = gt - g
            # s = alpha p: the products with s come from p.g' and the direction's own p.g, p.p (the kernel's formulas)
            pgt = p @ gt
            sy = alpha * (pgt - gp)
            ss = (alpha * alpha) * (p @ p)
            sg = alpha * pgt
            # weak-Wolfe curvature condition violated (in particular: negative curvature along p, where the
            # update below is skipped and H never learns to take longer steps): lengthen the next trial step
            short = sy < (1.0 - WOLFE_C2) * alpha * (-gp)
            q = (H @ gt.astype(h_dtype)).astype(np.float64)
            if (not short) and sy > CURV_EPS * np.sqrt(ss * (y @ y)):
                rho = 1.0 / sy
                fac = 1.0
                if not scaled:
                    # scale the initial inverse Hessian (the identity) before its first update (Nocedal & Wright
                    # eq. 6.20): H_eff = H + hs1 I with hs1 = fac - 1
                    fac = sy / (y @ y)
                    hs1 = fac - 1.0
                    scaled = True
                q = q + hs1 * gt
                u = q + fac * p  # H_eff y = H_eff g' - H_eff g,  p = -H_eff g before this round's scaling
                c = rho * (1.0 + rho * (y @ u))
                w = c * s - rho * u
                v = -rho * u
                H = H + np.outer(s.astype(h_dtype), w.astype(h_dtype)) + np.outer(v.astype(h_dtype), s.astype(h_dtype))
                pn = -(q + s * (w @ gt) + v * sg)
            else:
                pn = -(q + hs1 * gt)
            nstall = nstall + 1 if (f - ft) <= STALL_DF else 0
            x, f, g = xt, ft, gt
            it += 1
            nback = 0
            if trace is not None:
                trace.append(f)
            gnorm = np.abs(g).max()
            if converged():
                status = 0
                break
            if nstall >= 2:
                status = 4
                break
            p = pn
            grow = min(grow * GROW_FACTOR, GROW_MAX) if short else 1.0
            alpha = min(grow, STEP_MAX / max(np.sqrt(p @ p), 1e-300))
            if it % RESTART_PERIOD == 0:
                # Periodic restart.  One item in 1e3..1e5 ends up with a metric that has stopped learning (steps nearly
                # orthogonal to the gradient on a plateau: 400..1300 iterations where SciPy's BFGS needs 50..170 from the
                # same start); restarted from the identity it is through in ~40 more.  Items this long are past the 99th
                # percentile of the iteration counts, so the mean does not notice -- the stage's critical path does.
                H = np.eye(n, dtype=h_dtype)
                hs1 = 0.0
                scaled = False
                p = -g
                alpha = min(grow, STEP_MAX / max(np.sqrt(p @ p), 1e-300))
        else:
            # safeguarded quadratic interpolation backtrack
            denom = 2.0 * (ft - f - gp * alpha)
            a_new = -gp * alpha * alpha / denom if denom > 0 and np.isfinite(denom) else 0.5 * alpha
            alpha = min(max(a_new, 0.1 * alpha), 0.5 * alpha)
            grow = 1.0
            nback += 1
            if nback > MAX_BACKTRACK:
                status = 4 if gnorm < STALL_GNORM else 2
                break
    return f, x, it, status, nev
