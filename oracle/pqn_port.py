"""CPU port of the projected quasi-Newton loop of ``minimize_v2_kernel`` (csrc/slam_v2.hpp): box bounds, and one linear
cost constraint through an augmented Lagrangian around the same loop.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  The reference hands a template with bounds to SciPy's L-BFGS-B and one
with a cost constraint to SLSQP (src/slam/optimizer.py:255-265, basisv2.py:174-200); the HIP path runs ONE loop for all of them:
BFGS metric, trial points projected onto the box, gradient restricted to the variables that can move.  A cost constraint
``w.x <= cmax`` enters as the augmented Lagrangian  L(x) = loss(x) + rho / 2 max(0, w.x - cmax + mu / rho)^2 : when the loop has
come to rest on L the multiplier estimate moves, mu <- max(0, mu + rho c), and the loop goes on from the same point with the
same metric, until c <= tol and mu c = 0 within tol.  This file restates that iteration in NumPy (float64 throughout; the kernel
keeps the inverse Hessian in float32) so that its fixed points can be compared with SciPy's on a CPU.
"""
from __future__ import annotations

import numpy as np

from .bfgs_port import ARMIJO_C1, CURV_EPS, GROW_FACTOR, GROW_MAX, MAX_BACKTRACK, RESTART_PERIOD, STALL_DF, STALL_GNORM, STEP_MAX, WOLFE_C2

MAX_MULTIPLIER_UPDATES = 24  # multiplier updates + metric restarts after failed line searches, per item
RHO_TIMES_W2MAX = 30.0   # slam_v2_set_constraint: rho = 30 / max w^2
CONS_TOL = 1e-8          # feasibility / complementarity tolerance, relative to 1 + |cmax|


def _blocked(x, d, lo, hi):
    """Components of a direction d that leave the box at x."""
    return ((x <= lo) & (d < 0.0)) | ((x >= hi) & (d > 0.0))


def minimize_port(fun, x0, lo, hi, w=None, cmax=0.0, maxiter=2500, gtol=1e-9, stop_loss=1e-13, gtol_far=1e-5, far_loss=1e-6):
    """fun(x) -> (loss, gradient).  Returns (loss, x, iters, status, n_evals, multiplier); loss is the plain loss, +inf for an
    item that ran out of multiplier updates with the constraint still violated."""
    n = len(x0)
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    cons = w is not None
    bounded = cons or bool(np.any(np.isfinite(lo)) or np.any(np.isfinite(hi)))  # (the kernel's wave-uniform `bounded`)
    if cons:
        tol = CONS_TOL * (1.0 + abs(cmax))
        cm = cmax - tol  # results are feasible: the loop ends with c <= tol against a right-hand side lowered by tol
        rho = RHO_TIMES_W2MAX / np.max(w * w)
    mu, nouter = 0.0, 0

    def lagrangian(xx):
        f0, g = fun(xx)
        g = np.asarray(g, dtype=np.float64)
        if not cons:
            return f0, g, f0, 0.0
        c = w @ xx - cm
        t = rho * max(0.0, c + mu / rho)
        return f0 + 0.5 * t * t / rho, g + t * w, f0, c

    x = np.minimum(np.maximum(np.asarray(x0, dtype=np.float64), lo), hi)
    H = np.eye(n)
    hs1, scaled = 0.0, False
    g = np.zeros(n)
    p = np.zeros(n)
    hg = np.zeros(n)
    f = f0 = cc = 0.0
    alpha, gp, grow = 0.0, 0.0, 1.0
    iters = nback = nstall = nev = 0
    fresh = True
    while True:
        xt = np.minimum(np.maximum(x + alpha * p, lo), hi)
        s = xt - x
        gs = g @ s
        ft, gt, ft0, ct = lagrangian(xt)
        nev += 1
        finite = np.isfinite(ft)
        gt = np.where(_blocked(xt, -gt, lo, hi), 0.0, gt) if finite else np.zeros(n)
        acc = finite if fresh else (finite and ft <= f + ARMIJO_C1 * gs)
        step = acc and not fresh
        done, status = False, 1
        if acc:
            sv = s if step else np.zeros(n)
            qv = H @ gt
            y = gt - g
            sy, yy, ss, sg = sv @ y, y @ y, sv @ sv, sv @ gt
            too_short = sy < (1.0 - WOLFE_C2) * (-gs)
            curv = step and (not too_short) and sy > 0.0 and sy * sy > (CURV_EPS * CURV_EPS) * (ss * yy)
            first = curv and not scaled
            scaled = scaled or curv
            fac = sy / yy if first else 1.0
            if first:
                hs1 = fac - 1.0
            qv = qv + hs1 * gt
            u = qv - fac * hg
            rho_b = 1.0 / sy if curv else 0.0
            cf = rho_b * (1.0 + rho_b * (y @ u))
            wv = cf * sv - rho_b * u
            vv = -rho_b * u
            H = H + np.outer(sv, wv) + np.outer(vv, sv)
            nstall = nstall + 1 if (step and (f - ft) <= STALL_DF) else 0
            f, f0, cc = ft, ft0, ct
            if step:
                iters += 1
            nback = 0
            grow = min(grow * GROW_FACTOR, GROW_MAX) if (step and too_short) else 1.0
            x, g = xt, gt
            hg = qv + sv * (wv @ gt) + vv * sg
            p = np.where(_blocked(x, -hg, lo, hi), 0.0, -hg)
        elif fresh:
            return ft, x, 0, 3, nev, mu
        else:
            denom = 2.0 * (ft - f - gs)
            anew = (-gs * alpha / denom) if (finite and denom > 0.0 and np.isfinite(denom)) else 0.5 * alpha
            alpha = min(max(anew, 0.1 * alpha), 0.5 * alpha)
            grow = 1.0
            nback += 1
        gnorm = np.abs(np.where(_blocked(x, -g, lo, hi), 0.0, g)).max()
        gp, pp = g @ p, p @ p
        if acc:
            alpha = min(grow, STEP_MAX / np.sqrt(pp)) if pp > 1e-300 else grow
            if f < stop_loss or gnorm < gtol or ((not cons) and gnorm < gtol_far and f > far_loss):
                done, status = True, 0
            elif nstall >= 2:
                done, status = True, 4
            elif iters >= maxiter:
                done, status = True, 1
        elif nback > MAX_BACKTRACK:
            done, status = True, (4 if gnorm < STALL_GNORM else 2)
        fresh = False
        periodic = step and not done and iters % RESTART_PERIOD == 0
        if not done and (not gp < 0.0 or periodic):
            H = np.eye(n)
            hs1 = 0.0
            if periodic:
                scaled = False
            hg = g.copy()
            p = np.where(_blocked(x, -g, lo, hi), 0.0, -g)
            gp = g @ p
            if periodic:
                alpha = min(grow, STEP_MAX / np.sqrt(-gp)) if gp < -1e-300 else grow
            if not gp < 0.0:
                done, status = True, 0  # nothing left to move along: a KKT point of the box-constrained problem
        # A failed line search is not a point of rest.  With bounds the projected step of a quasi-Newton direction need not be a
        # descent direction (a variable about to reach its bound is clamped mid-step, and the metric couples it to the others);
        # the projected step of the steepest-descent direction always is: the metric starts over from the identity and the loop
        # goes on (the variable lands ON its bound and is left out from then on).  No multiplier update from such a point.
        if bounded and done and status == 2 and nouter < MAX_MULTIPLIER_UPDATES:
            H = np.eye(n)
            hs1, scaled = 0.0, False
            nouter += 1
            done, fresh, alpha, nstall, nback = False, True, 0.0, 0, 0
        elif cons:
            rest = done and status in (0, 4)
            mu_new = max(0.0, mu + rho * cc)
            kkt = cc <= tol and (mu_new == 0.0 or abs(cc) <= tol)
            if rest and not kkt and nouter < MAX_MULTIPLIER_UPDATES:
                mu = mu_new
                nouter += 1
                done, fresh, alpha, nstall, nback = False, True, 0.0, 0, 0
        if done:
            if cons:
                return (f0 if cc <= tol else np.inf), x, iters, status, nev, mu
            return f, x, iters, status, nev, mu
