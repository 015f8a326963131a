"""``TemplateOptimizer(override_method=...)`` for SciPy methods other than the quasi-Newton loop the kernels run
(reference: src/slam/optimizer.py:266-268 hands ``override_method`` to ``scipy.optimize.minimize``;
scripts/cost_function_comparison.ipynb uses ``"Nelder-Mead"``).

Derivative-free methods are driven from the host, all (target, restart) items of a span in lock-step, with the objective --
``CircuitTemplate.eval`` + the cost function, the whole hot path's arithmetic -- evaluated for the whole batch by the device
(``slam_eval_loss_grad`` without the gradient): one kernel launch per simplex operation instead of one Python callback per item.
There is no CPU evaluation path here either.

``nelder_mead_batch`` restates SciPy's ``_minimize_neldermead`` (non-adaptive coefficients rho = 1, chi = 2, psi = 1/2, sigma = 1/2;
initial simplex x0 and x0 with one coordinate scaled by 1.05 (0.00025 where it is zero); stop when the simplex is within
``xatol`` and the values within ``fatol``, or at ``maxiter``), vectorised over the items: an item takes exactly the decisions SciPy
takes from the same function values.
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np

SUPPORTED = ("Nelder-Mead",)


def nelder_mead_batch(fun: Callable[[np.ndarray, np.ndarray], np.ndarray], x0: np.ndarray, maxiter: int = 2500, xatol: float = 1e-4,
                      fatol: float = 1e-4) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """Minimise ``fun`` for every row of ``x0[M, n]`` with the Nelder-Mead simplex method.

    ``fun(items, X)`` returns the objective of item ``items[j]`` at ``X[j]`` (any subset of the items, in any multiplicity).
    Returns ``(x[M, n], f[M], iterations[M], evaluations[M])``."""
    x0 = np.asarray(x0, dtype=np.float64)
    M, n = x0.shape
    rho, chi, psi, sigma = 1.0, 2.0, 0.5, 0.5
    sim = np.repeat(x0[:, None, :], n + 1, axis=1)  # [M, n + 1, n]
    idx = np.arange(n)
    d = sim[:, 1:, :][:, idx, idx]
    sim[:, 1:, :][:, idx, idx] = np.where(d != 0.0, (1.0 + 0.05) * d, 0.00025)
    items_all = np.arange(M)
    fsim = fun(np.repeat(items_all, n + 1), sim.reshape(M * (n + 1), n)).reshape(M, n + 1)
    nfev = np.full(M, n + 1, dtype=np.int64)
    order = np.argsort(fsim, axis=1, kind="stable")
    fsim = np.take_along_axis(fsim, order, axis=1)
    sim = np.take_along_axis(sim, order[:, :, None], axis=1)
    nit = np.ones(M, dtype=np.int64)
    active = np.ones(M, dtype=bool)
    while True:
        conv = (np.abs(sim[:, 1:, :] - sim[:, :1, :]).max(axis=(1, 2)) <= xatol) & (np.abs(fsim[:, :1] - fsim[:, 1:]).max(axis=1) <= fatol)
        active &= ~conv & (nit < maxiter)
        a = np.nonzero(active)[0]
        if len(a) == 0:
            break
        S, F = sim[a], fsim[a]
        xbar = S[:, :-1, :].sum(axis=1) / n
        worst = S[:, -1, :]
        xr = (1 + rho) * xbar - rho * worst
        fxr = fun(a, xr)
        nfev[a] += 1
        new_x, new_f = worst.copy(), F[:, -1].copy()
        shrink = np.zeros(len(a), dtype=bool)
        expand = fxr < F[:, 0]
        reflect = ~expand & (fxr < F[:, -2])
        contract_out = ~expand & ~reflect & (fxr < F[:, -1])
        contract_in = ~expand & ~reflect & ~contract_out
        # second evaluation of the iteration, all kinds in one device call
        x2 = np.where(expand[:, None], (1 + rho * chi) * xbar - rho * chi * worst,
                      np.where(contract_out[:, None], (1 + psi * rho) * xbar - psi * rho * worst, (1 - psi) * xbar + psi * worst))
        need2 = expand | contract_out | contract_in
        f2 = np.full(len(a), np.inf)
        if need2.any():
            j = np.nonzero(need2)[0]
            f2[j] = fun(a[j], x2[j])
            nfev[a[j]] += 1
        take2 = (expand & (f2 < fxr)) | (contract_out & (f2 <= fxr)) | (contract_in & (f2 < F[:, -1]))
        take_r = (expand & ~(f2 < fxr)) | reflect
        shrink = (contract_out & ~(f2 <= fxr)) | (contract_in & ~(f2 < F[:, -1]))
        new_x = np.where(take2[:, None], x2, np.where(take_r[:, None], xr, new_x))
        new_f = np.where(take2, f2, np.where(take_r, fxr, new_f))
        S[:, -1, :], F[:, -1] = new_x, new_f
        if shrink.any():
            j = np.nonzero(shrink)[0]
            S[j, 1:, :] = S[j, :1, :] + sigma * (S[j, 1:, :] - S[j, :1, :])
            F[j, 1:] = fun(np.repeat(a[j], n), S[j, 1:, :].reshape(len(j) * n, n)).reshape(len(j), n)
            nfev[a[j]] += n
        order = np.argsort(F, axis=1, kind="stable")
        fsim[a] = np.take_along_axis(F, order, axis=1)
        sim[a] = np.take_along_axis(S, order[:, :, None], axis=1)
        nit[a] += 1
    return sim[:, 0, :].copy(), fsim[:, 0].copy(), nit, nfev
