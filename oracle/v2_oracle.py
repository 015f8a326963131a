"""CPU restatement of the CircuitTemplateV2 path (TEST INFRASTRUCTURE ONLY, see ``oracle/__init__.py``).

Reference: src/slam/basisv2.py:27-299 (template whose 2Q gates are classes / lambdas with their own "Q" parameters,
optional box bounds), src/slam/optimizer.py:253-278 (``scipy.optimize.minimize`` with ``method="L-BFGS-B"`` when the
template has bounds, ``"BFGS"`` otherwise, no ``jac``), src/slam/utils/gates/custom_gates.py:163-212,534-606 and
src/slam/hamiltonian.py:84-111 (the gates).  Parity unpinned by reference tests (there are none); pinned by the recorded
V2 + SquareCost run of scripts/decomp_trajectory.ipynb:84-90,140-162 (KAT-1: RiSwapGate class, every Q bounded to
[0.5, 0.5], target SWAP), which ``tests/test_oracle_kat.py`` reproduces through this module.

Parameter vectors are in *index order*: the 1Q parameters ``P0..`` (6 per layer, or 2 with ``vz_only``), then the
parameters of gate 1, gate 2, ...
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import scipy.optimize as opt

from . import slam_oracle as o

DEFAULT_BOUND = (-4 * np.pi, 4 * np.pi)  # basisv2.py:160


def rz(lam: float) -> np.ndarray:
    """qiskit RZGate (basisv2.py:256-259 with vz_only)."""
    return np.array([[np.exp(-0.5j * lam), 0], [0, np.exp(0.5j * lam)]], dtype=np.complex128)


def template_eval(x, gate_fns: Sequence[Callable[..., np.ndarray]], qn: int, k: int, vz_only: bool = False) -> np.ndarray:
    """W = K_k G_k(q_k) ... G_1(q_1) K_0; ``gate_fns[j](*q)`` returns the 4x4 matrix of gate j + 1."""
    x = np.asarray(x, dtype=np.float64)
    npl = 2 if vz_only else 6
    n_p = npl * (k + 1)
    assert x.size == n_p + qn * k

    def layer(j):
        xs = x[npl * j : npl * (j + 1)]
        if vz_only:
            return np.kron(rz(xs[1]), rz(xs[0]))
        return o.layer_matrix(xs)

    W = layer(0)
    for j in range(k):
        q = x[n_p + qn * j : n_p + qn * (j + 1)]
        W = layer(j + 1) @ np.asarray(gate_fns[j](*q), dtype=np.complex128) @ W
    return W


def loss(x, gate_fns, qn, k, target, vz_only=False, square=False) -> float:
    W = template_eval(x, gate_fns, qn, k, vz_only)
    return o.square_cost(W, target) if square else o.basic_cost(W, target)


def fd_grad(x, gate_fns, qn, k, target, vz_only=False, square=False, h=1e-6) -> np.ndarray:
    x = np.asarray(x, dtype=np.float64)
    g = np.zeros_like(x)
    for i in range(x.size):
        e = np.zeros_like(x)
        e[i] = h
        g[i] = (loss(x + e, gate_fns, qn, k, target, vz_only, square) - loss(x - e, gate_fns, qn, k, target, vz_only, square)) / (2 * h)
    return g


def cg_matrix(raw) -> np.ndarray:
    """G(a, phi_c, b, phi_g): the conversion-gain closed form (hamiltonian.py:84-111) with a = gc t, b = gg t."""
    a, pc, b, pg = (float(v) for v in raw)
    return o.conversion_gain_matrix(pc, pg, a, b, 1.0)


def cg_dmatrix(raw) -> List[np.ndarray]:
    """dG/d(a, phi_c, b, phi_g), entry by entry from the closed form."""
    a, pc, b, pg = (float(v) for v in raw)
    d = [np.zeros((4, 4), dtype=np.complex128) for _ in range(4)]
    ca, sa, cb, sb = np.cos(a), np.sin(a), np.cos(b), np.sin(b)
    d[0][1, 1] = d[0][2, 2] = -sa
    d[0][2, 1] = -1j * np.exp(1j * pc) * ca
    d[0][1, 2] = -1j * np.exp(-1j * pc) * ca
    d[1][2, 1] = np.exp(1j * pc) * sa          # d/dphi (-i e^{i phi} s) = e^{i phi} s
    d[1][1, 2] = -np.exp(-1j * pc) * sa        # d/dphi (-i e^{-i phi} s) = -e^{-i phi} s
    d[2][0, 0] = d[2][3, 3] = -sb
    d[2][3, 0] = -1j * np.exp(1j * pg) * cb
    d[2][0, 3] = -1j * np.exp(-1j * pg) * cb
    d[3][3, 0] = np.exp(1j * pg) * sb
    d[3][0, 3] = -np.exp(-1j * pg) * sb
    return d


def raw_of(q, gmap) -> np.ndarray:
    """raw[r] = scale[r] * q[sel[r]] + offset[r]  (sel = -1: constant); gmap = (sel, scale, offset)."""
    sel, scale, offset = gmap
    return np.array([offset[r] + (scale[r] * q[sel[r]] if sel[r] >= 0 else 0.0) for r in range(4)])


def loss_and_grad(x, gmaps, qn, k, target, vz_only=False, square=False) -> Tuple[float, np.ndarray]:
    """Loss and analytic gradient with respect to every parameter for gates of the conversion-gain family given by their
    angle maps (one (sel, scale, offset) per gate): dL/dx_i = -Re(conj(t) Tr(T^+ d_i W)) / (4 |t|), t = Tr(T^+ W)."""
    x = np.asarray(x, dtype=np.float64)
    npl = 2 if vz_only else 6
    n_p = npl * (k + 1)
    qs = [x[n_p + qn * j : n_p + qn * (j + 1)] for j in range(k)]
    Gs = [cg_matrix(raw_of(qs[j], gmaps[j])) for j in range(k)]

    def layer(j):
        xs = x[npl * j : npl * (j + 1)]
        return np.kron(rz(xs[1]), rz(xs[0])) if vz_only else o.layer_matrix(xs)

    Ks = [layer(j) for j in range(k + 1)]
    right = [np.eye(4, dtype=np.complex128)]
    for j in range(1, k + 1):
        right.append(Gs[j - 1] @ Ks[j - 1] @ right[j - 1])
    left = [None] * (k + 1)
    left[k] = np.eye(4, dtype=np.complex128)
    for j in range(k - 1, -1, -1):
        left[j] = left[j + 1] @ Ks[j + 1] @ Gs[j]
    W = Ks[k] @ right[k]
    Th = np.asarray(target).conj().T
    t = np.trace(Th @ W)
    at = abs(t)
    val = 1.0 - at / 4.0
    grad = np.zeros(x.size)

    def dl(dW):
        return -np.real(np.conj(t) * np.trace(Th @ dW)) / (4.0 * at)

    for j in range(k + 1):
        xs = x[npl * j : npl * (j + 1)]
        if vz_only:
            drz = [np.diag([-0.5j * np.exp(-0.5j * v), 0.5j * np.exp(0.5j * v)]) for v in xs]
            grad[npl * j] = dl(left[j] @ np.kron(rz(xs[1]), drz[0]) @ right[j])
            grad[npl * j + 1] = dl(left[j] @ np.kron(drz[1], rz(xs[0])) @ right[j])
        else:
            A, B = o.u3(*xs[3:6]), o.u3(*xs[0:3])
            dB, dA = o._du3(*xs[0:3]), o._du3(*xs[3:6])
            for m in range(3):
                grad[6 * j + m] = dl(left[j] @ np.kron(A, dB[m]) @ right[j])
                grad[6 * j + 3 + m] = dl(left[j] @ np.kron(dA[m], B) @ right[j])
    for j in range(1, k + 1):  # gate j
        sel, scale, offset = gmaps[j - 1]
        dG = cg_dmatrix(raw_of(qs[j - 1], gmaps[j - 1]))
        pre, post = left[j] @ Ks[j], Ks[j - 1] @ right[j - 1]
        for r in range(4):
            if sel[r] >= 0:
                grad[n_p + qn * (j - 1) + sel[r]] += scale[r] * dl(pre @ dG[r] @ post)
    if square:  # SquareCost = 0.8 (2 L - L^2) of BasicCost L (cost_function.py:169-173)
        return 0.8 * val * (2.0 - val), 1.6 * (1.0 - val) * grad
    return float(val), grad


def run_reference(
    target: np.ndarray,
    gate_fns_of_k: Callable[[int], List[Callable[..., np.ndarray]]],
    qn: int,
    spanning_range,
    training_restarts: int,
    success_threshold: float,
    bounds_of_k: Callable[[int], Optional[List[Tuple[Optional[float], Optional[float]]]]],
    x0_fn: Callable[[int, int], np.ndarray],
    vz_only: bool = False,
    square: bool = False,
):
    """``TemplateOptimizer._run`` (optimizer.py:188-313) for a V2 template: L-BFGS-B when ``bounds_of_k(k)`` is a list
    (basis.using_bounds), BFGS otherwise; SciPy's finite-difference gradient, as the reference passes no ``jac``."""
    best, best_x, best_k = None, None, -1
    for k in spanning_range:
        fns = gate_fns_of_k(k)
        bounds = bounds_of_k(k)
        for r in range(training_restarts):
            res = opt.minimize(
                fun=lambda xx: loss(xx, fns, qn, k, target, vz_only, square),
                method="L-BFGS-B" if bounds is not None else "BFGS",
                x0=x0_fn(k, r),
                options={"maxiter": 2500},
                bounds=bounds,
            )
            if best is None or res.fun < best:
                best, best_x, best_k = float(res.fun), res.x, k
            if best < success_threshold:
                break
        if best < success_threshold:
            break
    return best, best_x, best_k
