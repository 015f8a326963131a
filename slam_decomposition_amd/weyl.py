"""Weyl-chamber coordinates on the host (two calls per target: SURVEY.md §8(a) A10).

Restates ``weylchamber.c1c2c3`` (called at src/slam/basis_abc.py:80-84 and
src/slam/optimizer.py:85,103); a batched device version is a "next" row (§8(f) rank 1).
"""
from __future__ import annotations

import numpy as np

_SY = np.array([[0, -1j], [1j, 0]], dtype=np.complex128)
_YY = np.kron(_SY, _SY)
_M = np.array([[1, 1, 0], [1, 0, 1], [0, 1, 1]])


def c1c2c3(U, ndigits: int = 8):
    """(c1, c2, c3) in units of pi, rounded to ``ndigits`` like weylchamber does."""
    U = np.asarray(U, dtype=np.complex128)
    Ut = _YY @ U.T @ _YY
    ev = np.linalg.eigvals(U @ Ut / np.sqrt(complex(np.linalg.det(U))))
    two_S = np.angle(ev) / np.pi
    two_S = np.where(two_S <= -0.5, two_S + 2.0, two_S)
    S = np.sort(two_S / 2.0)[::-1]
    n = int(round(float(S.sum())))
    S = S - np.r_[np.ones(n), np.zeros(4 - n)]
    S = np.roll(S, -n)
    c1, c2, c3 = _M @ S[:3]
    if c3 < 0:
        c1 = 1 - c1
        c3 = -c3
    return tuple(float(round(v + 0.0, ndigits) + 0.0) for v in (c1, c2, c3))


def c1c2c3_batch(U, ndigits: int = 8) -> np.ndarray:
    """:func:`c1c2c3` for a stack of unitaries ``U[N, 4, 4]`` -> float64[N, 3] (same values, one batched
    ``eigvals`` call instead of N)."""
    U = np.asarray(U, dtype=np.complex128)
    if U.ndim != 3 or U.shape[1:] != (4, 4):
        raise ValueError("expected an array of shape [N, 4, 4]")
    if U.shape[0] == 0:
        return np.zeros((0, 3))
    Ut = _YY @ np.swapaxes(U, 1, 2) @ _YY
    det = np.linalg.det(U).astype(np.complex128)
    ev = np.linalg.eigvals(U @ Ut / np.sqrt(det)[:, None, None])
    two_S = np.angle(ev) / np.pi
    two_S = np.where(two_S <= -0.5, two_S + 2.0, two_S)
    S = -np.sort(-two_S / 2.0, axis=1)  # descending
    n = np.rint(S.sum(axis=1)).astype(int)
    out = np.empty((U.shape[0], 3))
    for nn in np.unique(n):  # n is 0..2 in practice: a handful of groups
        m = n == nn
        Sg = S[m] - np.r_[np.ones(nn), np.zeros(4 - nn)]
        Sg = np.roll(Sg, -nn, axis=1)
        out[m] = Sg[:, :3] @ _M.T
    flip = out[:, 2] < 0
    out[flip, 0] = 1 - out[flip, 0]
    out[flip, 2] = -out[flip, 2]
    return np.round(out + 0.0, ndigits) + 0.0
