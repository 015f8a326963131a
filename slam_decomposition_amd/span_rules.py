"""Analytic minimal-span rules: the stand-in for the reference's ``use_polytopes=True`` mode.

The reference asks the ``monodromy`` package for the first precomputed coverage polytope that contains the
target's Weyl coordinates and optimises only at that template size
(``monodromy_range_from_target``, src/slam/utils/polytopes/polytope_wrap.py:39-94, used by
``CircuitTemplate.get_spanning_range``, src/slam/basis.py:95-100).  ``monodromy`` (an un-vendored fork) and
its precomputed coverage sets cannot be loaded here; for the basis gates whose coverage regions are known in
closed form the same answer comes from a few comparisons on (c1, c2, c3):

* CX class (0.5, 0, 0): 0 gates for a local target, 1 for the CX class, 2 iff c3 = 0, otherwise 3
  (Shende-Bullock-Markov / Vidal-Dawson);
* iSWAP class (0.5, 0.5, 0): the same thresholds -- two applications also span exactly the c3 = 0 face
  (checked against the brute-force span loop on the GPU, tests/test_gpu_api.py);
* sqrt(iSWAP) class (0.25, 0.25, 0): 2 iff |z| <= x - y in the folded chamber x >= y >= |z|, x <= 1/2 --
  the test of the reference's own ``riswapWeylDecomp``
  (src/slam/utils/transpiler_pass/weyl_decompose.py:343-387, arXiv:2105.06074) -- otherwise 3;
* B class (0.5, 0.25, 0): every non-local target in 2 (Zhang et al., PRL 93, 020502).

Coordinates are in units of pi, as returned by ``weylchamber.c1c2c3`` (8 digits).
"""
from __future__ import annotations

import numpy as np

_TOL = 2e-8  # coordinates are rounded to 8 digits

FAMILIES = {
    "cx": (0.5, 0.0, 0.0),
    "iswap": (0.5, 0.5, 0.0),
    "sqiswap": (0.25, 0.25, 0.0),
    "b": (0.5, 0.25, 0.0),
}


def _fold(coords: np.ndarray) -> np.ndarray:
    """(c1, c2, c3) with c3 >= 0, c1 in [0, 1]  ->  (x, y, z) with x <= 1/2 (z may become negative)."""
    c = np.array(coords, dtype=np.float64, copy=True).reshape(-1, 3)
    m = c[:, 0] > 0.5
    c[m, 0] = 1.0 - c[m, 0]
    c[m, 2] = -c[m, 2]
    return c


def family_of(gate_coords) -> str:
    """Name of the supported family a basis gate with these Weyl coordinates belongs to."""
    g = _fold(gate_coords)[0]
    for name, ref in FAMILIES.items():
        if np.max(np.abs(np.abs(g) - np.array(ref))) < _TOL:
            return name
    raise NotImplementedError(
        "use_polytopes: coverage polytopes (monodromy) are not available; analytic span rules exist for basis gates "
        f"in the classes {sorted(FAMILIES)} only (got Weyl coordinates {tuple(float(v) for v in np.ravel(gate_coords))})"
    )


def minimal_span(target_coords, gate_coords) -> np.ndarray:
    """Smallest number of applications of the basis gate that reaches each target: int array [N]."""
    fam = family_of(gate_coords)
    c = _fold(target_coords)
    x, y, z = c[:, 0], c[:, 1], c[:, 2]
    local = (np.abs(x) < _TOL) & (np.abs(y) < _TOL) & (np.abs(z) < _TOL)
    same = np.max(np.abs(np.abs(c) - np.array(FAMILIES[fam])), axis=1) < _TOL
    if fam in ("cx", "iswap"):
        k = np.where(np.abs(z) < _TOL, 2, 3)
    elif fam == "sqiswap":
        k = np.where(np.abs(z) <= x - y + _TOL, 2, 3)
    else:  # b
        k = np.full(len(c), 2)
    k = np.where(same, 1, k)
    k = np.where(local, 0, k)
    return k.astype(np.int64)
