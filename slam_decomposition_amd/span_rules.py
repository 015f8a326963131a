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

For every other template -- a basis gate outside those classes (the ConversionGain(0, 0, gc, gg, 1) family of config 5),
or a SEQUENCE of different gates (``[iSWAP, B]``, the territory of ``MixedOrderBasisCircuitTemplate``, basis.py:213-359) --
rounds 2-3 had a LOWER bound on the template size (superseded by ``coverage.py``, see the end; ``strength`` stays as a helper):

* 0 gates iff the target is local; 1 gate iff the target is in the first gate's own class (exact);
* k >= 2 gates g_1 .. g_k only if  m(T) <= m(g_1) + ... + m(g_k)  for the two interaction-strength measures
  ``m1 = x + y + |z|`` and ``m2 = max(x, (x + y + |z|) / 2)`` of the folded coordinates (x >= y >= |z|, x <= 1/2).
  m1 and m2 are the minimal times to simulate the gate with the Hamiltonians XX and XX + YY and fast local unitaries
  (Childs, Haselgrove, Nielsen, PRA 68, 052311: optimal simulation time = the smallest t with a representative of the
  canonical parameters s-majorised by t lambda(H); the folded representative minimises both expressions), and a
  simulation time is subadditive under composition with local gates in between ("chaining").

Round 4 -- exact coverage of TWO-gate products beyond the single-gate classes (``sequence_minimal_span``).  The set of Weyl
coordinates of ``g_2 L g_1`` over all local unitaries L is what monodromy's polytope for the pair describes; for the pairs below it
was obtained here by sampling L over SU(2) x SU(2) (4e5 draws, ``tools/fit_two_gate_region.py``), reading off the supporting
half-spaces in the folded coordinates (x >= y >= |z|, x <= 1/2) and checking that the polytope they cut out is FILLED (no empty cell
of a 0.02 grid inside, no sample outside) -- and then against the brute-force span loop on the GPU (tests/test_gpu_round4.py):

* iSWAP . L . B (either order):  x >= 1/4  and  |z| <= 1/4   -- 97.75 % of the Haar measure, the fraction of targets the
  brute-force loop of config 3 ([iSWAP, B, iSWAP][:k]) solves with two gates;
* g . L . g for an XY-type gate g = (a, a, 0), a <= 1/4 (``RiSwapGate(alpha)``, alpha = 4a <= 1; zero-phase conversion-only or
  gain-only ``ConversionGainGate``):  |z| <= x - y,  x + y + |z| <= 4a,  x <= 2a   (a = 1/4: the sqrt(iSWAP) rule above).

Round 4, later -- the general case (``coverage.py``): the coverage set of ANY list of gates from the inequalities of the
multiplicative eigenvalue problem for SU(4) (what monodromy computes for the reference), 14 half-spaces in the target's alcove
coordinates per circuit.  It reproduces every rule above with no mismatch and replaces the strength bounds: ``two_gate_region``
falls back to it, ``sequence_minimal_span`` / ``span_lower_bound`` / ``multiset_coverage`` are exact for every gate.  The closed
forms stay as the fast path for their classes and as an independent check of the general code (tests/test_coverage.py).

``span_lower_bound`` is sound (a target is never placed above its true size: tests/test_gpu_round3.py checks it against the
brute-force span loop on the config-4 and config-5 bases) but not tight: the span loop starts at the bound instead of at 1
and targets whose bound exceeds the template's maximum are not optimised at all.

Coordinates are in units of pi, as returned by ``weylchamber.c1c2c3`` (8 digits).
"""
from __future__ import annotations

import numpy as np

from . import coverage

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


def _unfold(c: np.ndarray) -> np.ndarray:
    """Inverse of ``_fold``: (x, y, z) with x <= 1/2, z of either sign -> (c1, c2, c3) with c3 >= 0."""
    c = np.array(c, dtype=np.float64, copy=True).reshape(-1, 3)
    m = c[:, 2] < 0
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


def strength(coords) -> np.ndarray:
    """(m1, m2) per row: interaction-strength measures of Weyl coordinates (subadditive under composition with locals)."""
    c = np.abs(_fold(coords))
    c = -np.sort(-c, axis=1)  # x >= y >= |z|
    m1 = c.sum(axis=1)
    return np.stack([m1, np.maximum(c[:, 0], 0.5 * m1)], axis=1)


def span_lower_bound(target_coords, gate_coords_seq, k_max=None, slack: float = 4 * _TOL) -> np.ndarray:
    """(Since ``coverage.py`` the bound is the exact size; the name stays for its callers.)  Lower bound on the number of leading gates of the template sequence ``gate_coords_seq`` (Weyl coordinates of
    g_1, g_2, ... in circuit order) that reach each target: int array [N] with values 0 .. k_max + 1 (k_max + 1 = not
    reachable with the whole sequence).  ``slack`` (in coordinate units) widens the test: a caller that accepts a loss
    below L as success accepts targets up to about sqrt(L) outside the reachable set (|coordinate error| ~ sqrt(loss))."""
    g = np.asarray(gate_coords_seq, dtype=np.float64).reshape(-1, 3)
    k_max = len(g) if k_max is None else int(k_max)
    if k_max > len(g):
        raise ValueError("gate sequence shorter than k_max")
    c = _fold(target_coords)
    n = len(c)
    local = np.max(np.abs(c), axis=1) < _TOL
    gf = _fold(g[:1])[0]
    same = np.max(np.abs(np.abs(c) - np.abs(gf)), axis=1) < _TOL
    lb = np.full(n, k_max + 1, dtype=np.int64)
    sums = None
    for k in range(k_max, 1, -1):
        closed = _closed_two_gate_region(g[0], g[1]) if k == 2 else None
        if k >= 3 and _reaches_everything(g[:k]):
            ok = np.ones(n, dtype=bool)
        elif closed is not None:
            ok = closed(c[:, 0], c[:, 1], c[:, 2], _TOL + slack)
        else:
            if sums is None:
                sums = coverage.target_sums(_unfold(c))  # the targets' side of the half-spaces, once for all k
            ok = coverage.contains(None, g[:k], _TOL + slack, sums=sums)  # exact (it implies the strength test of round 3)
        lb = np.where(ok, k, lb)
    lb = np.where(same, 1, lb)
    lb = np.where(local, 0, lb)
    return lb


def _is(g, ref) -> bool:
    return bool(np.max(np.abs(np.abs(_fold(g)[0]) - np.array(ref))) < _TOL)


def two_gate_region(g1, g2):
    """Exact coverage of the two-gate product g_2 L g_1 (L any local unitary) in the folded chamber: a function
    ``(x, y, z arrays, tol) -> bool array`` -- the closed form for the pairs that have one here, ``coverage.contains`` for the rest.
    Symmetric in (g1, g2): transposition swaps the order and leaves the Weyl coordinates alone."""
    closed = _closed_two_gate_region(g1, g2)
    if closed is not None:
        return closed
    gg = np.array([np.ravel(g1), np.ravel(g2)], dtype=np.float64)
    return lambda x, y, z, tol: coverage.contains(_unfold(np.stack([x, y, z], axis=1)), gg, tol)  # any other pair: coverage.py


def _closed_two_gate_region(g1, g2):
    """The pairs with a closed-form region (or None)."""
    a1, a2 = np.abs(_fold(g1)[0]), np.abs(_fold(g2)[0])
    same = bool(np.max(np.abs(a1 - a2)) < _TOL)
    if same:
        if _is(g1, FAMILIES["cx"]) or _is(g1, FAMILIES["iswap"]):
            return lambda x, y, z, tol: np.abs(z) < tol  # the c3 = 0 face
        if _is(g1, FAMILIES["b"]):
            return lambda x, y, z, tol: np.ones(len(x), dtype=bool)
        a = float(a1[0])
        if abs(a1[1] - a) < _TOL and a1[2] < _TOL and _TOL < a <= 0.25 + _TOL:  # XY-type (a, a, 0), a <= 1/4
            return lambda x, y, z, tol: (np.abs(z) <= x - y + tol) & (x + y + np.abs(z) <= 4 * a + tol) & (x <= 2 * a + tol)
    pair = (_is(g1, FAMILIES["iswap"]) and _is(g2, FAMILIES["b"])) or (_is(g1, FAMILIES["b"]) and _is(g2, FAMILIES["iswap"]))
    if pair:
        return lambda x, y, z, tol: (x >= 0.25 - tol) & (np.abs(z) <= 0.25 + tol)
    return None


_UNIVERSAL_IN_3 = ("cx", "iswap", "sqiswap", "b")


def _reaches_everything(g) -> bool:
    """Shortcut for three or more gates (``coverage.contains`` says the same, tests/test_coverage.py): two of them B (B L B is the
    whole chamber), all of ONE class among CX / iSWAP / sqrt(iSWAP) / B, or any mixture of iSWAP and B."""
    if len(g) < 3:
        return False
    classes = [next((f for f in _UNIVERSAL_IN_3 if _is(gi, FAMILIES[f])), None) for gi in g]
    return classes.count("b") >= 2 or (None not in classes and (len(set(classes)) == 1 or set(classes) <= {"iswap", "b"}))
MAX_EXACT_SPAN = 8  # (any length works -- coverage.region is linear in it; the kernels stop at 5)


def sequence_is_exact(gate_coords_seq, k_max: int) -> bool:
    """True when ``sequence_minimal_span`` knows the exact template size of every target for the first ``k_max`` gates of the
    sequence -- since ``coverage.py``: for every gate sequence."""
    g = np.asarray(gate_coords_seq, dtype=np.float64).reshape(-1, 3)
    return 1 <= k_max <= min(len(g), MAX_EXACT_SPAN)


def sequence_minimal_span(target_coords, gate_coords_seq, k_max: int, slack: float = 0.0) -> np.ndarray:
    """Exact number of leading gates of the sequence that reaches each target (0 = local, 1 = the first gate's class, 2 = inside
    the pair's region, ...); targets beyond ``k_max`` get ``k_max + 1``.  ``slack`` widens the regions (a caller that accepts
    loss < L accepts targets ~ sqrt(L) outside them)."""
    g = np.asarray(gate_coords_seq, dtype=np.float64).reshape(-1, 3)
    if not sequence_is_exact(g, k_max):
        raise NotImplementedError("gate sequence shorter than k_max, or longer than MAX_EXACT_SPAN")
    c = _fold(target_coords)
    x, y, z = c[:, 0], c[:, 1], c[:, 2]
    tol = _TOL + slack
    local = np.max(np.abs(c), axis=1) < _TOL
    gf = np.abs(_fold(g[:1])[0])
    same = np.max(np.abs(np.abs(c) - gf), axis=1) < _TOL
    k = np.full(len(c), k_max + 1, dtype=np.int64)
    for kk in range(k_max, 2, -1):
        if _reaches_everything(g[:kk]):
            k[:] = kk
        else:
            k = np.where(coverage.contains(_unfold(c), g[:kk], tol), kk, k)
    if k_max >= 2:
        k = np.where(two_gate_region(g[0], g[1])(x, y, z, tol), 2, k)
    k = np.where(same, 1, k)
    k = np.where(local, 0, k)
    return k


def multiset_coverage(target_coords, gate_coords_list, slack: float = 4 * _TOL):
    """Coverage test for a circuit of the gates ``gate_coords_list`` (any order -- transposition and inversion map the Weyl
    coordinates of a product onto those of the reordered one -- with free local gates in between): ``(inside, exact)``.
    ``inside`` is the coverage set itself (what ``CircuitPolytope.has_element`` answers in the reference, polytope_wrap.py:78-90);
    ``exact`` is True for every gate list since ``coverage.py`` (it was False where only the strength bound was known).  Local targets are reported outside (the reference handles
    them before the lookup, polytope_wrap.py:53-54)."""
    g = np.asarray(gate_coords_list, dtype=np.float64).reshape(-1, 3)
    k = len(g)
    if k < 1:
        raise ValueError("a coverage entry needs at least one gate")
    c = _fold(target_coords)
    x, y, z = c[:, 0], c[:, 1], c[:, 2]
    tol = _TOL + slack
    local = np.max(np.abs(c), axis=1) < _TOL
    if k == 1:
        gf = np.abs(_fold(g[:1])[0])
        return (np.max(np.abs(np.abs(c) - gf), axis=1) < _TOL) & ~local, True
    if k == 2:
        return two_gate_region(g[0], g[1])(x, y, z, tol) & ~local, True
    if _reaches_everything(g):
        return ~local, True
    return coverage.contains(_unfold(c), g, tol) & ~local, True
