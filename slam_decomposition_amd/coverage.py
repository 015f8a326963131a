"""Exact coverage sets of two-qubit circuits: which targets ``g_k L_{k-1} ... L_1 g_1`` reaches over all local gates ``L_j``.

This is what the reference asks the ``monodromy`` package for (``monodromy_range_from_target`` /
``get_polytope_from_circuit`` / ``gate_set_to_coverage``, src/slam/utils/polytopes/polytope_wrap.py:39-196, used by
``CircuitTemplate.get_spanning_range`` with ``use_polytopes=True``, src/slam/basis.py:95-100, and by
``MixedOrderBasisCircuitTemplate``, basis.py:213-359).  ``monodromy`` (an un-vendored fork with the ``lrs`` vertex enumerator) is
not available; the membership test itself is short once the inequalities are known, and they are computed here from first principles:

* A two-qubit gate is, up to local gates, ``CAN(c1, c2, c3) = exp(i pi/2 (c1 XX + c2 YY + c3 ZZ))``; in the magic basis it is
  diagonal with phases ``pi a_j``, ``a = ((c1+c2-c3)/2, (c1-c2+c3)/2, (-c1+c2+c3)/2, (-c1-c2-c3)/2)`` -- the Cartan projection of the
  symmetric space SU(4)/SO(4), whose restricted root system is that of SU(4) itself.  Folded into the alcove
  ``a_1 >= a_2 >= a_3 >= a_4 >= a_1 - 1, sum a = 0`` these are the "monodromy coordinates" (Peterson, Crooks, Smith, Quantum 4, 247
  (2020)); multiplying the gate by i shifts every ``a_j`` by 1/2, so a gate CLASS has two alcove points.
* The products of classes are governed by the multiplicative eigenvalue problem (Agnihotri & Woodward, Math. Res. Lett. 5 (1998);
  Belkale, Compositio Math. 129 (2001)): ``A_1 A_2 ... A_s = C`` is solvable in SU(n) with alcove spectra ``alpha^(l)``, ``gamma`` iff
  for every ``0 < r < n``, every choice of r-subsets ``I_1 .. I_s, K`` of ``{1..n}`` and degree ``d`` with non-vanishing
  Gromov-Witten invariant ``<sigma_{I_1}, ..., sigma_{I_s}, sigma_K>_d`` of the Grassmannian Gr(r, n)

      sum_l sum_{i in I_l} alpha^(l)_i  -  sum_{k in K} gamma_{n+1-k}  <=  d

  (subset ``I = {i_1 < .. < i_r}`` <-> Schubert class of the partition ``lambda_j = n - r + j - i_j``).  For n = 4 the small quantum
  cohomology rings of Gr(1, 4) = P^3, Gr(3, 4) and Gr(2, 4) are tiny tables (below; the Gr(2, 4) table is checked for associativity at
  import), and the invariant is the coefficient of ``q^d [point]`` in the product of the classes.  s = 2 gives 72 inequalities.

Pinned against the reference's own data (tests/test_coverage.py::test_regions_equal_the_coverage_sets_the_reference_ships): the
coverage sets it ships precomputed (src/slam/data/polytopes/polytope_coverage_[...].pkl, monodromy's output for 17 ConversionGainGate
bases, circuits of up to 26 gates; their inequality rows are the fixture tests/golden/reference_coverage_polytopes.json) contain
exactly the targets ``contains`` says, entry by entry -- the reference's monodromy coordinates are the first three alcove coordinates.

Also checked (tests/test_coverage.py, no GPU): every inequality holds -- and is attained to 1e-4 -- on random products in SU(4) and on
random ``CAN . L . CAN (. L . CAN)`` circuits; the regions reproduce every closed-form rule of ``span_rules`` (CX / iSWAP /
sqrt(iSWAP) / B classes, ``iSWAP . L . B``, XY-type pairs) with no mismatch; sampled circuits fill the predicted regions.  On the GPU
(tests/test_gpu_round4.py): the template size predicted for conversion-gain gates of BASELINE configs[4] equals the size the
brute-force span loop finds.

Coordinates are ``weylchamber.c1c2c3`` triples in units of pi.
"""
from __future__ import annotations

import itertools
from functools import lru_cache
from typing import Dict, List, Sequence, Tuple

import numpy as np

_N = 4

# ---- small quantum cohomology of Gr(r, 4) ------------------------------------------------------------------------------------------
# a class is a partition (tuple); a ring element is {(d, partition): coefficient} (d = power of q)
_E, _S1, _S2, _S11, _S21, _S22 = (0, 0), (1, 0), (2, 0), (1, 1), (2, 1), (2, 2)
_GR24 = [_E, _S1, _S2, _S11, _S21, _S22]
_T24: Dict[Tuple, Dict] = {}


def _set(x, y, res):
    _T24[(x, y)] = res
    _T24[(y, x)] = res


for _b in _GR24:
    _set(_E, _b, {(0, _b): 1})
# quantum Pieri for Gr(2, 4) (Bertram): sigma_1 . sigma_21 = sigma_22 + q, sigma_1 . sigma_22 = q sigma_1, ...
_set(_S1, _S1, {(0, _S2): 1, (0, _S11): 1})
_set(_S1, _S2, {(0, _S21): 1})
_set(_S1, _S11, {(0, _S21): 1})
_set(_S1, _S21, {(0, _S22): 1, (1, _E): 1})
_set(_S1, _S22, {(1, _S1): 1})
_set(_S2, _S2, {(0, _S22): 1})
_set(_S2, _S11, {(1, _E): 1})
_set(_S11, _S11, {(0, _S22): 1})
_set(_S2, _S21, {(1, _S1): 1})
_set(_S11, _S21, {(1, _S1): 1})
_set(_S2, _S22, {(1, _S11): 1})
_set(_S11, _S22, {(1, _S2): 1})
_set(_S21, _S21, {(1, _S2): 1, (1, _S11): 1})
_set(_S21, _S22, {(1, _S21): 1})
_set(_S22, _S22, {(2, _E): 1})


def _mul24(p: Dict, x) -> Dict:
    out: Dict = {}
    for (d, b), c in p.items():
        for (d2, b2), c2 in _T24[(b, x)].items():
            out[(d + d2, b2)] = out.get((d + d2, b2), 0) + c * c2
    return out


for _x, _y, _z in itertools.product(_GR24, repeat=3):  # the table is a ring
    assert _mul24(_T24[(_x, _y)], _z) == _mul24(_T24[(_y, _z)], _x)


def _partition_of(I: Sequence[int], r: int) -> Tuple[int, ...]:
    return tuple(_N - r + j - i for j, i in enumerate(I, 1))


def _subset_of(part: Sequence[int], r: int) -> Tuple[int, ...]:
    return tuple(_N - r + j - p for j, p in enumerate(part, 1))


def _product_terms(r: int, subsets: Sequence[Tuple[int, ...]]) -> List[Tuple[int, Tuple[int, ...]]]:
    """``sigma_{I_1} * ... * sigma_{I_s}`` in QH*(Gr(r, 4)) as a list of (d, partition) with non-zero coefficient."""
    parts = [_partition_of(I, r) for I in subsets]
    if r == 2:
        p = {(0, parts[0]): 1}
        for x in parts[1:]:
            p = _mul24(p, x)
        return [key for key, c in p.items() if c]
    # Gr(1, 4) = P^3 (partitions (m,), m = 0..3) and Gr(3, 4) (partitions (1^m)): sigma_a sigma_b = q^{(a+b) div 4} sigma_{(a+b) mod 4}
    tot = sum(sum(pp) for pp in parts)
    d, m = divmod(tot, _N)
    return [(d, (m,) if r == 1 else tuple([1] * m + [0] * (3 - m)))]


def _dual(part: Tuple[int, ...], r: int) -> Tuple[int, ...]:
    return tuple(_N - r - p for p in reversed(part))


@lru_cache(maxsize=None)
def inequalities(s: int):
    """The inequalities of ``A_1 ... A_s = C`` in SU(4): ``(IA[s][M, 4], IC[M, 4], D[M])`` with 0/1 rows such that the condition is
    ``sum_l IA[l] @ alpha^(l) - IC @ gamma <= D`` (all M rows)."""
    if s < 2:
        raise ValueError("at least two factors")
    rows_a: List[List[np.ndarray]] = []
    rows_c: List[np.ndarray] = []
    ds: List[int] = []
    seen = set()
    for r in (1, 2, 3):
        subs = list(itertools.combinations(range(1, _N + 1), r))
        for tup in itertools.product(subs, repeat=s):
            for d, nu in _product_terms(r, tup):
                # <sigma_{I_1}, ..., sigma_{I_s}, sigma_K>_d != 0  <=>  q^d sigma_{K^dual} appears in the product
                K = _subset_of(_dual(nu, r), r)
                key = (tup, K, d)
                if key in seen:
                    continue
                seen.add(key)
                ra = []
                for I in tup:
                    v = np.zeros(_N)
                    v[[i - 1 for i in I]] = 1
                    ra.append(v)
                vc = np.zeros(_N)
                vc[[_N - k for k in K]] = 1  # gamma_{n + 1 - k}, zero-based index n - k
                rows_a.append(ra)
                rows_c.append(vc)
                ds.append(d)
    IA = [np.array([ra[l] for ra in rows_a]) for l in range(s)]
    return IA, np.array(rows_c), np.array(ds, dtype=np.float64)


# ---- coordinates -------------------------------------------------------------------------------------------------------------------
def _alcove_columns(coords, shift: float = 0.0):
    """``alcove_coordinates`` as four contiguous column vectors (the batch form: everything is an elementwise pass over [N])."""
    c = np.asarray(coords, dtype=np.float64).reshape(-1, 3)
    x, y, z = (np.ascontiguousarray(c[:, j]) for j in range(3))
    cols = [0.5 * (x + y - z) + shift, 0.5 * (x - y + z) + shift, 0.5 * (-x + y + z) + shift, 0.5 * (-x - y - z) + shift]
    c0, c1, c2, c3 = (v - np.floor(v) for v in cols)  # in [0, 1); their sum is an integer s in 0..3
    # decreasing order: a five-exchange sorting network on the columns
    c0, c1 = np.maximum(c0, c1), np.minimum(c0, c1)
    c2, c3 = np.maximum(c2, c3), np.minimum(c2, c3)
    c0, c2 = np.maximum(c0, c2), np.minimum(c0, c2)
    c1, c3 = np.maximum(c1, c3), np.minimum(c1, c3)
    c1, c2 = np.maximum(c1, c2), np.minimum(c1, c2)
    s = np.rint(c0 + c1 + c2 + c3).astype(np.int64)
    # subtract 1 from the s largest entries: they become the smallest, in the same order -- a rotation of the row by s
    e = [c0, c1, c2, c3, c0 - 1.0, c1 - 1.0, c2 - 1.0]
    return [np.choose(s, e[j : j + 4]) for j in range(_N)]


def alcove_coordinates(coords, shift: float = 0.0) -> np.ndarray:
    """Weyl coordinates (c1, c2, c3) [N, 3], units of pi  ->  alcove points [N, 4] (decreasing, sum 0, a_1 - a_4 <= 1) of
    ``i^{2 shift} CAN(c)``: ``shift`` 0 or 1/2 are the two points of the gate class."""
    return np.stack(_alcove_columns(coords, shift), axis=1)


_PATTERNS = [K for r in (1, 2, 3) for K in itertools.combinations(range(1, _N + 1), r)]  # the 14 subsets K, fixed order
_PATTERN_ROWS = np.zeros((len(_PATTERNS), _N))
for _p, _K in enumerate(_PATTERNS):
    _PATTERN_ROWS[_p, [_N - k for k in _K]] = 1  # gamma_{n + 1 - k}


def _ring_mul(r: int, part: Tuple[int, ...], x: Tuple[int, ...]):
    """Terms (d, partition) of ``sigma_part * sigma_x`` in QH*(Gr(r, 4)) (all structure constants are >= 0)."""
    if r == 2:
        return list(_T24[(part, x)])
    d, m = divmod(sum(part) + sum(x), _N)
    return [(d, (m,) if r == 1 else tuple([1] * m + [0] * (3 - m)))]


def region(gate_coords_list) -> np.ndarray:
    """``_region`` through a small cache (a basis sweep asks for the same few circuits at every step)."""
    g = np.asarray(gate_coords_list, dtype=np.float64).reshape(-1, 3)
    return _region_cached(tuple(np.round(g, 12).ravel().tolist())).copy()


@lru_cache(maxsize=4096)
def _region_cached(flat: Tuple[float, ...]) -> np.ndarray:
    return _region(np.array(flat).reshape(-1, 3))


def _region(gate_coords_list) -> np.ndarray:
    """The coverage region of a circuit of these gates as 14 half-spaces in the target's alcove coordinates ``gamma``: the target is
    inside iff ``bounds[p] <= _PATTERN_ROWS[p] @ gamma`` for all p.  ``bounds[p]`` = the largest ``sum_l sum_{I_l} alpha^(l) - d`` over
    the inequalities that end in subset K_p -- found by a dynamic programme over the gates (per r, the best value for every term
    ``q^d sigma_nu`` of the growing product; structure constants are non-negative, so a term of the full product is reached through
    terms of the partial ones), which is linear in the number of gates where the explicit list (``inequalities``) grows like 6^s."""
    ga = alcove_coordinates(gate_coords_list)
    bounds = np.full(len(_PATTERNS), -np.inf)
    for r in (1, 2, 3):
        subs = list(itertools.combinations(range(1, _N + 1), r))
        parts = [_partition_of(I, r) for I in subs]
        vals = [[float(sum(a[i - 1] for i in I)) for I in subs] for a in ga]
        state = {(0, parts[j]): vals[0][j] for j in range(len(subs))}
        for l in range(1, len(ga)):
            nxt: Dict = {}
            for (d, p), v in state.items():
                for j, x in enumerate(parts):
                    for d2, p2 in _ring_mul(r, p, x):
                        key = (d + d2, p2)
                        w = v + vals[l][j]
                        if w > nxt.get(key, -np.inf):
                            nxt[key] = w
            state = nxt
        for (d, nu), v in state.items():
            p = _PATTERNS.index(_subset_of(_dual(nu, r), r))
            bounds[p] = max(bounds[p], v - d)
    return bounds


def target_sums(target_coords):
    """The targets' side of every half-space, for both alcove points of each target class: per point ``(columns, sums)`` -- the four
    alcove coordinates and the 14 vectors ``_PATTERN_ROWS[p] @ gamma`` (plain column sums), each [N].  Computed once per batch and shared by the circuits it is tested against."""
    t = np.asarray(target_coords, dtype=np.float64).reshape(-1, 3)
    out = []
    for shift in (0.0, 0.5):
        cols = _alcove_columns(t, shift)
        c = cols
        sums = []
        for K in _PATTERNS:
            idx = [_N - k for k in K]
            v = cols[idx[0]]
            for j in idx[1:]:
                v = v + cols[j]
            sums.append(v)
        out.append((c, sums))
    return out


def contains(target_coords, gate_coords_list, tol: float = 1e-9, sums=None) -> np.ndarray:
    """bool[N]: target t is reachable (up to local gates) by a circuit of the gates with Weyl coordinates ``gate_coords_list`` -- in any
    order, the double cosets of a Gelfand pair commute -- with arbitrary local gates in between.  ``tol`` (alcove units = units of pi)
    widens (> 0) or shrinks (< 0) the region.  ``sums`` = ``target_sums(target_coords)`` when several circuits are tested."""
    g = np.asarray(gate_coords_list, dtype=np.float64).reshape(-1, 3)
    if len(g) < 1:
        raise ValueError("a circuit needs at least one gate")
    if sums is None:
        sums = target_sums(target_coords)
    n = len(sums[0][0][0])
    out = np.zeros(n, dtype=bool)
    if len(g) == 1:
        a = alcove_coordinates(g)[0]
        for c, _ in sums:
            ok = np.ones(n, dtype=bool)
            for j in range(_N):
                ok &= np.abs(c[j] - a[j]) <= max(tol, 0.0) + 1e-12
            out |= ok
        return out
    bounds = region(g)
    for _, cols in sums:  # the target class has two alcove points; the gates' are fixed by their CAN matrices
        ok = np.ones(n, dtype=bool)
        for p, v in enumerate(cols):
            if np.isfinite(bounds[p]):
                ok &= v >= bounds[p] - tol
        out |= ok
    return out


def minimal_prefix(target_coords, gate_coords_seq, k_max: int, tol: float = 1e-9) -> np.ndarray:
    """Smallest k such that the first k gates of the sequence reach the target (0 for local targets, ``k_max + 1`` if none does)."""
    t = np.asarray(target_coords, dtype=np.float64).reshape(-1, 3)
    g = np.asarray(gate_coords_seq, dtype=np.float64).reshape(-1, 3)
    if k_max > len(g):
        raise ValueError("gate sequence shorter than k_max")
    k_of = np.full(len(t), k_max + 1, dtype=np.int64)
    sums = target_sums(t)
    for k in range(k_max, 0, -1):
        k_of = np.where(contains(t, g[:k], tol, sums=sums), k, k_of)
    ident = np.zeros(len(t), dtype=bool)
    for c, _ in sums:
        ident |= (np.abs(c[0]) <= 1e-8) & (np.abs(c[3]) <= 1e-8)  # decreasing, sum 0: all four vanish
    return np.where(ident, 0, k_of)
