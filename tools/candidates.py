"""Dev tool (CPU; NOT part of the product package -- SURVEY.md section 2 rows 10 / 13 are out of scope; kept as a consumer of
``coverage.py`` that checks it against the reference's recorded Haar volumes): scoring candidate basis gates by what their circuits can
reach -- the sweep BASELINE configs[4] is shaped like.

The reference builds a grid of ``ConversionGainGate`` candidates (``build_gates``, src/slam/utils/gates/bare_candidates.py:47-69) and gives
every one three scores from its monodromy coverage set (``collect_data``, bare_candidates.py:75-126): the Haar expectation of the number
of gates a target needs (``coverage_to_haar_expectation``, src/slam/utils/polytopes/polytope_wrap.py:206-215 -> monodromy's
``expected_cost``: sum over k of k times the volume first covered with k gates) and the sizes at which CNOT and SWAP are first reached
(``monodromy_range_from_target``, polytope_wrap.py:39-94).  With ``coverage.py`` the regions of ANY number of applications of a gate are
14 half-spaces each, so the three scores are a few comparisons per (gate, k) -- no optimisation, no polytope library:

* ``build_gates``  the reference's candidate grid, duplicates of a Weyl class removed the same way;
* ``gate_scores``  (haar_score, cnot_score, swap_score, volumes) of one gate: volumes over a common set of Haar targets (their
  half-space sums are computed once and shared by all gates and all k), exact membership for the two named targets;
* ``score_gates``  the sweep.

Volumes are Monte-Carlo estimates over Haar-random targets (standard error sqrt(v (1 - v) / n)); tests/test_coverage.py checks them and
the resulting Haar scores against the volumes the reference recorded (src/slam/data/extended_results.json).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_decomposition_amd import coverage  # noqa: E402
from slam_decomposition_amd.gates import ConversionGainGate, gate_matrix  # noqa: E402
from slam_decomposition_amd.weyl import c1c2c3  # noqa: E402

CNOT_COORDS = (0.5, 0.0, 0.0)
SWAP_COORDS = (0.5, 0.5, 0.5)


def build_gates(elim_extra_weyl: bool = True) -> Tuple[List[ConversionGainGate], List[list]]:
    """bare_candidates.py:47-69: ``ConversionGainGate(0, 0, p k pi, (1 - p) k pi)`` for 17 strengths k in [0, 1/2] x 21 splits p in
    [0, 1], skipping a gate whose Weyl coordinates (c1 folded to <= 1/2) were seen before.  Returns (gates, coordinates per strength)."""
    unitary_list: List[ConversionGainGate] = []
    coordinate_list: List[list] = []
    for k in np.linspace(0, 0.5, 17):
        inner_list: list = []
        for p in np.linspace(0, 1, 21):
            gate = ConversionGainGate(0, 0, p * k * np.pi, (1 - p) * k * np.pi)
            c = list(c1c2c3(gate.to_matrix()))
            if elim_extra_weyl and c[0] > 0.5:
                c[0] = -1 * c[0] + 1
            if c in inner_list or any(c in inner for inner in coordinate_list):
                continue
            inner_list.append(c)
            unitary_list.append(gate)
        coordinate_list.append(inner_list)
    return unitary_list, coordinate_list


def haar_targets(n: int, seed: int = 0) -> np.ndarray:
    """Weyl coordinates (a canonical triple per class, not folded) of n Haar-random two-qubit gates, from SciPy's sampler."""
    from scipy.stats import unitary_group

    u = unitary_group.rvs(4, size=n, random_state=np.random.default_rng(seed))
    u = u * np.exp(-1j * np.angle(np.linalg.det(u)) / 4)[:, None, None]
    sy = np.array([[0, -1j], [1j, 0]])
    sysy = np.kron(sy, sy)
    ev = np.linalg.eigvals(u @ (sysy @ np.swapaxes(u, -1, -2) @ sysy))
    a = np.angle(ev) / (2 * np.pi)
    a = a - np.floor(a)
    a = -np.sort(-a, axis=1)
    s = np.rint(a.sum(axis=1)).astype(np.int64)
    a = a - (np.arange(4)[None, :] < s[:, None])
    a = -np.sort(-a, axis=1)
    return np.stack([a[:, 0] + a[:, 1], a[:, 0] + a[:, 2], a[:, 1] + a[:, 2]], axis=1)


def first_size(target_coords, gate_coords, k_cap: int = 64) -> Optional[int]:
    """Number of applications of the gate at which the target is first reached (``monodromy_range_from_target(...)[0]``), None beyond
    ``k_cap``; 0 for a local target."""
    t = np.asarray(target_coords, dtype=np.float64).reshape(1, 3)
    sums = coverage.target_sums(t)
    if any(abs(c[0][0]) <= 1e-8 and abs(c[3][0]) <= 1e-8 for c, _ in sums):
        return 0
    g = np.asarray(gate_coords, dtype=np.float64).reshape(1, 3)
    for k in range(1, k_cap + 1):
        if coverage.contains(None, np.repeat(g, k, axis=0), tol=2e-8, sums=sums)[0]:
            return k
    return None


def gate_scores(gate, sums=None, n_samples: int = 100000, seed: int = 0, k_cap: int = 64) -> Dict:
    """Scores of one candidate gate (an object with ``to_matrix()`` / ``__array__`` or a 4x4 array):

    ``volumes[k]``  Haar volume reached with k applications, k = 1 .. the first k that reaches everything (or ``k_cap``);
    ``haar_score``  sum_k k x (volume first reached with k applications) -- the reference's expected cost with unit gate cost; None if
                    the coverage is still incomplete at ``k_cap`` (a local gate reaches nothing);
    ``cnot_score``, ``swap_score``  first k reaching CNOT / SWAP (None beyond ``k_cap``).

    ``sums`` = ``coverage.target_sums(haar_targets(n))`` shares the Haar sample between gates."""
    g = np.asarray(c1c2c3(gate_matrix(gate)), dtype=np.float64)
    if sums is None:
        sums = coverage.target_sums(haar_targets(n_samples, seed))
    n = len(sums[0][0][0])
    volumes: Dict[int, float] = {}
    covered = np.zeros(n, dtype=bool)
    score: Optional[float] = 0.0
    local = bool(np.max(np.abs(coverage.alcove_coordinates(g))) < 1e-8 or np.max(np.abs(coverage.alcove_coordinates(g, 0.5))) < 1e-8)
    k = 0
    while not local:
        k += 1
        inside = coverage.contains(None, np.repeat(g[None], k, axis=0), tol=0.0, sums=sums)
        volumes[k] = float(inside.sum()) / n
        first = inside & ~covered  # the targets that k applications reach and fewer do not
        score += k * float(first.sum()) / n
        covered |= inside
        if covered.all() or k >= k_cap:
            break
    if local or not covered.all():
        score = None
    return {"coords": tuple(float(x) for x in g), "volumes": volumes, "haar_score": score,
            "cnot_score": first_size(CNOT_COORDS, g, k_cap), "swap_score": first_size(SWAP_COORDS, g, k_cap)}


def score_gates(gates: Sequence, n_samples: int = 100000, seed: int = 0, k_cap: int = 64) -> List[Dict]:
    """``collect_data`` (bare_candidates.py:75-126) for a list of candidate gates: one Haar sample for all of them."""
    sums = coverage.target_sums(haar_targets(n_samples, seed))
    return [gate_scores(g, sums=sums, k_cap=k_cap) for g in gates]
