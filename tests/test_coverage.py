"""coverage.py (no GPU): the coverage sets of two-qubit circuits from the inequalities of the multiplicative eigenvalue problem for
SU(4) -- what the reference gets from the monodromy package (src/slam/utils/polytopes/polytope_wrap.py:39-196).

  * every inequality holds, and the families are attained, on random products in SU(4) and on random CAN . L . CAN (. L . CAN) circuits;
  * the dynamic programme (``region``) equals the explicit inequality list (``inequalities``);
  * the regions reproduce the closed-form rules of span_rules (single-gate classes, iSWAP . L . B, XY-type pairs) with no mismatch;
  * known facts: three CX / sqrt(iSWAP) / iSWAP reach everything, two B do, SWAP needs three CX, [CX, CX, sqrt(iSWAP)] is NOT universal;
  * circuits sampled with random local gates fill the predicted region.
"""
import numpy as np
import pytest
from scipy.stats import unitary_group

from oracle import slam_oracle as o
from slam_decomposition_amd import coverage as cov
from slam_decomposition_amd import span_rules
from slam_decomposition_amd.weyl import c1c2c3


def _su(n, count, rng):
    u = unitary_group.rvs(n, size=count, random_state=rng)
    return u * np.exp(-1j * np.angle(np.linalg.det(u)) / n)[:, None, None]


def _locals(count, rng):
    return np.einsum("nij,nkl->nikjl", _su(2, count, rng), _su(2, count, rng)).reshape(count, 4, 4)


def _alcove(a):
    a = np.mod(a, 1.0)
    a = -np.sort(-a, axis=-1)
    s = np.rint(a.sum(-1)).astype(int)
    a = a - (np.arange(4)[None, :] < s[:, None])
    return -np.sort(-a, axis=-1)


_SYSY = np.kron(np.array([[0, -1j], [1j, 0]]), np.array([[0, -1j], [1j, 0]]))


def _logspec_gamma(U):
    """Alcove point of U in SU(4) as a two-qubit gate: spectrum of U (sy sy) U^T (sy sy) (= U U^T in the magic basis)."""
    Ut = _SYSY @ np.swapaxes(U, -1, -2) @ _SYSY
    return _alcove(np.angle(np.linalg.eigvals(U @ Ut)) / (2 * np.pi))


def _chamber(count, rng):
    t = rng.uniform(0, 1, (count, 3))
    t[:, 1] *= 0.5
    t[:, 2] *= 0.5
    return t[(t[:, 1] <= np.minimum(t[:, 0], 1 - t[:, 0])) & (t[:, 2] <= t[:, 1])]


def test_inequalities_hold_on_random_products_in_su4():
    rng = np.random.default_rng(1)
    n = 40000
    A, B = _su(4, n, rng), _su(4, n, rng)
    spec = lambda U: _alcove(np.angle(np.linalg.eigvals(U)) / (2 * np.pi))  # noqa: E731
    a, b, c = spec(A), spec(B), spec(A @ B)
    IA, IC, D = cov.inequalities(2)
    assert len(D) == 72
    v = a @ IA[0].T + b @ IA[1].T - c @ IC.T - D[None, :]
    assert v.max() < 1e-9  # never violated ...
    assert v.max(axis=0).min() > -0.15 and v.max() > -0.03  # ... and none of them is slack everywhere


def test_inequalities_hold_on_two_qubit_circuits_with_random_local_gates():
    rng = np.random.default_rng(2)
    n = 30000
    cs = [rng.uniform(-1, 1, (n, 3)) for _ in range(3)]
    G = [np.array([o.canonical_matrix(*c) for c in cc]) for cc in cs]
    al = [_logspec_gamma(g) for g in G]
    for l in range(3):  # the closed form of the alcove point
        assert np.abs(cov.alcove_coordinates(cs[l]) - al[l]).max() < 1e-12
    U2 = G[1] @ _locals(n, rng) @ G[0]
    U3 = G[2] @ _locals(n, rng) @ U2
    for s, U in ((2, U2), (3, U3)):
        IA, IC, D = cov.inequalities(s)
        v = sum(al[l] @ IA[l].T for l in range(s)) - _logspec_gamma(U) @ IC.T - D[None, :]
        assert v.max() < 1e-9, s
        assert v.max() > -0.02, s


def test_region_equals_the_explicit_inequality_list():
    rng = np.random.default_rng(3)
    t = _chamber(20000, rng)
    for s in (2, 3, 4):
        IA, IC, D = cov.inequalities(s)
        for _ in range(4):
            g = rng.uniform(0, 0.5, (s, 3)) * np.array([1.0, rng.uniform(0, 1), 0.3])
            ga = cov.alcove_coordinates(g)
            base = sum(IA[l] @ ga[l] for l in range(s)) - D
            ref = np.zeros(len(t), bool)
            for sh in (0.0, 0.5):
                ref |= np.all(base[None, :] - cov.alcove_coordinates(t, sh) @ IC.T <= 1e-9, axis=1)
            assert np.array_equal(ref, cov.contains(t, g))
    assert len(cov.inequalities(3)[2]) == 392


def test_regions_reproduce_the_closed_form_rules():
    rng = np.random.default_rng(4)
    t = _chamber(150000, rng)
    f = span_rules._fold(t)
    x, y, z = f[:, 0], f[:, 1], f[:, 2]
    cx, isw, sq, b = (0.5, 0, 0), (0.5, 0.5, 0), (0.25, 0.25, 0), (0.5, 0.25, 0)
    closed = {  # the closed forms, written out here independently of span_rules.two_gate_region
        (sq, sq): lambda tol: np.abs(z) <= x - y + tol,
        (isw, b): lambda tol: (x >= 0.25 - tol) & (np.abs(z) <= 0.25 + tol),
        (b, b): lambda tol: np.ones(len(x), bool),
        ((0.15, 0.15, 0), (0.15, 0.15, 0)): lambda tol: (np.abs(z) <= x - y + tol) & (x + y + np.abs(z) <= 0.6 + tol) & (x <= 0.3 + tol),
    }
    for (g1, g2), rule in closed.items():
        pred = cov.contains(t, [g1, g2])
        clear = rule(1e-7) == rule(-1e-7)
        assert np.array_equal(pred[clear], rule(0.0)[clear]), (g1, g2)
        assert 0 < pred.mean()
    # CX . L . CX and iSWAP . L . iSWAP: the c3 = 0 face (measure zero: no random target inside, every face point inside)
    face = t.copy()
    face[:, 2] = 0.0
    for g in (cx, isw):
        assert not cov.contains(t[t[:, 2] > 1e-6], [g, g]).any() and cov.contains(face, [g, g]).all()
    # three equal gates of the classes CX, iSWAP, sqrt(iSWAP), [iSWAP, B, iSWAP]: everything; single gates: their own class only
    for g3 in ([cx] * 3, [isw] * 3, [sq] * 3, [isw, b, isw], [b, b]):
        assert cov.contains(t, g3).all()
    assert cov.contains([sq, (0.75, 0.25, 0), cx], [sq]).tolist() == [True, True, False]
    swap = [(0.5, 0.5, 0.5)]
    assert not cov.contains(swap, [cx, cx])[0] and cov.contains(swap, [cx] * 3)[0] and not cov.contains(swap, [sq, isw])[0]
    assert not cov.contains(swap, [cx, cx, sq])[0] and 0.8 < cov.contains(t, [cx, cx, sq]).mean() < 0.95  # NOT universal
    # minimal_prefix = span_rules.minimal_span for the four classes
    for g in (cx, isw, sq, b):
        want = span_rules.minimal_span(t, g)
        got = cov.minimal_prefix(t, [g] * 3, 3, tol=2e-8)
        edge = np.abs(np.abs(z) - (x - y)) < 1e-6 if g == sq else np.abs(z) < 1e-6
        assert np.array_equal(got[~edge], want[~edge]), g
    assert cov.minimal_prefix([(0, 0, 0), (1, 0, 0)], [cx] * 3, 3).tolist() == [0, 0]


@pytest.mark.parametrize("gates", [[(0.5, 0, 0), (0.5, 0, 0), (0.25, 0.25, 0)], [(0.3, 0.1, 0), (0.3, 0.1, 0)], [(0.2, 0.1, 0)] * 3,
                                   [(0.45, 0.2, 0.1), (0.15, 0.1, 0.05)]])
def test_sampled_circuits_lie_inside_and_fill_the_region(gates):
    """Necessity and (statistically) sufficiency: 40 000 circuits g_k L ... L g_1 with Haar-random local gates all land inside the
    predicted region, and no 0.05 cell lying 0.03 inside the region stays empty except a few at low-density corners."""
    rng = np.random.default_rng(5)
    n = 40000
    U = np.broadcast_to(o.canonical_matrix(*gates[0]), (n, 4, 4))
    for g in gates[1:]:
        U = o.canonical_matrix(*g)[None] @ _locals(n, rng) @ U
    cs = np.array([c1c2c3(u) for u in U])
    assert cov.contains(cs, gates, tol=1e-7).all()
    f = span_rules._fold(cs)
    h = 0.05
    pts = np.array([[xx, yy, zz] for xx in np.arange(h / 2, 0.5, h) for yy in np.arange(h / 2, 0.5, h) for zz in np.arange(-0.5 + h / 2, 0.5, h)
                    if yy <= xx and abs(zz) <= yy])
    inside = cov.contains(span_rules._unfold(pts), gates, tol=-0.03)
    occupied = set(map(tuple, np.floor(np.stack([f[:, 0] / h, f[:, 1] / h, (f[:, 2] + 0.5) / h], 1)).astype(int)))
    cells = np.floor(np.stack([pts[:, 0] / h, pts[:, 1] / h, (pts[:, 2] + 0.5) / h], 1)).astype(int)
    empty = [tuple(c) for c, i in zip(cells, inside) if i and tuple(c) not in occupied]
    assert inside.sum() > 20 and len(empty) <= 0.06 * inside.sum(), (inside.sum(), len(empty))


def test_regions_equal_the_coverage_sets_the_reference_ships():
    """The pin: tests/golden/reference_coverage_polytopes.json holds the coverage sets the reference ships as data
    (src/slam/data/polytopes/polytope_coverage_[...].pkl -- monodromy's output, precomputed by the reference's authors for 16 gain-only
    ConversionGainGates and one with both drives, circuits of up to 26 gates; extracted by tools/make_reference_coverage_fixture.py, which
    reads the numbers without importing monodromy or the reference).  A target's membership in every one of the 116 entries, evaluated
    from the reference's inequality rows in its monodromy coordinates, must equal ``coverage.contains`` for the same gate list -- for
    every target that is not within 1e-6 of a face."""
    import json
    import os

    from slam_decomposition_amd.gates import ConversionGainGate

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_coverage_polytopes.json")
    ref = json.load(open(path))
    assert len(ref) == 17
    rng = np.random.default_rng(0)
    t = _chamber(30000, rng)
    # the reference's monodromy coordinates of a target in the chamber (c3 >= 0) are the first three alcove coordinates
    mono = cov.alcove_coordinates(t)[:, :3]
    num = lambda x: x[0] / x[1] if isinstance(x, list) else x  # noqa: E731  (Fractions are stored as [numerator, denominator])

    def ref_inside(entry, tol):
        out = np.zeros(len(mono), bool)
        for cp in entry["convex_subpolytopes"]:
            ok = np.ones(len(mono), bool)
            for row in cp["inequalities"]:
                r = [num(x) for x in row]
                ok &= r[0] + mono @ np.array(r[1:]) >= -tol
            for row in cp["equalities"]:
                r = [num(x) for x in row]
                ok &= np.abs(r[0] + mono @ np.array(r[1:])) <= 1e-9
            out |= ok
        return out

    checked = 0
    for name, v in ref.items():
        gc, gg, dur = v["gates"][0]
        gate = ConversionGainGate(0, 0, gc, gg, dur)
        g = c1c2c3(gate.to_matrix())
        assert v["gate_keys"] == [str(gate)]  # the reference's gate key (custom_gates.py:185-191) is ours
        for e in v["coverage"]:
            k = len(e["operations"])
            assert e["operations"] == [str(gate)] * k
            assert abs(e["cost"] - k * gate.cost()) < 1e-7 * max(k, 1)  # cost = sum of gate.cost() (polytope_wrap.py:175-176; 8-digit gate keys)
            if k == 0:
                continue  # the identity polytope (skipped by the reference's lookup too, polytope_wrap.py:82-84)
            if k == 1:
                # the gate's own class: both alcove points, as equalities
                pts = {tuple(np.round(cov.alcove_coordinates(g, sh)[0][:3], 7)) for sh in (0.0, 0.5)}
                got = set()
                for cp in e["convex_subpolytopes"]:
                    A = np.array([[num(x) for x in row[1:]] for row in cp["equalities"]])
                    b = -np.array([num(row[0]) for row in cp["equalities"]])
                    got.add(tuple(np.round(np.linalg.solve(A, b), 7)))
                assert got <= pts and len(got) >= 1, (name, got, pts)
                continue
            mine = cov.contains(t, [g] * k, tol=0.0)
            clear = cov.contains(t, [g] * k, tol=1e-6) == cov.contains(t, [g] * k, tol=-1e-6)
            theirs = ref_inside(e, 0.0)
            assert np.array_equal(mine[clear], theirs[clear]), (name, k, int((mine != theirs)[clear].sum()))
            checked += 1
    assert checked == 99


def test_haar_volumes_of_the_regions_equal_the_volumes_the_reference_recorded():
    """A second pin, on the measure: src/slam/data/extended_results.json holds the Haar volume monodromy computed for the coverage set of
    k applications of six ConversionGainGates (parallel_drive_volume.py:340-348: sqrt(iSWAP) x 2 = 0.79012, sqrt(B) x 3 = 0.99581,
    sqrt(CNOT) x 4 = 0.95988, x 5 = 0.999863, ...; fixture tests/golden/reference_haar_volumes.json).  Here: 300 000 Haar-random
    unitaries (SciPy), the fraction ``coverage.contains`` puts inside each region -- within 4 standard errors of the recorded volume."""
    import json
    import os

    from slam_decomposition_amd.gates import ConversionGainGate

    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_haar_volumes.json")))
    rng = np.random.default_rng(11)
    n = 300000
    a = _logspec_gamma(_su(4, n, rng))  # alcove points of Haar-random gates
    coords = np.stack([a[:, 0] + a[:, 1], a[:, 0] + a[:, 2], a[:, 1] + a[:, 2]], axis=1)  # a canonical triple of the same class
    sums = cov.target_sums(coords)
    checked = 0
    for name, v in ref.items():
        g = c1c2c3(ConversionGainGate(0, 0, v["gc"], v["gg"], v["t"]).to_matrix())
        for k, vol in v["base_vol"].items():
            k = int(k)
            if k == 1:
                assert vol == 0.0  # (a single gate's class: measure zero)
                continue
            frac = float(cov.contains(None, [g] * k, tol=0.0, sums=sums).mean())
            se = max(np.sqrt(max(vol * (1 - vol), 1e-9) / n), 1e-5)
            assert abs(frac - vol) <= 4 * se + 2e-5, (name, k, frac, vol)
            checked += 1
    assert checked == 15


def test_candidate_gate_scores_from_the_coverage_regions():
    """tools/candidates.py (a dev tool, not in the product package) -- the sweep BASELINE configs[4] is shaped like (bare_candidates.py:47-126: every candidate gate gets a Haar score
    and the sizes at which CNOT and SWAP are reached, from its coverage set).  For the six gates whose volumes the reference recorded
    (extended_results.json): full coverage at the size the reference's table of those gates gives (parallel_drive_volume.py:91-96:
    iSwap 3, sqiSwap 3, CNOT 3, sqCNOT 6, B 2, sqB 4), Haar score = sum_k k (vol_k - vol_{k-1}) of the RECORDED volumes within the
    sample's error, CNOT / SWAP sizes; and the candidate grid itself."""
    import json
    import os

    import importlib.util

    spec = importlib.util.spec_from_file_location("slam_tools_candidates", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "candidates.py"))
    cd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cd)
    from slam_decomposition_amd.gates import ConversionGainGate

    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_haar_volumes.json")))
    names = list(ref)
    res = cd.score_gates([ConversionGainGate(0, 0, ref[n]["gc"], ref[n]["gg"], ref[n]["t"]) for n in names], n_samples=120000, seed=3)
    want_cnot_swap = {"iSwap": (2, 3), "sqiSwap": (2, 3), "CNOT": (1, 3), "sqCNOT": (2, 6), "B": (2, 2), "sqB": (2, 4)}
    for n, r in zip(names, res):
        vols = {int(k): v for k, v in ref[n]["base_vol"].items()}
        assert max(r["volumes"]) == max(vols), (n, r["volumes"])  # full coverage at the reference's size
        score = sum(k * (vols[k] - vols.get(k - 1, 0.0)) for k in sorted(vols))
        assert abs(r["haar_score"] - score) < 0.01, (n, r["haar_score"], score)
        assert (r["cnot_score"], r["swap_score"]) == want_cnot_swap[n], (n, r["cnot_score"], r["swap_score"])
    gates, coords = cd.build_gates()
    assert len(gates) == sum(len(c) for c in coords) and len(coords) == 17 and coords[0] == [[0.0, 0.0, 0.0]]
    assert cd.gate_scores(gates[0], n_samples=1000)["haar_score"] is None  # the identity reaches nothing
    # a weak candidate (gain-only, pi/32): many applications -- CNOT at 16 (16 x 1/32 = 1/2), SWAP later, the Haar score below both
    weak = cd.gate_scores(gates[1], n_samples=20000)
    assert weak["cnot_score"] == 16 and weak["haar_score"] < weak["cnot_score"] < weak["swap_score"] and weak["volumes"][max(weak["volumes"])] == 1.0
