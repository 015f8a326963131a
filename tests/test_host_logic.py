"""CPU: host-side mirror of the reference interface (gates, sampler, template structure, Weyl
coordinates, sharding) checked against the oracle.  No GPU calls."""
import numpy as np
import pytest

from oracle import slam_oracle as o
from slam_decomposition_amd import gates as G
from slam_decomposition_amd.gates import CXGate, RiSwapGate
from slam_decomposition_amd.basis import CircuitTemplate
from slam_decomposition_amd.basis_abc import DataDictEntry
from slam_decomposition_amd.cost_function import BasicCost
from slam_decomposition_amd.parallel import LocalComm, merge_results, shard_range
from slam_decomposition_amd.sampler import GateSample, HaarBatch, HaarSample
from slam_decomposition_amd.weyl import c1c2c3


def test_gate_matrices_match_oracle():
    assert np.array_equal(G.CXGate().to_matrix(), o.cx_matrix())
    for a in (0.25, 0.5, 1.0):
        assert np.allclose(G.RiSwapGate(a).to_matrix(), o.riswap_matrix(a), atol=0, rtol=0)
    assert np.allclose(G.BerkeleyGate().to_matrix(), o.berkeley_matrix(), atol=1e-16)
    assert np.allclose(G.CanonicalGate(0.3, 0.2, 0.1).to_matrix(), o.canonical_matrix(0.6 / np.pi, 0.4 / np.pi, 0.2 / np.pi), atol=1e-15)
    p = (0.3, -0.7, 0.9, 0.4, 1.3)
    assert np.allclose(G.ConversionGainGate(*p).to_matrix(), o.conversion_gain_matrix_expm(*p), atol=5e-15)
    assert np.asarray(G.RiSwapGate(0.5)).shape == (4, 4)  # __array__ protocol like the reference gates
    assert G.RiSwapGate(0.5).cost() == 0.5 and G.RiSwapGate(0.5).duration == 0.5
    assert str(G.ConversionGainGate(0, 0, np.pi / 2, 0, 1)) == "2QGate(1.57079633, 0.00000000, 1.00000000)"
    with pytest.raises(ValueError):
        G.gate_matrix(np.eye(3))


def test_weyl_coordinates_match_oracle():
    for s in range(20):
        U = o.haar_unitary(s)
        assert c1c2c3(U) == o.c1c2c3(U)
    assert c1c2c3(G.SwapGate().to_matrix()) == (0.5, 0.5, 0.5)
    from slam_decomposition_amd.weyl import c1c2c3_batch

    U = np.stack([o.haar_unitary(s) for s in range(64)] + [o.cx_matrix(), o.riswap_matrix(1.0), np.eye(4, dtype=complex)])
    assert np.array_equal(c1c2c3_batch(U), np.array([c1c2c3(u) for u in U]))
    assert c1c2c3_batch(np.zeros((0, 4, 4))).shape == (0, 3)


def test_samplers():
    a = list(HaarSample(seed=3, n_samples=2))
    ref = o.haar_sample_reference(3, 2)
    assert np.array_equal(a[0], ref[0]) and np.array_equal(a[1], ref[1]) and np.array_equal(a[0], a[1])
    b = HaarBatch(seed0=o.BENCH_TARGET_SEED0, n_samples=3).as_array()
    assert np.array_equal(b, o.haar_batch(3))
    (g,) = list(GateSample(G.CXGate()))
    assert np.array_equal(g, o.cx_matrix())
    assert len(list(HaarSample(seed=None, n_samples=0))) == 0


def test_basic_cost_matches_oracle():
    U, V = o.haar_unitary(1), o.haar_unitary(2)
    c = BasicCost()
    assert c.normalization == 1
    assert c.unitary_fidelity(U, V) == pytest.approx(o.basic_cost(U, V), abs=1e-16)
    assert c.unitary_fidelity(np.exp(0.3j) * V, V) < 1e-15  # global-phase invariant
    from slam_decomposition_amd.cost_function import SquareCost

    assert SquareCost().unitary_fidelity(U, V) == pytest.approx(o.square_cost(U, V), abs=1e-16)
    x = np.linspace(0.1, 2.0, 18)
    gs = [o.riswap_matrix(0.5)] * 2
    v, g = o.square_loss_and_grad(x, gs, V)
    assert abs(v - o.square_cost(o.template_eval(x, gs), V)) < 1e-15
    h = 1e-6
    fd = np.array([(o.square_loss_and_grad(x + h * e, gs, V)[0] - o.square_loss_and_grad(x - h * e, gs, V)[0]) / (2 * h) for e in np.eye(18)])
    assert np.max(np.abs(fd - g)) < 2e-9


def test_circuit_template_structure():
    t = CircuitTemplate(base_gates=[G.RiSwapGate(1.0), G.BerkeleyGate()], maximum_span_guess=3)
    assert t.n_qubits == 2 and list(t.spanning_range) == [1, 2, 3]
    assert (t.using_bounds, t.bounds_list, t.using_constraints, t.constraint_func, t.preseeded) == (False, None, False, None, False)
    with pytest.raises(ValueError):
        t.build(0)
    t.build(3)
    assert t.cycles == 3 and t.n_params == 24
    assert t.gate_sequence() == [0, 1, 0]  # cycle restarts at every build (documented deviation C-2)
    x = t.parameter_guess()
    assert x.shape == (24,) and np.all((x >= 0) & (x < 2 * np.pi))
    gl = t.to_gate_list(np.arange(24.0))
    assert [g[0] for g in gl] == ["u", "u", "gate", "u", "u", "gate", "u", "u", "gate", "u", "u"]
    assert gl[0] == ("u", 0, (0.0, 1.0, 2.0)) and gl[1] == ("u", 1, (3.0, 4.0, 5.0)) and gl[3] == ("u", 0, (6.0, 7.0, 8.0))
    assert t.target_invariant(o.cx_matrix()) == (0.5, 0.0, 0.0)
    assert t.target_invariant(np.eye(8)) == (-1, -1, -1, -1)
    assert CircuitTemplate(no_exterior_1q=True).no_exterior_1q is True  # (round 5: implemented, SLAM_FLAG_NO_EXTERIOR)
    for kwargs in (dict(n_qubits=3), dict(edge_params=[[(1, 0)]])):
        with pytest.raises(NotImplementedError):
            CircuitTemplate(**kwargs)


def test_span_rules_and_polytope_mode_ranges():
    """use_polytopes=True: analytic stand-in for monodromy_range_from_target (polytope_wrap.py:39-94)."""
    from slam_decomposition_amd import span_rules

    cx, sq, b = (0.5, 0.0, 0.0), (0.25, 0.25, 0.0), (0.5, 0.25, 0.0)
    coords = np.array([(0, 0, 0), (1.0, 0, 0), cx, sq, b, (0.37, 0.11, 0.0), (0.4, 0.3, 0.2), (0.7, 0.2, 0.05), (0.3, 0.1, 0.1),
                       (0.5, 0.5, 0.5)])
    assert span_rules.minimal_span(coords, cx).tolist() == [0, 0, 1, 2, 2, 2, 3, 3, 3, 3]
    # sqrt(iSWAP): 2 iff |z| <= x - y in the folded chamber (weyl_decompose.py:348); (0.7, 0.2, 0.05) folds to (0.3, 0.2, -0.05)
    assert span_rules.minimal_span(coords, sq).tolist() == [0, 0, 2, 1, 2, 2, 3, 2, 2, 3]
    assert span_rules.minimal_span(coords, b).tolist() == [0, 0, 2, 2, 1, 2, 2, 2, 2, 2]
    assert span_rules.minimal_span(coords, (0.5, 0.5, 0.0)).tolist() == [0, 0, 2, 2, 2, 2, 3, 3, 3, 3]  # iSWAP
    assert span_rules.family_of((0.75, 0.25, 0.0)) == "sqiswap"  # mirror image of (0.25, 0.25, 0)
    t = CircuitTemplate(base_gates=[RiSwapGate(0.5)], use_polytopes=True)
    assert t.spanning_range is None and t.use_polytopes  # basis.py:82-86
    assert t.get_spanning_range(o.riswap_matrix(0.5)) == range(1, 2)
    assert t.get_spanning_range(o.cx_matrix()) == range(2, 3)
    assert t.get_spanning_range(np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=complex)) == range(3, 4)
    assert t.get_spanning_range(np.eye(4)) == range(0, 1)  # polytope_wrap.py:55-56


def test_span_lower_bounds_for_mixed_sequences_and_other_gates():
    """use_polytopes=True without closed-form coverage regions -- a sequence of different gates (MixedOrderBasisCircuitTemplate
    territory, basis.py:213-359) or a gate outside the known classes: a sound LOWER bound on the template size from the
    interaction-strength measures (span_rules.span_lower_bound); the span loop starts there."""
    from slam_decomposition_amd import span_rules
    from slam_decomposition_amd.gates import BerkeleyGate

    isw, b, cx = (0.5, 0.5, 0.0), (0.5, 0.25, 0.0), (0.5, 0.0, 0.0)
    coords = np.array([(0, 0, 0), isw, b, (0.4, 0.3, 0.2), (0.5, 0.5, 0.5), (0.7, 0.2, 0.05)])
    # [iSWAP, B, iSWAP]: local 0, the FIRST gate's class 1; round 4: at k = 2 the pair's exact coverage region (x >= 1/4, |z| <= 1/4)
    # stands in for the strength test -- SWAP (|z| = 1/2) needs the third gate
    assert span_rules.span_lower_bound(coords, [isw, b, isw]).tolist() == [0, 1, 2, 2, 3, 2]
    # the bound never exceeds the exact rules of the single-gate families
    rng = np.random.default_rng(0)
    c = np.sort(rng.uniform(0, 0.5, (2000, 3)), axis=1)[:, ::-1]
    for g in (cx, isw, (0.25, 0.25, 0.0), b):
        assert np.all(span_rules.span_lower_bound(c, [g] * 3) <= span_rules.minimal_span(c, g))
    # a weak gate: three applications offer m1 = 3 (x + y) of interaction strength; SWAP (m1 = 1.5) is out of reach
    weak = (0.1, 0.1, 0.0)
    lb = span_rules.span_lower_bound(np.array([(0.05, 0.05, 0.0), (0.15, 0.1, 0.05), (0.3, 0.2, 0.1), (0.5, 0.5, 0.5)]), [weak] * 3)
    assert lb.tolist() == [2, 2, 3, 4]  # 4 = not reachable with the whole sequence
    assert np.allclose(span_rules.strength(np.array([(0.7, 0.2, 0.05)])), [[0.55, 0.3]])  # folds to (0.3, 0.2, -0.05)
    # the template: exact for one known gate and (round 4) for the [iSWAP, B] sequence; lower bound + brute force from there otherwise
    t = CircuitTemplate(base_gates=[RiSwapGate(1.0), BerkeleyGate()], use_polytopes=True, maximum_span_guess=3)
    assert t.span_rules_exact and CircuitTemplate(base_gates=[RiSwapGate(0.5)], use_polytopes=True).span_rules_exact
    assert t.get_spanning_range(o.riswap_matrix(1.0)) == range(1, 2)       # iSWAP's own class: one gate
    assert t.get_spanning_range(o.cx_matrix()) == range(2, 3)               # (1/2, 0, 0): inside the pair's region
    assert t.get_spanning_range(np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=complex)) == range(3, 4)  # SWAP
    assert t.get_spanning_range(np.eye(4)) == range(0, 1)
    # (later in round 4: coverage.py makes every sequence exact -- templates of five gates, gates without a closed form)
    t5 = CircuitTemplate(base_gates=[RiSwapGate(1.0), BerkeleyGate()], use_polytopes=True, maximum_span_guess=5)
    assert t5.span_rules_exact and t5.get_spanning_range(o.cx_matrix()) == range(2, 3)
    w = CircuitTemplate(base_gates=[RiSwapGate(0.2)], use_polytopes=True, maximum_span_guess=3)  # (0.1, 0.1, 0): no closed form
    assert w.span_rules_exact
    swap = np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=complex)
    with pytest.raises(ValueError, match="did not find a polytope"):          # polytope_wrap.py:91-93
        w.get_spanning_range(swap)
    w8 = CircuitTemplate(base_gates=[RiSwapGate(0.2)], use_polytopes=True, maximum_span_guess=8)  # 8 x 0.3 of strength >= SWAP's 1.5
    assert w8.get_spanning_range(swap) == range(8, 9) and w8.get_spanning_range(o.cx_matrix()) == range(5, 6)


def test_qiskit_parameter_order_helpers():
    """The reference zips Xk with name-sorted parameters (P0, P1, P10, ..., P2, ...)."""
    order = CircuitTemplate.qiskit_parameter_order(12)
    assert order == [0, 1, 10, 11, 2, 3, 4, 5, 6, 7, 8, 9]
    x = np.arange(24.0)
    assert np.array_equal(CircuitTemplate.from_qiskit_order(CircuitTemplate.to_qiskit_order(x)), x)
    assert CircuitTemplate.qiskit_parameter_order(6) == list(range(6))


def test_data_dict_entry():
    d = DataDictEntry(1, 1e-12, [0.0] * 12, 1)
    assert (d.success_label, d.loss_result, d.cycles) == (1, 1e-12, 1)


def test_shard_range_and_local_merge():
    for n, w in ((262144, 8), (10, 3), (5, 8), (0, 2)):
        cover = []
        for r in range(w):
            f, c = shard_range(n, r, w)
            cover.extend(range(f, f + c))
        assert cover == list(range(n))
    loss, x, cyc = merge_results(LocalComm(), 4, 0, np.arange(4.0), np.ones((4, 6)), np.array([1, 2, 3, 1]))
    assert np.array_equal(loss, np.arange(4.0)) and x.shape == (4, 6) and list(cyc) == [1, 2, 3, 1]


def test_two_gate_coverage_regions_hold_on_sampled_products():
    """span_rules.two_gate_region (round 4): the Weyl coordinates of g2 . L . g1 for random local L all lie inside the region the
    rules state -- and reach its faces -- for iSWAP . L . B and for two XY-type gates (a, a, 0), a <= 1/4 (CPU: NumPy sampling, the
    way the regions were obtained, tools/fit_two_gate_region.py)."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fit_two_gate_region as fit

    from slam_decomposition_amd import span_rules
    from slam_decomposition_amd.weyl import c1c2c3

    for s1, s2 in (("iswap", "b"), ("b", "iswap"), ("riswap:0.4", "riswap:0.4"), ("sqiswap", "sqiswap"), ("riswap:0.2", "riswap:0.2"), ("cx", "cx"), ("b", "b")):
        g1, g2 = fit.gate(s1), fit.gate(s2)
        f = fit.fold(fit.sample_products(g1, g2, 30000, seed=5))
        region = span_rules.two_gate_region(c1c2c3(g1), c1c2c3(g2))
        assert region is not None
        assert region(f[:, 0], f[:, 1], f[:, 2], 1e-9).all(), (s1, s2)
    # the faces are reached (the region is not larger than the samples say): iSWAP . L . B touches x = 1/4 and |z| = 1/4
    f = fit.fold(fit.sample_products(fit.gate("iswap"), fit.gate("b"), 30000, seed=6))
    assert f[:, 0].min() < 0.2502 and np.abs(f[:, 2]).max() > 0.2495 and f[:, 0].max() > 0.4995
    # pairs without a closed form go through coverage.py: a general conversion-gain gate with iSWAP, an XY-type gate with a > 1/4
    isw = c1c2c3(fit.gate("iswap"))
    for s1, s2 in (("cg:0.3:0.2", "iswap"), ("riswap:0.7", "riswap:0.7"), ("cg:0.9:0.4", "cg:0.2:0.7")):
        g1, g2 = fit.gate(s1), fit.gate(s2)
        f = fit.fold(fit.sample_products(g1, g2, 30000, seed=7))
        assert span_rules.two_gate_region(c1c2c3(g1), c1c2c3(g2))(f[:, 0], f[:, 1], f[:, 2], 1e-7).all(), (s1, s2)
    g = c1c2c3(fit.gate("riswap:0.4"))
    assert span_rules.sequence_is_exact([isw, g, isw], 3) and span_rules.sequence_is_exact([g, g, g], 3) and not span_rules.sequence_is_exact([g], 2)


def test_target_data_list_semantics_on_the_host():
    from slam_decomposition_amd.basis_abc import DataDictEntry, TargetDataList

    x = np.arange(4 * 24, dtype=float).reshape(4, 24)
    cyc = np.array([2, 3, 2, 1])
    d = TargetDataList(np.array([True, True, False, True]), np.array([1e-12, 2e-12, 0.5, 3e-12]), x, cyc, lambda c: 6 * (c + 1))
    assert len(d) == 4 and d[0].cycles == 2 and len(d[0].Xk) == 18 and len(d[1].Xk) == 24 and len(d[3].Xk) == 12
    assert d[2].success_label == 0 and d[2].loss_result == 0.5 and d[-1] is d[3] and d[1] is d[1]
    assert [e.cycles for e in d] == [2, 3, 2, 1] and isinstance(d[1:3], list) and len(d[1:3]) == 2
    plain = [DataDictEntry(int(l), float(v), x[i, : 6 * (c + 1)], int(c)) for i, (l, v, c) in enumerate(zip([1, 1, 0, 1], [1e-12, 2e-12, 0.5, 3e-12], cyc))]
    assert d == plain and plain == d and not (d == plain[:3])
    import pytest

    with pytest.raises(IndexError):
        d[4]
    (a, b, c, e) = d  # unpacks like a list
    assert a is d[0] and e is d[3]


def test_fd_reference_fixture_is_what_the_oracle_computes():
    """tests/golden/fd_reference.npz (the GPU parity test's reference-path results) spot-checked against a live run of the same
    oracle call: scipy BFGS with finite differences, sequential restarts (src/slam/optimizer.py:233-303)."""
    import os

    from oracle import slam_oracle as o

    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fd_reference.npz"))
    assert int(ref["n"]) >= 64
    idx = 3
    target = o.haar_philox_port(int(ref["target_seed"]), idx)
    best, xk, k, _ = o.run_reference(target, [o.riswap_matrix(0.5)], range(1, 4), int(ref["restarts"]), float(ref["level"]),
                                     x0_fn=lambda kk, r: o.x0_philox(int(ref["opt_seed"]), idx, r, kk), analytic_jac=False)
    assert k == int(ref["sqiswap_cycles"][idx]) and abs(best - float(ref["sqiswap_loss"][idx])) < 1e-12
    assert np.max(np.abs(o.c1c2c3_raw(target) - ref["sqiswap_target_coords"][idx])) < 1e-12


def test_mixed_order_template_constructor_and_coverage_set():
    """MixedOrderBasisCircuitTemplate (basis.py:213-359): the constructor's checks and errors, gc < gg ordering, unit duration, the
    coverage set sorted by cost, set_polytope / build / unit_cost -- no device needed."""
    from slam_decomposition_amd.basis import CircuitCoverage, MixedOrderBasisCircuitTemplate
    from slam_decomposition_amd.gates import ConversionGainGate, RiSwapGate

    pi = np.pi
    with pytest.raises(ValueError, match="just don't do this lol"):
        MixedOrderBasisCircuitTemplate([ConversionGainGate(0, 0, pi / 4, 0, 1)], cost_1q=0.1)
    with pytest.raises(ValueError, match="just don't do this lol"):
        MixedOrderBasisCircuitTemplate([ConversionGainGate(0, 0, pi / 4, 0, 1)], bare_cost=False)
    with pytest.raises(ValueError, match="all base gates must be ConversionGainGate"):
        MixedOrderBasisCircuitTemplate([RiSwapGate(0.5)])
    with pytest.raises(ValueError, match="Smush Polytope not in memory"):
        MixedOrderBasisCircuitTemplate([ConversionGainGate(0, 0, pi / 4, 0, 1)], use_smush_polytope=True)
    with pytest.raises(ValueError, match="need unique gate strings"):
        MixedOrderBasisCircuitTemplate([ConversionGainGate(0, 0, pi / 4, 0, 1), ConversionGainGate(0, 0, 0, pi / 4, 1)])  # equal after gc < gg
    # sqrt(iSWAP) given with t = 2 and half the strength, iSWAP: durations become 1, gc <= gg, costs 0.5 and 1
    b = MixedOrderBasisCircuitTemplate([ConversionGainGate(0, 0, pi / 8, 0, 2), ConversionGainGate(0, 0, pi / 2, 0, 1)], maximum_span_guess=3)
    assert not b.homogenous and b.use_polytopes and b.spanning_range is None and b.scores is None
    for g in b.base_gates:
        assert g.params[4] == 1 and g.params[2] <= g.params[3]
    assert [round(g.cost(), 12) for g in b.base_gates] == [0.5, 1.0]
    assert list(b.gate_hash) == [str(g) for g in b.base_gates]
    costs = [e.cost for e in b.coverage]
    assert costs == sorted(costs) and len(b.coverage) == 2 + 3 + 4
    assert [tuple(e.gate_indices) for e in b.coverage[:5]] == [(0,), (1,), (0, 0), (0, 1), (0, 0, 0)]
    assert all(isinstance(e, CircuitCoverage) and e.operations == [str(b.base_gates[i]) for i in e.gate_indices] for e in b.coverage)
    # every entry's region is exact (coverage.py)
    assert all(e.exact for e in b.coverage) and b.span_rules_exact
    # membership: sqrt(iSWAP)'s own class in entry (0,), CNOT in (0, 0) [|z| <= x - y], SWAP in no two-gate entry
    cnot, swap, sq = np.array([[0.5, 0, 0]]), np.array([[0.5, 0.5, 0.5]]), np.array([[0.25, 0.25, 0]])
    assert b.coverage[0].has_element(sq) and not b.coverage[0].has_element(cnot)
    assert b.coverage[2].has_element(cnot) and not b.coverage[2].has_element(swap)
    # [sqrt(iSWAP), iSWAP]: the plain product inside; CNOT and SWAP -- the latter exactly ON the strength bounds of round 3,
    # m1 = 1.5 = 0.5 + 1, m2 = 0.75 = 0.25 + 0.5 -- outside
    from slam_decomposition_amd.weyl import c1c2c3

    prod = np.array([c1c2c3(b.gate_matrices[1] @ np.kron(o.u3(0.3, 0.2, 0.1), o.u3(1.0, 0.5, 0.2)) @ b.gate_matrices[0])])
    assert b.coverage[3].has_element(prod) and not b.coverage[3].has_element(cnot) and not b.coverage[3].has_element(swap)
    assert list(b.minimal_spans(np.concatenate([sq, cnot, swap, np.zeros((1, 3))]))) == [1, 2, 3, 0]
    # set_polytope / build / gate_sequence / unit_cost
    with pytest.raises(AssertionError):
        b.build(2)
    b.set_polytope(b.coverage[3])
    b.build(2)
    assert b.cycles == 2 and b.gate_sequence() == [0, 1] and b.unit_cost(2) == 1.5 and b.n_params == 18
    with pytest.raises(AssertionError):
        b.build(3)
    with pytest.raises(ValueError, match="hacky substitute"):
        b.build(2, scaled_gate=ConversionGainGate(0, 0, 0, pi / 4, 1))
    h = MixedOrderBasisCircuitTemplate([ConversionGainGate(0, 0, pi / 4, 0, 1)])
    assert h.homogenous and h.span_rules_exact and [len(e) for e in h.coverage] == [1, 2, 3, 4, 5]
    h.set_polytope(h.coverage[1])
    h.build(2, scaled_gate=ConversionGainGate(0, 0, 0, pi / 8, 1))
    assert h.gate_sequence() == [0, 0] and abs(h.base_gates[0].params[3] - pi / 8) < 1e-15


def test_no_exterior_template_structure_and_row_blocks():
    """CircuitTemplate(no_exterior_1q=True) (src/slam/basis.py:57,154,165): G_k K_{k-1} ... K_1 G_1 has 6 (k - 1) parameters, which
    sit in the interior of a device row; RowBlocks: the rows of several windows' blocks seen as one array (no copy)."""
    from slam_decomposition_amd.basis_abc import RowBlocks, TargetDataList

    b = CircuitTemplate(base_gates=[RiSwapGate(0.5)], no_exterior_1q=True, maximum_span_guess=3)
    b.build(3)
    assert b.n_params == 12 and b.param_slice(3) == slice(6, 18)
    x = np.arange(12.0) + 1
    full = b.device_vector(x)
    assert full.shape == (1, 24) and np.all(full[0, :6] == 0) and np.all(full[0, 18:] == 0) and np.array_equal(full[0, 6:18], x)
    gl = b.to_gate_list(x)
    assert [g[0] for g in gl] == ["gate", "u", "u", "gate", "u", "u", "gate"]
    assert gl[1] == ("u", 0, (1.0, 2.0, 3.0)) and gl[5] == ("u", 1, (10.0, 11.0, 12.0))
    full_t = CircuitTemplate(base_gates=[RiSwapGate(0.5)], maximum_span_guess=3)
    full_t.build(2)
    assert full_t.n_params == 18 and full_t.param_slice(2) == slice(0, 18) and full_t.device_vector(np.zeros(18)).shape == (1, 18)

    blocks = [np.arange(12.0).reshape(3, 4), np.arange(12.0, 20.0).reshape(2, 4)]
    rb = RowBlocks(blocks)
    assert len(rb) == 5 and rb.shape == (5, 4) and rb.ndim == 2
    assert np.array_equal(rb[3], blocks[1][0]) and np.array_equal(rb[-1], blocks[1][1]) and np.array_equal(rb[2, 1:3], blocks[0][2, 1:3])
    assert np.array_equal(rb.as_array(), np.concatenate(blocks))
    with pytest.raises(IndexError):
        rb[5]
    with pytest.raises(ValueError):
        RowBlocks([np.zeros((1, 2)), np.zeros((1, 3))])
    # a lazy target_data over blocks, entries cut by a slice (no_exterior) or a width
    td = TargetDataList(np.ones(5, int), np.zeros(5), rb, np.array([1, 1, 1, 1, 1]), lambda c: slice(1, 3))
    assert np.array_equal(td[4].Xk, blocks[1][1][1:3]) and td[0].cycles == 1
    td2 = TargetDataList(np.ones(5, int), np.zeros(5), rb, np.array([1, 1, 1, 1, 1]), lambda c: 2)
    assert np.array_equal(td2[3].Xk, blocks[1][0][:2])


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("group,n_streams", [(1, 5), (1, 16), (4, 3), (20, 4), (3, 1)])
def test_merge_slices_tile_the_job_vector(world, group, n_streams):
    """The slice arithmetic of the N > 1 bench's device-to-device merge (parallel.merge_slices, what feeds slam_comm_merge_add):
    for 1 / 2 / 4 / 8 ranks, grouped and ungrouped steps, the slices of all ranks cover the job vector exactly once, every step's
    window comes from the context that ran it and from where that step sits in the context's resident array, and rank r's block
    is [r n_loc, (r + 1) n_loc) with its steps in order."""
    from slam_decomposition_amd.parallel import merge_slices, step_groups

    n_per_step, warmup, steps = 8192, 5, 20
    step_ids = list(range(warmup, warmup + steps))
    n_loc = steps * n_per_step
    cover = np.zeros(world * n_loc, dtype=np.int32)
    for rank in range(world):
        seen = []
        groups = step_groups(step_ids, group)
        owner = {s: gi % n_streams for gi, g in enumerate(groups) for s in g}
        for w, lf, cnt, gf in merge_slices(step_ids, warmup, n_per_step, rank, world, n_streams, group):
            assert cnt == n_per_step and lf % n_per_step == 0
            s = lf // n_per_step
            assert owner[s] == w
            assert gf == rank * n_loc + (s - warmup) * n_per_step
            cover[gf : gf + cnt] += 1
            seen.append(s)
        assert sorted(seen) == step_ids
    assert np.all(cover == 1)
    with pytest.raises(ValueError):
        list(merge_slices([5, 7], 5, 16, 0, 1, 1))
    with pytest.raises(ValueError):
        list(merge_slices(step_ids, warmup, n_per_step, world, world, 1))


def test_lazy_list_behaves_like_the_reference_lists():
    """``training_loss`` / ``best_cycle_list`` (optimizer.py:38-40,307-311) as LazyList: list behaviour with array chunks inside."""
    from slam_decomposition_amd.basis_abc import LazyList

    l = LazyList()
    assert len(l) == 0 and list(l) == [] and l == []
    l.insert(0, 1.5)
    l.extend_array(np.array([2.0, 3.0]))
    l.append(4.0)
    l.extend([5.0])
    l.extend_array(np.array([7, 8], dtype=np.int32))
    l.append([-1, 2, 0.5])  # use_callback appends lists (optimizer.py:238,291)
    assert len(l) == 8 and l[1] == 2.0 and type(l[1]) is float and type(l[6]) is int and l[-1] == [-1, 2, 0.5]
    assert l[1:4] == [2.0, 3.0, 4.0] and l == [1.5, 2.0, 3.0, 4.0, 5.0, 7, 8, [-1, 2, 0.5]] and l != [1.5]
    assert l + [1] == list(l) + [1] and [0] + l == [0] + list(l)
    del l[-1]
    l[0] = 9.0
    assert list(l) == [9.0, 2.0, 3.0, 4.0, 5.0, 7, 8]
    with pytest.raises(IndexError):
        l[7]
    m = LazyList()
    m.extend_array(np.arange(5.0))
    m.extend_array(np.arange(5.0, 8.0))
    assert np.array_equal(np.asarray(m), np.arange(8.0)) and m == LazyList(list(np.arange(8.0))) and m.tolist() == list(range(8))
    assert sum(1 for _ in m) == 8 and all(type(v) is float for v in m)


@pytest.mark.parametrize("n,window,helpers,stagger", [(327680, 65536, 4, False), (327680, 65536, 5, False), (5000, 1024, 4, False), (5000, 1999, 3, False),
                                                       (131073, 65536, 5, True), (65537, 65536, 4, True), (70000, 65536, 4, False), (7, 2, 4, True)])
def test_window_plan_tiles_the_sampler_in_contiguous_helper_shares(n, window, helpers, stagger):
    """``TemplateOptimizer._window_plan``: every helper owns ONE contiguous share (it is made resident before the first window starts), the
    shares differ by at most one target, windows are at most WINDOW_TARGETS, and together they tile [0, n) in order."""
    from slam_decomposition_amd.optimizer import TemplateOptimizer

    class Knobs:
        WINDOW_TARGETS, windows_in_flight, window_stagger = window, helpers, stagger

    plan = TemplateOptimizer._window_plan(Knobs(), n)
    assert 1 <= len(plan) <= helpers and len(plan) <= -(-n // window)
    flat = [w for wins in plan for w in wins]
    assert flat[0][0] == 0 and flat[-1][0] + flat[-1][1] == n
    assert all(a[0] + a[1] == b[0] for a, b in zip(flat, flat[1:]))
    assert all(0 < c <= window for _, c in flat)
    shares = [sum(c for _, c in wins) for wins in plan]
    assert max(shares) - min(shares) <= 1
    if stagger and len(plan) > 1 and min(shares) >= 2 * len(plan):
        assert len({wins[0][1] for wins in plan}) > 1  # first windows of different sizes


class _FakeWindowCtx:
    """Stands in for a helper context of ``_run_batch_windows``: 'decomposes' a window into its GLOBAL target indices."""

    def __init__(self, slot, fail_slot=None):
        self.slot, self.fail_slot, self.n = slot, fail_slot, 0

    def set_gates(self, g):
        if self.slot == self.fail_slot:
            raise RuntimeError("helper %d cannot come up" % self.slot)

    def set_cost(self, kind):
        pass

    def set_targets(self, t):
        self.n = len(t)
        self.first_value = float(np.real(t[0, 0, 0]))

    def reset_stats(self):
        pass

    def stats(self):
        return {"evals": [0, 1, 0, 0], "total_ms": 1.0}

    def decompose_range(self, first, count, k_min, k_max, gate_seqs, params, thr):
        assert 0 <= first and first + count <= self.n and params.target_base == int(self.first_value)
        idx = params.target_base + first + np.arange(count)
        return idx.astype(np.float64), np.repeat(idx[:, None], 6 * (k_max + 1), axis=1).astype(np.float64), np.full(count, k_max, dtype=np.int32)


@pytest.mark.parametrize("n,window,helpers", [(1000, 128, 4), (1001, 300, 3), (257, 256, 4)])
def test_windowed_path_puts_the_windows_back_in_target_order(monkeypatch, n, window, helpers):
    """``TemplateOptimizer._run_batch_windows`` with stand-in contexts: every helper makes its contiguous share resident, runs its windows
    with ``target_base`` = the share's first target, and the results come back in target order (losses, cycles, parameter blocks)."""
    from slam_decomposition_amd import optimizer as O, runtime

    monkeypatch.setattr(runtime, "get_context", lambda device, slot=0: _FakeWindowCtx(slot))
    opt = O.TemplateOptimizer(CircuitTemplate(base_gates=[CXGate()], maximum_span_guess=3), BasicCost(), training_restarts=4, seed=1, windows_in_flight=helpers)
    opt.WINDOW_TARGETS = window
    opt._want_span_losses = False
    targets = np.zeros((n, 4, 4), dtype=np.complex128)
    targets[:, 0, 0] = np.arange(n)  # a target "is" its index
    loss, xs, cycles = opt._run_batch_windows(n, targets, [1, 2, 3], [[0], [0, 0], [0, 0, 0]], opt._opt_params())
    assert np.array_equal(loss, np.arange(n)) and np.all(cycles == 3)
    assert all(xs[i][0] == i for i in (0, 1, n // 2, n - 1))
    assert len(opt.last_stats_per_device) == sum(len(w) for w in opt._window_plan(n)) and opt.last_stats["evals"][1] == len(opt.last_stats_per_device)


def test_windowed_path_surfaces_a_failing_helper_without_hanging(monkeypatch):
    """A helper that fails while making its share resident aborts the barrier the others wait at; its exception is the one raised."""
    import threading

    from slam_decomposition_amd import optimizer as O, runtime

    monkeypatch.setattr(runtime, "get_context", lambda device, slot=0: _FakeWindowCtx(slot, fail_slot=2))
    opt = O.TemplateOptimizer(CircuitTemplate(base_gates=[CXGate()], maximum_span_guess=3), BasicCost(), training_restarts=4, seed=1, windows_in_flight=4)
    opt.WINDOW_TARGETS = 64
    opt._want_span_losses = False
    targets = np.zeros((600, 4, 4), dtype=np.complex128)
    targets[:, 0, 0] = np.arange(600)
    result = {}

    def call():
        try:
            opt._run_batch_windows(600, targets, [1, 2, 3], [[0], [0, 0], [0, 0, 0]], opt._opt_params())
        except Exception as exc:
            result["exc"] = exc

    t = threading.Thread(target=call, daemon=True)
    t.start()
    t.join(timeout=20)
    assert not t.is_alive(), "the helpers deadlocked at the barrier"
    assert isinstance(result.get("exc"), RuntimeError) and "helper 2" in str(result["exc"])
