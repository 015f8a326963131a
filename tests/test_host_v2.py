"""CPU: host logic of CircuitTemplateV2 (src/slam/basisv2.py:27-299) -- gate-callable probing, parameter naming and
ordering, bounds, the user <-> device parameter layouts -- and the V2 oracle's internal consistency.  No GPU."""
import numpy as np
import pytest

from oracle import slam_oracle as o
from oracle import v2_oracle as v
from slam_decomposition_amd.basisv2 import DEFAULT_BOUND, CircuitTemplateV2, gate_map
from slam_decomposition_amd.gates import ConversionGainGate, CXGate, RiSwapGate


def test_gate_map_of_the_reference_callers_gates():
    qn, sel, scale, off = gate_map(RiSwapGate)  # decomp_trajectory.ipynb cell 5
    assert qn == 1 and sel == [0, -1, -1, -1] and np.isclose(scale[0], -np.pi / 2) and off == [0.0] * 4
    # RiSwap(alpha) really is that member of the conversion-gain family
    for a in (0.0, 0.5, 1.0, -0.3, 2.7):
        assert np.allclose(RiSwapGate(a).to_matrix(), v.cg_matrix([-0.5 * np.pi * a, 0, 0, 0]), atol=1e-15)
    qn, sel, scale, off = gate_map(lambda gc, gg: ConversionGainGate(0.3, -0.2, gc, gg, 1.5))  # parallel_drive_volume.py:91-96
    assert qn == 2 and sel == [0, -1, 1, -1] and np.allclose(scale, [1.5, 0, 1.5, 0]) and np.allclose(off, [0, 0.3, 0, -0.2])
    qn, sel, scale, off = gate_map(lambda p1, p2: ConversionGainGate(p1, p2, 0.9, 0.4, 2.0))  # family_extend.py:40-47
    assert qn == 2 and sel == [-1, 0, -1, 1] and np.allclose(off, [1.8, 0, 0.8, 0])
    for fn in (ConversionGainGate, lambda a, b: ConversionGainGate(0, 0, a * b, b, 1), lambda a, b: ConversionGainGate(0, 0, a + b, b, 1),
               lambda a: CXGate(), lambda a, b, c, d, e, f: RiSwapGate(a)):
        with pytest.raises(NotImplementedError):
            gate_map(fn)


def test_parameter_names_orders_and_layouts():
    b = CircuitTemplateV2(base_gates=[lambda p1, g1, g2: ConversionGainGate(p1, 0.1, g1, g2, 0.8)])  # 3 -> 4 on the device
    b.build(3)
    names = b.parameter_names()
    assert names[:3] == ["P0", "P1", "P2"] and names[24:] == [f"Q{i}" for i in range(9)] and b.n_params == 33
    x = np.arange(33, dtype=float)
    xq = b.to_qiskit_order(x)  # name-sorted: P0, P1, P10, P11, ..., P2, P20, ..., Q0, ...
    assert xq[2] == 10 and xq[3] == 11 and np.array_equal(b.from_qiskit_order(xq), x)
    n_dev, idx, ilo, ihi, blo, bhi = b.device_layout(3)
    assert n_dev == 24 + 4 * 3 and len(idx) == 33 and len(set(idx)) == 33
    assert list(idx[24:]) == [24, 25, 26, 28, 29, 30, 32, 33, 34]  # the dummy fourth gate parameter is skipped ...
    dummy = np.setdiff1d(np.arange(n_dev), idx)
    assert list(dummy) == [27, 31, 35] and np.all(blo[dummy] == 0) and np.all(bhi[dummy] == 0)  # ... and fixed at 0
    assert np.all(ilo[idx] == DEFAULT_BOUND[0]) and np.all(ihi[idx] == DEFAULT_BOUND[1]) and np.all(np.isinf(blo[idx]))
    assert np.array_equal(b.from_device_vector(b.to_device_vector(x)[0]), x)
    # vz_only: one rz angle per qubit and layer -> the lambda slot of U(0, 0, lambda); theta, phi fixed at 0
    z = CircuitTemplateV2(base_gates=[RiSwapGate], vz_only=True)
    z.build(2)
    assert z.parameter_names() == ["P0", "P1", "P2", "P3", "P4", "P5", "Q0", "Q1"]
    n_dev, idx, ilo, ihi, blo, bhi = z.device_layout(2)
    assert list(idx) == [2, 5, 8, 11, 14, 17, 18, 19]
    fixed = np.setdiff1d(np.arange(n_dev), idx)
    assert np.all(blo[fixed] == 0) and np.all(bhi[fixed] == 0) and np.all(ilo[fixed] == 0) and np.all(ihi[fixed] == 0)


def test_bounds_follow_the_reference_semantics():
    b = CircuitTemplateV2(base_gates=[RiSwapGate])
    b.build(2)
    assert b.parameter_guess() is not None and b.bounds_list is None and not b.using_bounds  # BFGS (basisv2.py:170-171)
    with pytest.raises(ValueError, match="Parameter Name not found"):
        b.add_bound("Q2", 1, 0)
    b.add_bound("Q0", max=0.5, min=0.5)
    b.add_bound("Q1", max=None, min=0.0)
    guess = b.parameter_guess()
    assert b.using_bounds and len(b.bounds_list) == 20 and b.bounds_list[0] == DEFAULT_BOUND and b.bounds_list[18] == (0.5, 0.5)
    assert guess[18] == 0.5 and 0.0 <= guess[19] <= DEFAULT_BOUND[1] and all(abs(g) <= 4 * np.pi for g in guess)
    _, idx, ilo, ihi, blo, bhi = b.device_layout(2)
    # once one bound is set every parameter is bounded: explicit, or the default (-4 pi, 4 pi) (basisv2.py:160-169)
    assert blo[idx[0]] == DEFAULT_BOUND[0] and bhi[idx[0]] == DEFAULT_BOUND[1]
    assert (blo[idx[18]], bhi[idx[18]]) == (0.5, 0.5) and blo[idx[19]] == 0.0 and np.isinf(bhi[idx[19]])
    b.set_constraint(1.0)  # (the HIP path runs cost constraints since round 3: test_constraint_layout_... below)
    assert b.using_constraints
    b.remove_constraint()
    for kw in (dict(param_vec_expand=[1, 2]), dict(n_qubits=3)):
        with pytest.raises(NotImplementedError):
            CircuitTemplateV2(**kw)
    # use_polytopes (round 5): the coverage regions describe circuits of FIXED gates -- free gate parameters raise at the lookup,
    # parameters bounded to a point give range(k, k + 1) (basisv2.py:77-85; GPU side: tests/test_gpu_round5.py)
    pv = CircuitTemplateV2(use_polytopes=True, maximum_span_guess=3)
    assert pv.spanning_range is None
    with pytest.raises(NotImplementedError):
        pv.get_spanning_range(np.eye(4)[[0, 2, 1, 3]])
    pv.build(3)
    for name in ("Q0", "Q1", "Q2"):
        pv.add_bound(name, max=1.0, min=1.0)  # three iSWAPs
    from oracle import slam_oracle as oo
    assert list(pv.get_spanning_range(oo.riswap_matrix(1.0))) == [1] and list(pv.get_spanning_range(oo.cx_matrix())) == [2]
    assert list(pv.get_spanning_range(np.eye(4)[[0, 2, 1, 3]])) == [3]  # SWAP needs three iSWAPs
    with pytest.raises(NotImplementedError):
        b.build(6)  # spans above SLAM_V2_MAX_SPAN = 5
    with pytest.raises(ValueError):
        b.build(0)
    b.build(2)
    Xk = np.concatenate([np.zeros(18), [0.5, 1.0]])
    assert b.circuit_cost(Xk) == pytest.approx(1.5) and b.circuit_fidelity(Xk) == pytest.approx(0.5)


def test_v2_oracle_vz_only_template_is_the_rz_circuit():
    """rz layers of the oracle against explicit qiskit-convention matrices; U(0, 0, l) differs from rz(l) by a phase only."""
    x = np.array([0.3, -1.1, 0.7, 2.2, 0.25])  # k = 1: rz(q0), rz(q1), rz(q0), rz(q1), alpha
    W = v.template_eval(x, [lambda a: o.riswap_matrix(a)], 1, 1, vz_only=True)
    ref = np.kron(v.rz(x[3]), v.rz(x[2])) @ o.riswap_matrix(x[4]) @ np.kron(v.rz(x[1]), v.rz(x[0]))
    assert np.allclose(W, ref, atol=1e-15)
    assert np.allclose(o.u3(0, 0, 0.7), np.exp(0.35j) * v.rz(0.7), atol=1e-15)


def test_constraint_layout_and_the_multiplier_method_port():
    """set_constraint (basisv2.py:192-200) on the host: the probed half-space of the device, its refusals, and oracle/pqn_port.py
    -- the NumPy restatement of the kernel's multiplier method -- against SciPy SLSQP on a small constrained problem."""
    import scipy.optimize as opt

    from oracle import pqn_port
    from slam_decomposition_amd.basisv2 import CircuitTemplateV2
    from slam_decomposition_amd.gates import ConversionGainGate, RiSwapGate

    basis = CircuitTemplateV2(base_gates=[RiSwapGate])
    basis.build(2)
    basis.set_constraint(0.7)
    assert basis.using_constraints and basis.constraint_func["type"] == "ineq"
    x = np.r_[np.zeros(18), 0.25, 0.35]
    assert abs(basis.constraint_func["fun"](x) - 0.1) < 1e-12  # C(x) = param_max_cost - circuit_cost(x) >= 0
    w, cm = basis.constraint_layout(2)
    assert np.array_equal(w, np.r_[np.zeros(18), 1.0, 1.0]) and cm == 0.7  # RiSwapGate.cost() = alpha, no bounds needed
    basis.remove_constraint()
    assert not basis.using_constraints and basis.constraint_func is None
    cg = CircuitTemplateV2(base_gates=[lambda gc, gg: ConversionGainGate(0.0, 0.0, gc, gg, 2.0)])
    cg.build(1)
    cg.set_constraint(1.0)
    with pytest.raises(NotImplementedError):
        cg.constraint_layout(1)  # |gc| + |gg| is not affine around 0
    cg.add_bound("Q0", 1.0, 0.0)
    cg.add_bound("Q1", 0.0, -1.0)  # a negative drive strength: cost falls with the parameter
    w, cm = cg.constraint_layout(1)
    assert np.allclose(w[12:], [4 / np.pi, -4 / np.pi]) and abs(cm - 1.0) < 1e-12
    cg.set_constraint(-1.0)
    with pytest.raises(ValueError):
        cg.constraint_layout(1)

    # the port on a convex quadratic with one active bound and an active constraint: SLSQP's solution
    rng = np.random.default_rng(0)
    A = rng.normal(size=(6, 6))
    Q = A @ A.T + 0.5 * np.eye(6)
    b = rng.normal(size=6) * 3
    fun = lambda xx: (0.5 * xx @ Q @ xx - b @ xx, Q @ xx - b)
    lo, hi = np.full(6, -0.4), np.full(6, 2.0)
    wv = np.array([1.0, 2.0, 0.0, 0.5, 0.0, 1.0])
    free = opt.minimize(fun, np.zeros(6), jac=True, method="L-BFGS-B", bounds=list(zip(lo, hi)))
    cmax = wv @ free.x - 0.8  # cuts the bounded optimum off
    ref = opt.minimize(fun, np.zeros(6), jac=True, method="SLSQP", bounds=list(zip(lo, hi)), options={"ftol": 1e-14},
                       constraints={"type": "ineq", "fun": lambda xx: cmax - wv @ xx, "jac": lambda xx: -wv})
    f, xs, iters, status, nev, mu = pqn_port.minimize_port(fun, np.zeros(6), lo, hi, wv, cmax, stop_loss=-np.inf)
    assert status in (0, 4) and mu > 0 and wv @ xs <= cmax and wv @ xs > cmax - 1e-7
    assert abs(f - ref.fun) < 1e-6 and np.allclose(xs, ref.x, atol=1e-4)
    f2, x2, *_ = pqn_port.minimize_port(fun, np.zeros(6), lo, hi, None, 0.0, stop_loss=-np.inf)
    assert abs(f2 - free.fun) < 1e-7  # without the constraint: the L-BFGS-B optimum
