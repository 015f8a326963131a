"""GPU: templates with parametrised 2Q gates -- CircuitTemplateV2 (src/slam/basisv2.py:27-299).
Loss, template unitary and the gradient with respect to the U-gate AND the gate parameters against the NumPy oracle
(oracle/v2_oracle.py); the projected quasi-Newton loop against SciPy's L-BFGS-B / BFGS on the oracle; the reference's own
recorded V2 + SquareCost run (scripts/decomp_trajectory.ipynb:84-90,140-162)."""
import json
import os

import numpy as np
import pytest
import scipy.optimize as opt

from oracle import slam_oracle as o
from oracle import v2_oracle as v
from slam_decomposition_amd import _ffi
from slam_decomposition_amd.basisv2 import CircuitTemplateV2, gate_map
from slam_decomposition_amd.cost_function import BasicCost, SquareCost
from slam_decomposition_amd.gates import ConversionGainGate, RiSwapGate, SwapGate
from slam_decomposition_amd.optimizer import TemplateOptimizer
from slam_decomposition_amd.sampler import GateSample, HaarBatch

pytestmark = pytest.mark.gpu

GATE_FNS = {
    "riswap": RiSwapGate,                                                       # 1 parameter (decomp_trajectory.ipynb cell 5)
    "cg_gc_gg": lambda gc, gg: ConversionGainGate(0.3, -0.2, gc, gg, 1.5),      # 2: drive strengths (parallel_drive_volume.py:91-96)
    "cg_phases": lambda p1, p2: ConversionGainGate(p1, p2, 0.9, 0.4, 1.0),      # 2: phases (family_extend.py:40-47)
    "cg_3": lambda p1, g1, g2: ConversionGainGate(p1, 0.1, g1, g2, 0.8),        # 3 -> padded to 4 on the device
    "cg_4": lambda p1, p2, g1, g2: ConversionGainGate(p1, p2, g1, g2, 0.7),     # 4
}


@pytest.mark.parametrize("name", sorted(GATE_FNS))
@pytest.mark.parametrize("k", [1, 2, 3])
@pytest.mark.parametrize("vz_only", [False, True])
def test_v2_loss_unitary_and_full_gradient_match_the_oracle(hip_ctx, name, k, vz_only):
    fn = GATE_FNS[name]
    qn, sel, scale, off = gate_map(fn)
    basis = CircuitTemplateV2(base_gates=[fn], vz_only=vz_only)
    basis.build(k)
    rng = np.random.default_rng(100 * k + qn)
    M = 37
    targets = o.haar_batch(5, seed0=31)
    X = rng.uniform(-4 * np.pi, 4 * np.pi, (M, basis.n_params))
    tof = rng.integers(0, 5, M).astype(np.int32)
    hip_ctx.set_targets(targets)
    hip_ctx.v2_set_gates(basis._gate_maps)
    for square, kind in ((False, _ffi.COST_BASIC), (True, _ffi.COST_SQUARE)):
        hip_ctx.set_cost(kind)
        loss, grad, W = hip_ctx.v2_eval(basis.gate_sequence(), basis.to_device_vector(X), tof, want_unitary=True)
        _, idx, *_ = basis.device_layout(k)
        fns = [lambda *q: fn(*q).to_matrix()] * k
        for m in range(M):
            Wref = v.template_eval(X[m], fns, qn, k, vz_only)   # built from the gate OBJECTS' own matrices
            if vz_only:
                # rz(l) = e^{-i l / 2} U(0, 0, l): the device template differs by a global phase only
                ph = np.vdot(Wref.ravel(), W[m].ravel())
                assert abs(abs(ph) - 4.0) < 1e-11
                assert np.max(np.abs(W[m] - Wref * ph / abs(ph))) < 1e-12
            else:
                assert np.max(np.abs(W[m] - Wref)) < 1e-12
            f, g = v.loss_and_grad(X[m], [(sel, scale, off)] * k, qn, k, targets[tof[m]], vz_only, square)
            assert abs(loss[m] - f) < 1e-12
            assert np.max(np.abs(grad[m][idx] - g)) < 1e-12        # U-gate and GATE parameters
            fixed = np.setdiff1d(np.arange(grad.shape[1]), idx)    # device slots no user parameter maps to
            assert fixed.size == grad.shape[1] - basis.n_params
    hip_ctx.set_cost(_ffi.COST_BASIC)


def test_oracle_gradient_is_the_derivative_of_the_gate_objects_loss():
    """The oracle's analytic gradient (angle maps) against central differences of the loss built from the gate callables."""
    T = o.haar_unitary(5)
    rng = np.random.default_rng(0)
    for name, fn in GATE_FNS.items():
        qn, sel, scale, off = gate_map(fn)
        for k, vz in ((1, False), (3, True), (2, False)):
            x = rng.uniform(-3, 3, (2 if vz else 6) * (k + 1) + qn * k)
            fns = [lambda *q: fn(*q).to_matrix()] * k
            f, g = v.loss_and_grad(x, [(sel, scale, off)] * k, qn, k, T, vz)
            assert abs(f - v.loss(x, fns, qn, k, T, vz)) < 1e-14
            assert np.max(np.abs(g - v.fd_grad(x, fns, qn, k, T, vz))) < 2e-9


def test_recorded_v2_square_cost_run_kat1():
    """scripts/decomp_trajectory.ipynb:84-90,140-162: CircuitTemplateV2(base_gates=[RiSwapGate]), every Q bounded to
    [0.5, 0.5], SquareCost, target SWAP: the recorded bound parameters give the recorded loss and coordinates on the device."""
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat1.json")))
    x = np.array(kat["params"], dtype=np.float64)
    basis = CircuitTemplateV2(n_qubits=2, base_gates=[RiSwapGate], edge_params=[[(0, 1)]])
    basis.build(3)
    basis.spanning_range = range(3, 4)
    for name in basis.parameter_names():
        if "Q" in name:
            basis.add_bound(name, 0.5, 0.5)
    Xk = np.concatenate([x, [0.5, 0.5, 0.5]])
    W = basis.eval(Xk)
    assert abs(SquareCost().unitary_fidelity(W, SwapGate().to_matrix()) - kat["square_cost_vs_swap"]) < 1e-14  # 1 - (|t|^2 + 4) / 20 at 3.6e-9: a few ulp of 1
    from slam_decomposition_amd.weyl import c1c2c3

    assert tuple(c1c2c3(W)) == tuple(kat["c1c2c3_full"])
    assert abs(basis.circuit_cost(Xk) - 1.5) < 1e-12 and abs(basis.circuit_fidelity(Xk) - 0.125) < 1e-12
    # the optimizer run of that cell: success at k = 3, Q stay at their bound
    opt3 = TemplateOptimizer(basis=basis, objective=SquareCost(), use_callback=False, override_fail=True, success_threshold=1e-7,
                             training_restarts=25, seed=3)
    td = opt3.approximate_target_U(SwapGate().to_matrix())
    assert td.success_label == 1 and td.cycles == 3 and td.loss_result < 1e-7
    assert np.array_equal(np.asarray(td.Xk)[-3:], [0.5, 0.5, 0.5])
    basis.build(3)
    assert np.max(np.abs(np.array(c1c2c3(basis.eval(td.Xk))) - 0.5)) < 1e-3  # |coordinate error| ~ sqrt(loss)


def test_recorded_riswap_sweep_through_the_c_abi(hip_ctx):
    """decomp_trajectory.ipynb cell 12 (tests/golden/kat1_riswap_sweep.json): the KAT-1 circuit with its last RiSwapGate
    at alpha = t, t = linspace(0, 0.5, 25) -- the only recorded data with RiSwapGate(alpha != 1/2).  The 25 parameter
    vectors go through ``slam_v2_eval_loss_grad`` (template unitary with the gate parameter on the device) and
    ``slam_c1c2c3``; with cell 10's x-axis mirror the triples equal the recorded ones to all 8 digits."""
    here = os.path.dirname(__file__)
    kat = json.load(open(os.path.join(here, "golden", "kat1.json")))
    sweep = json.load(open(os.path.join(here, "golden", "kat1_riswap_sweep.json")))
    basis = CircuitTemplateV2(n_qubits=2, base_gates=[RiSwapGate], edge_params=[[(0, 1)]])
    basis.build(3)
    ts = np.linspace(0, 0.5, 25)
    X = np.stack([np.concatenate([kat["params"], [0.5, 0.5, t]]) for t in ts])
    hip_ctx.set_targets(np.eye(4, dtype=np.complex128)[None])
    hip_ctx.v2_set_gates(basis._gate_maps)
    _, _, W = hip_ctx.v2_eval(basis.gate_sequence(3), basis.to_device_vector(X, 3), want_grad=False, want_unitary=True)
    coords = np.array(hip_ctx.c1c2c3(W))
    m = coords[:, 0] > 0.5
    coords[m, 0] = -1 * coords[m, 0] + 1  # "eliminating x-axis symmetry" (cell 10)
    assert np.array_equal(np.round(coords, 8), np.array(sweep["c1c2c3"]))
    # ... and the unitaries themselves against the oracle's gate objects
    for t, w in zip(ts, W):
        ref = v.template_eval(np.concatenate([kat["params"], [0.5, 0.5, t]]), [lambda a: RiSwapGate(a).to_matrix()] * 3, 1, 3)
        assert np.max(np.abs(w - ref)) < 1e-13


def _scipy_best(fn, qn, k, target, bounds, x0s, square=False):
    best = np.inf
    gm = gate_map(fn)[1:]
    for x0 in x0s:
        res = opt.minimize(lambda xx: v.loss_and_grad(xx, [gm] * k, qn, k, target, False, square), x0, jac=True,
                           method="L-BFGS-B" if bounds is not None else "BFGS", bounds=bounds, options={"maxiter": 2500})
        best = min(best, res.fun)
    return best


def test_free_gate_parameters_reach_what_scipy_reaches():
    """RiSwapGate with a FREE alpha per gate, k = 2, no bounds (plain BFGS): any 2-qubit gate is two iSWAP-family gates
    away once alpha is free (e.g. one of them iSWAP-class), so every Haar target is solved at k <= 2 -- and the device's
    converged loss matches SciPy BFGS on the oracle from the same starts to 1e-6."""
    N, R = 6, 8
    sampler = HaarBatch(seed0=555, n_samples=N)
    basis = CircuitTemplateV2(base_gates=[RiSwapGate], maximum_span_guess=2)
    optm = TemplateOptimizer(basis, BasicCost(), training_restarts=R, seed=12, override_fail=True)
    _, _, data = optm.approximate_from_distribution(sampler)
    targets = list(sampler)
    for t, td in enumerate(data):
        basis.build(td.cycles)
        assert abs(o.basic_cost(basis.eval(td.Xk), targets[t]) - td.loss_result) < 1e-12
        assert td.cycles == 2 and td.success_label == 1 and td.loss_result < 1e-10
        rng = np.random.default_rng(t)
        ref = _scipy_best(RiSwapGate, 1, 2, targets[t], None, [rng.uniform(-4 * np.pi, 4 * np.pi, 20) for _ in range(R)])
        assert abs(ref - td.loss_result) < 1e-6


def test_bounded_gate_parameters_match_scipy_lbfgsb():
    """add_bound on every Q (0 <= alpha <= 0.3: gates weaker than sqrt(iSWAP), so k = 2 cannot reach a generic target and the
    optimum sits ON the bounds): best-of-restarts loss of the projected quasi-Newton loop vs SciPy L-BFGS-B on the oracle,
    same bounds, to 1e-6; bounds respected exactly."""
    target = o.haar_unitary(77)
    basis = CircuitTemplateV2(base_gates=[RiSwapGate], maximum_span_guess=2)
    basis.build(2)
    basis.spanning_range = range(2, 3)
    for name in basis.parameter_names():
        if name.startswith("Q"):
            basis.add_bound(name, 0.3, 0.0)
    R = 24
    optm = TemplateOptimizer(basis, BasicCost(), training_restarts=R, seed=5, override_fail=True)
    td = optm.approximate_target_U(target)
    x = np.asarray(td.Xk)
    assert td.success_label == 0 and td.cycles == 2
    assert np.all(x[-2:] >= 0.0) and np.all(x[-2:] <= 0.3) and np.all(np.abs(x[:-2]) <= 4 * np.pi)
    basis.build(2)
    assert abs(o.basic_cost(basis.eval(x), target) - td.loss_result) < 1e-12
    bounds = [(-4 * np.pi, 4 * np.pi)] * 18 + [(0.0, 0.3)] * 2  # basisv2.py:160-169: every parameter is bounded once one is
    rng = np.random.default_rng(1)
    ref = _scipy_best(RiSwapGate, 1, 2, target, bounds, [np.concatenate([rng.uniform(-4 * np.pi, 4 * np.pi, 18), rng.uniform(0, 0.3, 2)]) for _ in range(R)])
    assert abs(ref - td.loss_result) < 1e-6, (ref, td.loss_result)
    assert np.max(x[-2:]) > 0.3 - 1e-9  # the optimum uses the strongest gate allowed


def test_v2_stage_restart_order_and_errors(hip_ctx):
    fn = GATE_FNS["cg_gc_gg"]
    basis = CircuitTemplateV2(base_gates=[fn])
    hip_ctx.set_targets(o.haar_batch(4, seed0=9))
    hip_ctx.v2_set_gates(basis._gate_maps)
    _, idx, ilo, ihi, blo, bhi = basis.device_layout(2)
    prm = _ffi.OptParams(restarts=6, seed=4)
    out = hip_ctx.v2_minimize_stage([0, 0], prm, 1e-10, ilo, ihi, blo, bhi)
    for t in range(4):
        below = np.nonzero(out["item_loss"][t] < 1e-10)[0]
        want = below[0] if len(below) else int(np.argmin(out["item_loss"][t]))
        assert out["best_restart"][t] == want and out["best_loss"][t] == out["item_loss"][t, want]
    again = hip_ctx.v2_minimize_stage([0, 0], prm, 1e-10, ilo, ihi, blo, bhi)
    assert np.array_equal(again["best_x"], out["best_x"])  # no early-exit flag: every restart runs to its end, bitwise reproducible
    with pytest.raises(_ffi.SlamHipError) as e:
        z = np.zeros(6 * 6 + 2 * 5)
        hip_ctx.v2_minimize_stage([0] * 5, prm, 1e-10, z, z + 1, None, None)  # span 5 with two parameters per gate: n = 46 > 41
    assert e.value.code == -3
    with pytest.raises(_ffi.SlamHipError):
        hip_ctx.v2_minimize_stage([0, 3], prm, 1e-10, ilo, ihi, blo, bhi)  # gate index outside the table
    with pytest.raises(NotImplementedError):
        CircuitTemplateV2(base_gates=[ConversionGainGate])  # 5 parameters: a = gc * t is not affine
    with pytest.raises(NotImplementedError):
        CircuitTemplateV2(base_gates=[lambda a, b: ConversionGainGate(0, 0, a + b, b, 1)])  # an angle on two parameters
    with pytest.raises(ValueError, match="Parameter Name not found"):
        b2 = CircuitTemplateV2()
        b2.build(1)
        b2.add_bound("Q7", 1, 0)


@pytest.mark.parametrize("name,k", [("riswap", 4), ("riswap", 5), ("cg_gc_gg", 4)])
def test_v2_long_spans_loss_and_gradient_match_the_oracle(hip_ctx, name, k):
    """Spans 4 and 5 (the reference's default ``maximum_span_guess=5``, basisv2.py:35): loss and the full gradient against the
    oracle, as for the short spans."""
    fn = GATE_FNS[name]
    qn, sel, scale, off = gate_map(fn)
    basis = CircuitTemplateV2(base_gates=[fn])
    basis.build(k)
    rng = np.random.default_rng(7 * k + qn)
    M = 19
    targets = o.haar_batch(3, seed0=77)
    X = rng.uniform(-4 * np.pi, 4 * np.pi, (M, basis.n_params))
    tof = rng.integers(0, 3, M).astype(np.int32)
    hip_ctx.set_targets(targets)
    hip_ctx.v2_set_gates(basis._gate_maps)
    loss, grad, W = hip_ctx.v2_eval(basis.gate_sequence(), basis.to_device_vector(X), tof, want_unitary=True)
    _, idx, *_ = basis.device_layout(k)
    fns = [lambda *q: fn(*q).to_matrix()] * k
    for m in range(M):
        assert np.max(np.abs(W[m] - v.template_eval(X[m], fns, qn, k))) < 1e-12
        f, g = v.loss_and_grad(X[m], [(sel, scale, off)] * k, qn, k, targets[tof[m]])
        assert abs(loss[m] - f) < 1e-12 and np.max(np.abs(grad[m][idx] - g)) < 1e-12


def test_default_template_v2_runs_its_default_spanning_range(hip_ctx):
    """``CircuitTemplateV2()`` with its defaults (RiSwapGate class, maximum_span_guess = 5, basisv2.py:31-35) through the
    optimizer: alpha bounded to [0, 1/2] makes the template need several gates (sqrt(iSWAP) fragments), so spans 4 and 5 are
    entered; converged losses against SciPy L-BFGS-B on the oracle from the same start points at the long spans."""
    basis = CircuitTemplateV2()
    assert list(basis.spanning_range) == [1, 2, 3, 4, 5]
    # one target that three sqrt(iSWAP)-bounded gates reach, optimised at the LONG spans explicitly: the minimiser at k = 4, 5
    T = o.haar_batch(2, seed0=505)
    hip_ctx.set_targets(T)
    hip_ctx.v2_set_gates(basis._gate_maps)
    hip_ctx.set_cost(_ffi.COST_BASIC)
    R = 4
    for k in (4, 5):
        basis.build(k)
        for name in basis.parameter_names():
            if "Q" in name:
                basis.add_bound(name, max=0.5, min=0.0)
        n_dev, idx, ilo, ihi, blo, bhi = basis.device_layout(k)
        prm = _ffi.OptParams(restarts=R, seed=11 + k, flags=0)
        rng = np.random.default_rng(k)
        x0 = rng.uniform(ilo, ihi, (2, R, n_dev))
        out = hip_ctx.v2_minimize_stage(basis.gate_sequence(k), prm, 1e-10, ilo, ihi, blo, bhi, x0=x0)
        assert np.all(out["best_loss"] < 1e-10)                      # three or more sqrt(iSWAP)-class gates reach any target
        qs = out["best_x"][:, 6 * (k + 1):]
        assert np.all(qs >= 0.0) and np.all(qs <= 0.5)               # bounds respected exactly
        gm = gate_map(RiSwapGate)[1:]
        bounds = [(-np.inf, np.inf)] * (6 * (k + 1)) + [(0.0, 0.5)] * k
        for t in range(2):
            best = min(opt.minimize(lambda xx: v.loss_and_grad(xx, [gm] * k, 1, k, T[t]), np.clip(x0[t, r], [b[0] for b in bounds], [b[1] for b in bounds]),
                                    jac=True, method="L-BFGS-B", bounds=bounds, options={"maxiter": 2500, "ftol": 1e-15, "gtol": 1e-10}).fun for r in range(R))
            assert abs(out["best_loss"][t] - best) < 1e-6
    # and the whole default loop through the API
    basis = CircuitTemplateV2()
    td = TemplateOptimizer(basis=basis, objective=BasicCost(), training_restarts=4, seed=5).approximate_target_U(T[0])
    assert td.success_label == 1 and 1 <= td.cycles <= 5


def test_v2_early_exit_drops_later_restarts_only(hip_ctx):
    """SLAM_FLAG_EARLY_EXIT on the V2 stage: a restart is not started once a LOWER-index restart of its target has ended
    below exit_loss; the stage result is the one of the run without the flag (the reference's sequential break)."""
    basis = CircuitTemplateV2(base_gates=[RiSwapGate])
    basis.build(2)
    NT = 4096  # x 8 restarts = twice the quads the chip holds at one wavefront per SIMD: later restarts are pulled after earlier ones end
    hip_ctx.set_targets(o.haar_batch(NT, seed0=901))
    hip_ctx.v2_set_gates(basis._gate_maps)
    _, idx, ilo, ihi, blo, bhi = basis.device_layout(2)
    full = hip_ctx.v2_minimize_stage([0, 0], _ffi.OptParams(restarts=8, seed=2, flags=0), 1e-10, ilo, ihi, blo, bhi)
    fast = hip_ctx.v2_minimize_stage([0, 0], _ffi.OptParams(restarts=8, seed=2, flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED), 1e-10, ilo, ihi, blo, bhi)
    assert np.array_equal(full["best_loss"], fast["best_loss"]) and np.array_equal(full["best_restart"], fast["best_restart"])
    assert np.array_equal(full["best_x"], fast["best_x"])
    skipped = fast["item_status"] == _ffi.ST_PREEMPTED
    assert skipped.any()
    for t in range(NT):
        assert not skipped[t, : fast["best_restart"][t] + 1].any()  # nothing at or below the winner is ever dropped


def test_v2_sharded_devices_equal_the_single_device_run():
    """``TemplateOptimizer(devices=[...])`` with a CircuitTemplateV2 (round 3): contiguous target shards, seeds keyed on the
    global target index -- the sharded job returns the single-device results bit for bit (two contexts on the one GPU here)."""
    targets = o.haar_batch(7, seed0=321)

    def run(devices):
        basis = CircuitTemplateV2(base_gates=[RiSwapGate], maximum_span_guess=3)
        opt_ = TemplateOptimizer(basis=basis, objective=SquareCost(), training_restarts=6, seed=17, devices=devices)
        return opt_._approximate_batch(targets, log_index=False)

    a, b = run([0]), run([0, 0, 0])
    assert [d.cycles for d in a] == [d.cycles for d in b] and all(d.success_label == 1 for d in a)
    assert [d.loss_result for d in a] == [d.loss_result for d in b]
    assert all(np.array_equal(x.Xk, y.Xk) for x, y in zip(a, b))


def test_v2_use_callback_records_the_restart_loop():
    """``use_callback=True`` with a CircuitTemplateV2 (optimizer.py:217-224 applies to any template; round 3): the
    ``[-1, k, loss ...]`` training-loss record and one coordinate triple per recorded loss, the final result that of the run
    without the callback, the last recorded loss of the winning span = the reported loss."""
    T = o.haar_batch(2, seed0=777)

    def make():
        return CircuitTemplateV2(base_gates=[RiSwapGate], maximum_span_guess=3)

    plain = TemplateOptimizer(basis=make(), objective=BasicCost(), training_restarts=4, seed=23)._approximate_batch(T, log_index=False)
    cb = TemplateOptimizer(basis=make(), objective=BasicCost(), training_restarts=4, seed=23, use_callback=True)
    data = cb._approximate_batch(T, log_index=False)
    assert [d.cycles for d in data] == [d.cycles for d in plain] and [d.loss_result for d in data] == [d.loss_result for d in plain]
    assert all(np.array_equal(a.Xk, b.Xk) for a, b in zip(data, plain))
    assert len(cb.training_loss) == len(cb.coordinate_list) >= 2
    for tl, cl in zip(cb.training_loss, cb.coordinate_list):
        assert tl[0] == -1 and tl[1] in (1, 2, 3)
        n_loss = sum(1 for i, v in enumerate(tl) if not (v == -1 or (i > 0 and tl[i - 1] == -1)))
        assert len(cl) <= n_loss and all(len(c) == 3 for c in cl)
    # the last firing of target 0: its final recorded loss is the reported one (f(x_k) after the last accepted step)
    firing = [tl for tl in cb.training_loss][len([1 for _ in range(1)]) - 1]
    assert min(abs(v - data[0].loss_result) for v in firing if v >= 0) < 1e-15


def test_v2_no_exterior_1q_template():
    """``no_exterior_1q=True`` (basisv2.py:264,291; fsim_continuous.ipynb): no 1Q layer before the first and after the last 2Q
    gate -- W = G_k K_{k-1} ... K_1 G_1.  The device template fixes the two exterior layers at U(0, 0, 0) = identity."""
    basis = CircuitTemplateV2(base_gates=[RiSwapGate], no_exterior_1q=True, maximum_span_guess=3)
    basis.build(3)
    assert basis.parameter_names() == [f"P{i}" for i in range(12)] + ["Q0", "Q1", "Q2"]
    rng = np.random.default_rng(4)
    x = rng.uniform(-3, 3, 15)
    W = basis.eval(x)
    G = lambda a: RiSwapGate(a).to_matrix()
    ref = G(x[14]) @ o.layer_matrix(x[6:12]) @ G(x[13]) @ o.layer_matrix(x[0:6]) @ G(x[12])
    assert np.max(np.abs(W - ref)) < 1e-13
    basis.build(1)
    assert basis.parameter_names() == ["Q0"] and np.max(np.abs(basis.eval([0.7]) - G(0.7))) < 1e-14
    # the optimizer recovers a target of exactly that form (there are no exterior locals to absorb anything else)
    td = TemplateOptimizer(basis=CircuitTemplateV2(base_gates=[RiSwapGate], no_exterior_1q=True, maximum_span_guess=3), objective=BasicCost(),
                           training_restarts=24, seed=6).approximate_target_U(ref)
    assert td.success_label == 1 and td.cycles == 3 and len(td.Xk) == 15


def test_v2_device_span_loop_equals_the_host_driven_one(hip_ctx):
    """``slam_v2_decompose_range`` (the V2 span loop as one chain of kernels, round 3) against the loop driven from the host with
    one ``slam_v2_minimize_stage`` per template size over the unsolved targets: same seeds, ordered early exit -> the same
    (loss, parameters, cycles) bit for bit, for a bounded and an unbounded template."""
    N, R = 48, 6
    targets = o.haar_batch(N, seed0=1234)
    for bounded in (False, True):
        basis = CircuitTemplateV2(base_gates=[RiSwapGate], maximum_span_guess=3)
        lay = {}
        for k in (1, 2, 3):
            basis.build(k)
            if bounded:
                for name in basis.parameter_names():
                    if "Q" in name:
                        basis.add_bound(name, max=0.5, min=0.0)  # sqrt(iSWAP) fragments: spans up to 3 are needed
            lay[k] = basis.device_layout(k)
        hip_ctx.set_targets(targets)
        hip_ctx.v2_set_gates(basis._gate_maps)
        hip_ctx.set_cost(_ffi.COST_BASIC)
        prm = _ffi.OptParams(restarts=R, seed=77, flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED)
        thr = 1e-10
        best = np.full(N, np.inf)
        cyc = np.full(N, -1, dtype=np.int32)
        bx = np.zeros((N, lay[3][0]))
        for k in (1, 2, 3):
            act = np.nonzero(~(best < thr))[0].astype(np.int32)
            if not len(act):
                break
            out = hip_ctx.v2_minimize_stage([0] * k, prm, thr, *lay[k][2:6], active=act, want_items=False)
            better = (cyc[act] < 0) | (out["best_loss"] < best[act])
            for j, t in enumerate(act):
                if better[j]:
                    best[t], cyc[t] = out["best_loss"][j], k
                    bx[t] = 0.0
                    bx[t, : lay[k][0]] = out["best_x"][j]
        dl, dx, dc = hip_ctx.v2_decompose_range(0, N, 1, 3, [[0] * k for k in (1, 2, 3)], [lay[k][2:6] for k in (1, 2, 3)], prm, thr)
        assert np.array_equal(dc, cyc) and np.array_equal(dl, best) and np.array_equal(dx, bx)
        assert np.all(dl < 1e-10) and (len(np.unique(dc)) >= 2)
    # a window of the resident batch, and the optimizer path built on it
    wl, _, wc = hip_ctx.v2_decompose_range(16, 8, 1, 3, [[0] * k for k in (1, 2, 3)], [lay[k][2:6] for k in (1, 2, 3)], prm, thr)
    assert np.array_equal(wl, dl[16:24]) and np.array_equal(wc, dc[16:24])


def _slsqp_best(fn, qn, k, target, bounds, w_user, cost_max, x0s, square=False):
    """SciPy SLSQP on the oracle (analytic gradient), the method the reference switches to once the template has a
    constraint (optimizer.py:260-265), with the constraint in SciPy's C(x) >= 0 form (basisv2.py:196-199).  Best feasible
    result over the start points (a single start = a polish of that point)."""
    best = np.inf
    gm = gate_map(fn)[1:]
    cons = {"type": "ineq", "fun": lambda xx: cost_max - w_user @ xx, "jac": lambda xx: -w_user}
    for x0 in x0s:
        res = opt.minimize(lambda xx: v.loss_and_grad(xx, [gm] * k, qn, k, target, False, square), x0, jac=True, method="SLSQP",
                           bounds=bounds, constraints=cons, options={"maxiter": 2500, "ftol": 1e-13})
        if res.success and w_user @ res.x <= cost_max + 1e-9:
            best = min(best, res.fun)
    return best


def test_cost_constraint_matches_scipy_slsqp_riswap():
    """set_constraint (basisv2.py:192-200): three RiSwap gates with 0 <= alpha_i <= 1 and alpha_1 + alpha_2 + alpha_3 <= 1.0 -- too
    little total interaction for a generic target, so the optimum lies ON the constraint.  Best-of-restarts loss of the device's
    multiplier method vs SciPy SLSQP on the oracle: SLSQP started AT the device's result does not get below it by 1e-6 (it is a
    constrained minimum), SLSQP from as many random starts does not beat it by 1e-6 (the landscape has several minima per target:
    best-of-restarts from different starts need not coincide); bounds and constraint hold exactly."""
    cmax = 1.0
    for seed in (31, 32, 33):
        target = o.haar_unitary(seed)
        basis = CircuitTemplateV2(base_gates=[RiSwapGate], maximum_span_guess=3)
        basis.build(3)
        basis.spanning_range = range(3, 4)
        for name in basis.parameter_names():
            if name.startswith("Q"):
                basis.add_bound(name, 1.0, 0.0)
        basis.set_constraint(cmax)
        R = 32
        optm = TemplateOptimizer(basis, BasicCost(), training_restarts=R, seed=7 + seed, override_fail=True)
        td = optm.approximate_target_U(target)
        x = np.asarray(td.Xk)
        assert td.cycles == 3
        assert np.all(x[-3:] >= 0.0) and np.all(x[-3:] <= 1.0)
        basis.build(3)
        assert basis.circuit_cost(x) <= cmax + 1e-12
        assert abs(o.basic_cost(basis.eval(x), target) - td.loss_result) < 1e-12
        bounds = [(-4 * np.pi, 4 * np.pi)] * 24 + [(0.0, 1.0)] * 3
        w_user = np.concatenate([np.zeros(24), np.ones(3)])
        rng = np.random.default_rng(seed)
        x0s = [np.concatenate([rng.uniform(-4 * np.pi, 4 * np.pi, 24), rng.uniform(0, 1.0 / 3, 3)]) for _ in range(R)]
        ref = _slsqp_best(RiSwapGate, 1, 3, target, bounds, w_user, cmax, x0s)
        polished = _slsqp_best(RiSwapGate, 1, 3, target, bounds, w_user, cmax, [x])
        assert td.loss_result - polished < 1e-6, (seed, polished, td.loss_result)
        assert td.loss_result < ref + 1e-6, (seed, ref, td.loss_result)
        if td.loss_result > 1e-6:
            assert basis.circuit_cost(x) > cmax - 1e-6  # the constraint binds


def test_cost_constraint_conversion_gain_and_errors():
    """ConversionGainGate(0, 0, gc, gg, 1): cost = (|gc| + |gg|) t / (pi / 2) (custom_gates.py:208-212) is affine once gc, gg >= 0
    are bounds; without them the HIP path refuses; an unreachable cost raises; remove_constraint restores the plain run."""
    fn = lambda gc, gg: ConversionGainGate(0.0, 0.0, gc, gg, 1.0)
    target = o.haar_unitary(91)
    basis = CircuitTemplateV2(base_gates=[fn], maximum_span_guess=2)
    basis.build(2)
    basis.spanning_range = range(2, 3)
    basis.set_constraint(0.4)
    with pytest.raises(NotImplementedError):
        basis.constraint_layout(2)  # |gc|, |gg| with sign-indefinite parameters
    for name in basis.parameter_names():
        if name.startswith("Q"):
            basis.add_bound(name, 1.5, 0.0)
    w_dev, cm = basis.constraint_layout(2)
    assert np.allclose(w_dev[18:], 2 / np.pi) and np.all(w_dev[:18] == 0.0) and abs(cm - 0.4) < 1e-12
    R = 32
    optm = TemplateOptimizer(basis, BasicCost(), training_restarts=R, seed=3, override_fail=True)
    td = optm.approximate_target_U(target)
    x = np.asarray(td.Xk)
    basis.build(2)
    assert basis.circuit_cost(x) <= 0.4 + 1e-12 and np.all(x[-4:] >= 0.0) and np.all(x[-4:] <= 1.5)
    assert abs(o.basic_cost(basis.eval(x), target) - td.loss_result) < 1e-12
    bounds = [(-4 * np.pi, 4 * np.pi)] * 18 + [(0.0, 1.5)] * 4
    w_user = np.concatenate([np.zeros(18), np.full(4, 2 / np.pi)])
    rng = np.random.default_rng(4)
    x0s = [np.concatenate([rng.uniform(-4 * np.pi, 4 * np.pi, 18), rng.uniform(0, 0.15, 4)]) for _ in range(R)]
    ref = _slsqp_best(fn, 2, 2, target, bounds, w_user, 0.4, x0s)
    polished = _slsqp_best(fn, 2, 2, target, bounds, w_user, 0.4, [x])
    assert td.loss_result - polished < 1e-6, (polished, td.loss_result)
    assert td.loss_result < ref + 1e-6, (ref, td.loss_result)
    # without the constraint the same template does better (the constraint was binding)
    basis.remove_constraint()
    td_free = TemplateOptimizer(basis, BasicCost(), training_restarts=R, seed=3, override_fail=True).approximate_target_U(target)
    assert td_free.loss_result < td.loss_result - 1e-6
    basis.build(2)
    basis.set_constraint(-0.1)
    with pytest.raises(ValueError):
        basis.constraint_layout(2)  # costs are >= 0 inside the bounds


def test_cost_constraint_stage_matches_the_cpu_port_item_by_item(hip_ctx):
    """slam_v2_set_constraint + slam_v2_minimize_stage from explicit start points against oracle/pqn_port.py (the same multiplier
    method in NumPy, float64 metric) item by item: every device result is feasible; most items end in the port's minimum
    (float32 metric: a few take another path to another minimum) and the best of each target's restarts agrees; the C ABI's
    argument checks."""
    from oracle import pqn_port

    basis = CircuitTemplateV2(base_gates=[RiSwapGate])
    basis.build(3)
    for name in basis.parameter_names():
        if name.startswith("Q"):
            basis.add_bound(name, 1.0, 0.0)
    basis.set_constraint(1.0)
    n_dev, idx, ilo, ihi, blo, bhi = basis.device_layout(3)
    w_dev, cmax = basis.constraint_layout(3)
    assert n_dev == 27 and np.array_equal(w_dev, np.r_[np.zeros(24), np.ones(3)]) and cmax == 1.0
    T, R = 3, 10
    targets = o.haar_batch(T, seed0=640)
    hip_ctx.set_targets(targets)
    hip_ctx.v2_set_gates(basis._gate_maps)
    with pytest.raises(_ffi.SlamHipError):
        hip_ctx.v2_set_constraint(3, w_dev[:-1], cmax)  # wrong parameter count
    with pytest.raises(_ffi.SlamHipError):
        hip_ctx.v2_set_constraint(3, np.zeros(27), cmax)  # no weight
    hip_ctx.v2_set_constraint(3, w_dev, cmax)
    rng = np.random.default_rng(8)
    x0 = np.concatenate([rng.uniform(-4 * np.pi, 4 * np.pi, (T, R, 24)), rng.uniform(0.0, 0.6, (T, R, 3))], axis=2)  # some starts infeasible
    prm = _ffi.OptParams(restarts=R, seed=1)
    out = hip_ctx.v2_minimize_stage([0, 0, 0], prm, 1e-10, ilo, ihi, blo, bhi, x0=x0)
    gm = gate_map(RiSwapGate)[1:]
    agree = 0
    for t in range(T):
        fun = lambda xx: v.loss_and_grad(xx, [gm] * 3, 1, 3, targets[t], False, False)
        port = [pqn_port.minimize_port(fun, x0[t, r], blo, bhi, w_dev, cmax)[0] for r in range(R)]
        agree += int(np.sum(np.abs(np.array(port) - out["item_loss"][t]) < 1e-6))
        assert abs(min(port) - out["best_loss"][t]) < 1e-6, (t, min(port), out["best_loss"][t])
        xb = out["best_x"][t]
        assert xb[24:].sum() <= 1.0 and np.all(xb[24:] >= 0.0) and np.all(xb[24:] <= 1.0)
        assert abs(fun(xb)[0] - out["best_loss"][t]) < 1e-12  # the plain loss, not the augmented Lagrangian
    assert agree >= 0.7 * T * R, agree
    # removing the constraint gives the bounded run again: lower losses, cost above the limit somewhere
    hip_ctx.v2_set_constraint(3, None)
    free = hip_ctx.v2_minimize_stage([0, 0, 0], prm, 1e-10, ilo, ihi, blo, bhi, x0=x0)
    assert np.all(free["best_loss"] <= out["best_loss"] + 1e-9) and np.any(free["best_x"][:, 24:].sum(axis=1) > 1.0 + 1e-6)


def test_cost_constraint_with_callback_and_sharded_devices():
    """set_constraint through the other two host paths: use_callback=True (slam_v2_minimize_stage_trace: the recorded losses are
    plain losses, the last one of the winning restart is the result) and devices=[0, 0] (every shard's context gets the
    constraint; the sharded job returns the single-device results bit for bit)."""
    def make():
        b = CircuitTemplateV2(base_gates=[RiSwapGate], maximum_span_guess=2)
        b.build(2)
        b.spanning_range = range(2, 3)
        for name in b.parameter_names():
            if name.startswith("Q"):
                b.add_bound(name, 1.0, 0.0)
        b.set_constraint(0.6)
        return b

    targets = [o.haar_unitary(s) for s in (411, 412, 413, 414)]
    one = TemplateOptimizer(make(), BasicCost(), training_restarts=6, seed=21, override_fail=True)
    one.approximate_from_distribution(GateSample(targets[0]))  # warm path
    res1 = [TemplateOptimizer(make(), BasicCost(), training_restarts=6, seed=21, override_fail=True).approximate_target_U(t) for t in targets[:1]]
    basis = make()
    both = TemplateOptimizer(basis, BasicCost(), training_restarts=6, seed=21, override_fail=True, devices=[0, 0])
    single = TemplateOptimizer(make(), BasicCost(), training_restarts=6, seed=21, override_fail=True)

    class Batch:
        def __iter__(self):
            return iter(targets)

    _, _, d2 = both.approximate_from_distribution(Batch())
    _, _, d1 = single.approximate_from_distribution(Batch())
    for a, b in zip(d1, d2):
        assert a.loss_result == b.loss_result and np.array_equal(np.asarray(a.Xk), np.asarray(b.Xk))
        basis.build(2)
        assert basis.circuit_cost(a.Xk) <= 0.6 + 1e-12 and a.loss_result > 1e-6  # too little interaction: the constraint binds
    assert res1[0].loss_result == d1[0].loss_result  # seeds keyed on the target index: the same decomposition alone or in a batch
    cb = TemplateOptimizer(make(), BasicCost(), training_restarts=6, seed=21, override_fail=True, use_callback=True)
    loss, coords, data = cb.approximate_from_distribution(GateSample(targets[0]))
    assert abs(data[0].loss_result - d1[0].loss_result) < 1e-9
    rec = [v for v in loss[0] if v >= 0]  # (-1, k) markers removed: [k, losses..., ...]
    # the record holds PLAIN losses (iterates of the multiplier method may be infeasible and lie below the feasible optimum);
    # the winning restart's last one is the result
    assert any(abs(v - data[0].loss_result) < 1e-9 for v in rec)
