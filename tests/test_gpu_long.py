"""GPU, round 5: templates of 6 .. 16 two-qubit gates -- one wavefront per item, layers spread over the quads,
prefix / suffix products by a parallel scan, L-BFGS state in LDS (csrc/slam_long.hpp).  The reference's own use of
MixedOrderBasisCircuitTemplate (src/slam/basis.py:213-359, scripts/haar_improvements.ipynb) builds such circuits from weak gates.

  * loss, gradient and template unitary against the oracle (CircuitTemplate.eval + BasicCost / SquareCost) to 1e-12;
  * the optimizer against SciPy BFGS on the oracle from the same start points;
  * MixedOrderBasisCircuitTemplate(maximum_span_guess=12) through TemplateOptimizer: every target solved at the size its coverage
    region predicts, and not one gate earlier (the brute-force size).
"""
import numpy as np
import pytest
import scipy.optimize as opt

from oracle import slam_oracle as o
from slam_decomposition_amd import _ffi
from slam_decomposition_amd.basis import CircuitTemplate, MixedOrderBasisCircuitTemplate
from slam_decomposition_amd.cost_function import BasicCost, SquareCost
from slam_decomposition_amd.gates import ConversionGainGate, RiSwapGate
from slam_decomposition_amd.optimizer import TemplateOptimizer

pytestmark = pytest.mark.gpu

SQ = o.riswap_matrix(0.5)
WEAK = ConversionGainGate(0, 0, 0.0, np.pi / 16, 1.0).to_matrix()


@pytest.mark.parametrize("k", [6, 7, 11, 15, 16])
@pytest.mark.parametrize("cost", [0, 1])
def test_long_template_loss_gradient_unitary_match_the_oracle(hip_ctx, k, cost):
    """slam_eval_loss_grad / slam_eval_unitary beyond the register-resident spans: a mixed gate sequence (a dense Haar gate, an
    X-shaped gate with complex blocks, sqrt(iSWAP), CX), ragged item counts, both cost functions."""
    gates = np.stack([o.haar_unitary(9), o.conversion_gain_matrix(0.3, -0.7, 0.9, 0.4, 1.0), SQ, o.cx_matrix()])
    seq = [(3 * j + 1) % 4 for j in range(k)]
    T = o.haar_batch(5, seed0=800 + k)
    hip_ctx.set_targets(T)
    hip_ctx.set_gates(gates)
    hip_ctx.set_cost(cost)
    try:
        rng = np.random.default_rng(k)
        M = 37
        X = rng.uniform(-2 * np.pi, 2 * np.pi, (M, 6 * (k + 1)))
        tof = rng.integers(0, 5, M).astype(np.int32)
        loss, grad = hip_ctx.eval_loss_grad(seq, X, tof)
        W, _ = hip_ctx.eval_unitary(seq, X, tof)
        gs = [gates[i] for i in seq]
        for m in range(M):
            if cost == 0:
                f, g = o.loss_and_grad(X[m], gs, T[tof[m]])
            else:
                f, g = o.square_loss_and_grad(X[m], gs, T[tof[m]])
            assert abs(loss[m] - f) < 1e-12, (k, m, loss[m], f)
            assert np.max(np.abs(grad[m] - g)) < 1e-12, (k, m, np.max(np.abs(grad[m] - g)))
            assert np.max(np.abs(W[m] - o.template_eval(X[m], gs))) < 1e-12
    finally:
        hip_ctx.set_cost(0)


@pytest.mark.parametrize("k,gate", [(6, SQ), (8, WEAK)])
def test_long_template_optimizer_matches_scipy(hip_ctx, k, gate):
    """slam_minimize_stage at 6 and 8 gates from explicit start points against SciPy BFGS (analytic gradient) on the oracle: the same
    minimum for most (target, restart) pairs, the same best-of-restarts loss for every target, returned parameters reproduce the
    returned loss; ordered early exit drops later restarts only."""
    N, R = 4, 6
    rng = np.random.default_rng(40 + k)
    # two targets the template certainly reaches (itself at random angles), two Haar targets
    T = np.stack([o.template_eval(rng.uniform(0, 2 * np.pi, 6 * (k + 1)), [gate] * k) for _ in range(2)] + [o.haar_unitary(70 + i) for i in range(N - 2)])
    hip_ctx.set_targets(T)
    hip_ctx.set_gates(gate[None])
    x0 = rng.uniform(0, 2 * np.pi, (N, R, 6 * (k + 1)))
    prm = _ffi.OptParams(restarts=R, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=5)
    out = hip_ctx.minimize_stage([0] * k, prm, x0=x0)
    ref = np.empty((N, R))
    for t in range(N):
        assert abs(o.loss(out["best_x"][t], [gate] * k, T[t]) - out["best_loss"][t]) < 1e-12
        for r in range(R):
            ref[t, r] = opt.minimize(lambda xx: o.loss_and_grad(xx, [gate] * k, T[t]), x0[t, r], jac=True, method="BFGS",
                                     options={"maxiter": 2500, "gtol": 1e-9}).fun
    assert (np.abs(out["item_loss"] - ref) < 1e-6).mean() >= 0.6, (out["item_loss"], ref)
    assert np.all(np.abs(out["best_loss"] - ref.min(axis=1)) < 1e-6), (out["best_loss"], ref.min(axis=1))
    assert np.all(out["best_loss"][:2] < 1e-10)
    assert np.all(np.isin(out["item_status"], [0, 4])) and np.all(out["item_evals"] >= out["item_iters"] + 1)
    early = hip_ctx.minimize_stage([0] * k, _ffi.OptParams(restarts=R, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=5,
                                                           flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED), x0=x0)
    assert np.all(early["best_loss"][:2] < 1e-10) and early["item_evals"].sum() <= out["item_evals"].sum()
    for t in range(N):
        hit = np.nonzero(out["item_loss"][t] < 1e-13)[0]
        if len(hit):  # the ordered winner is the lowest-index successful restart of the full run
            assert early["best_restart"][t] == hit[0] and early["best_loss"][t] == out["item_loss"][t, hit[0]]


def test_long_template_philox_start_points_and_span_loop(hip_ctx):
    """The span loop running from the quad kernels into the wavefront-per-item kernels (k = 4 .. 7) on sqrt(iSWAP): Philox start points
    bit-equal with the oracle's at k = 6 (maxiter = 0 returns x0), every Haar target solved at k <= 3 keeps its answer when the loop is
    asked for k = 1 .. 7 (later spans are never reached)."""
    T = o.haar_batch(3, seed0=4242)
    hip_ctx.set_targets(T)
    hip_ctx.set_gates(SQ[None])
    seed = 0x5EED1234
    out = hip_ctx.minimize_stage([0] * 6, _ffi.OptParams(restarts=3, maxiter=0, seed=seed))
    for t in range(3):
        losses = [o.loss(o.x0_philox(seed, t, r, 6), [SQ] * 6, T[t]) for r in range(3)]
        rb = int(np.argmin(losses))
        assert out["best_restart"][t] == rb and np.array_equal(out["best_x"][t], o.x0_philox(seed, t, rb, 6))
        assert np.allclose(out["item_loss"][t], losses, atol=1e-13, rtol=0)
    prm = _ffi.OptParams(restarts=8, seed=3, flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED)
    l3, x3, c3 = hip_ctx.decompose(1, 3, [[0] * k for k in (1, 2, 3)], prm, 1e-10)
    l7, x7, c7 = hip_ctx.decompose(1, 7, [[0] * k for k in range(1, 8)], prm, 1e-10)
    assert np.array_equal(c3, c7) and np.array_equal(l3, l7) and x7.shape[1] == 48
    for t in range(3):
        w = 6 * (int(c3[t]) + 1)
        assert np.array_equal(x3[t, :w], x7[t, :w])
    # spans 6..7 alone: solved again (more gates than needed), the result's row has 6 (k + 1) parameters
    l, x, c = hip_ctx.decompose(6, 7, [[0] * 6, [0] * 7], prm, 1e-10)
    assert np.all(l < 1e-10) and np.all(c == 6)
    for t in range(3):
        assert abs(o.loss(x[t, :42], [SQ] * 6, T[t]) - l[t]) < 1e-12


def test_mixed_order_template_of_twelve_weak_gates_through_the_api():
    """MixedOrderBasisCircuitTemplate with ONE weak gate -- ConversionGainGate(gain pi/16): (1/16, 1/16, 0) in the Weyl chamber -- and
    maximum_span_guess = 12: Haar targets need 5 .. 10 applications (coverage.py).  Through TemplateOptimizer every target is solved
    at exactly the size its coverage region predicts; one gate less is NOT enough (brute force with the same restarts: the predicted
    size is the brute-force size)."""
    gate = ConversionGainGate(0, 0, 0.0, np.pi / 16, 1.0)
    basis = MixedOrderBasisCircuitTemplate([gate], maximum_span_guess=12)
    assert len(basis.coverage) == 12 and basis.span_rules_exact
    T = [o.haar_unitary(900 + i) for i in range(6)]
    from slam_decomposition_amd.weyl import c1c2c3

    coords = np.array([c1c2c3(t) for t in T])
    want = basis.minimal_spans(coords)
    assert want.min() >= 5 and want.max() <= 11

    class S:
        def __iter__(self):
            return iter(T)

    optm = TemplateOptimizer(basis, BasicCost(), training_restarts=24, seed=17, override_fail=True)
    loss, _, data = optm.approximate_from_distribution(S())
    G = gate.to_matrix() if hasattr(gate, "to_matrix") else np.asarray(gate)
    G = basis.gate_matrices[0]
    for t, td in enumerate(data):
        assert td.cycles == want[t], (t, td.cycles, want[t], td.loss_result)
        assert td.success_label == 1 and td.loss_result <= 1e-10
        assert len(td.Xk) == 6 * (td.cycles + 1)
        W = o.template_eval(np.asarray(td.Xk), [G] * td.cycles)
        assert abs(o.basic_cost(W, T[t]) - td.loss_result) < 1e-12
        assert np.max(np.abs(o.c1c2c3_raw(W) - o.c1c2c3_raw(T[t]))) < 1e-6
    # brute force one gate earlier: out of reach
    with _ffi.Context(0) as ctx:
        ctx.set_targets(np.stack(T))
        ctx.set_gates(G[None])
        prm = _ffi.OptParams(restarts=24, seed=17, flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED)
        for k in np.unique(want):
            sel = np.nonzero(want == k)[0].astype(np.int32)
            out = ctx.minimize_stage([0] * (int(k) - 1), prm, active=sel, want_items=False)
            assert np.all(out["best_loss"] > 1e-8), (k, out["best_loss"])


def test_sixteen_gates_optimizer_and_pinned_exterior(hip_ctx):
    """The two-pass case (17 layers: quad 0 owns layers 0 and 16) through the optimizer, and SLAM_FLAG_NO_EXTERIOR on the long kernels:
    a 16-gate template of the weak gate reaches targets built from it (SciPy BFGS on the oracle reaches the same loss from the same
    start); with the exterior layers pinned the result keeps them at exactly zero and equals SciPy on the reduced function."""
    k = 16
    rng = np.random.default_rng(2016)
    T = np.stack([o.template_eval(rng.uniform(0, 2 * np.pi, 6 * (k + 1)), [WEAK] * k) for _ in range(2)])
    hip_ctx.set_targets(T)
    hip_ctx.set_gates(WEAK[None])
    R = 3
    x0 = rng.uniform(0, 2 * np.pi, (2, R, 6 * (k + 1)))
    out = hip_ctx.minimize_stage([0] * k, _ffi.OptParams(restarts=R, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=1), x0=x0)
    assert np.all(out["best_loss"] < 1e-10)
    for t in range(2):
        assert abs(o.loss(out["best_x"][t], [WEAK] * k, T[t]) - out["best_loss"][t]) < 1e-12
        ref = opt.minimize(lambda xx: o.loss_and_grad(xx, [WEAK] * k, T[t]), x0[t, 0], jac=True, method="BFGS", options={"maxiter": 2500, "gtol": 1e-9}).fun
        assert ref < 1e-10
    # pinned exterior layers at k = 6: targets inside the reduced template's reach
    k = 6

    def pad(xr):
        x = np.zeros(6 * (k + 1))
        x[6 : 6 * k] = xr
        return x

    T = np.stack([o.template_eval(pad(rng.uniform(0, 2 * np.pi, 6 * (k - 1))), [SQ] * k) for _ in range(3)])
    hip_ctx.set_targets(T)
    hip_ctx.set_gates(SQ[None])
    x0 = rng.uniform(0, 2 * np.pi, (3, 4, 6 * (k + 1)))
    out = hip_ctx.minimize_stage([0] * k, _ffi.OptParams(restarts=4, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=1, flags=_ffi.FLAG_NO_EXTERIOR), x0=x0)
    assert np.all(out["best_x"][:, :6] == 0.0) and np.all(out["best_x"][:, 6 * k :] == 0.0)
    assert np.all(out["best_loss"] < 1e-10)
    for t in range(3):
        assert abs(o.loss(out["best_x"][t], [SQ] * k, T[t]) - out["best_loss"][t]) < 1e-12


def test_predicted_span_loop_crossing_into_the_long_kernels():
    """slam_decompose_predicted with k_max = 7 on the pi/16 gain gate: the targets of sizes 4 .. 7 join the loop at their stage -- quad
    kernels up to five gates, wavefront-per-item kernels beyond -- and everything the host-side lookup assigns a size <= 7 is solved
    at that size (exact regions, carry = 0); bigger targets come back as out of reach."""
    from slam_decomposition_amd import coverage
    from slam_decomposition_amd.weyl import c1c2c3

    n, R, kmax = 400, 12, 7
    coords = [c1c2c3(WEAK)] * kmax
    with _ffi.Context(0) as ctx:
        ctx.sample_haar(8080, n)
        ctx.set_gates(WEAK[None])
        tc = ctx.targets_c1c2c3(0, n)
        want = coverage.minimal_prefix(tc, coords, kmax)
        prm = _ffi.OptParams(restarts=R, seed=4, flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED)
        n_loc, n_unr = ctx.decompose_predicted(coords, kmax, [[0] * k for k in range(1, kmax + 1)], prm, 1e-10, 0, n)
        loss, x, cyc = ctx.fetch_results_range(kmax, 0, n)
    assert n_loc == 0 and n_unr == int((want > kmax).sum()) and 0 < n_unr < n
    ran = want <= kmax
    assert np.array_equal(cyc[ran], want[ran]) and np.all(cyc[~ran] == -1)
    assert np.mean(loss[ran] < 1e-10) > 0.97  # (a few targets next to a region's face miss it with 12 restarts)
    for t in np.nonzero(ran & (loss < 1e-10))[0][::9]:
        kk = int(cyc[t])
        assert abs(o.loss(x[t, : 6 * (kk + 1)], [WEAK] * kk, DeviceTargets.get(8080, t)) - loss[t]) < 1e-12


class DeviceTargets:
    """The device sampler's targets on the host (oracle port of slam_sample_haar)."""

    @staticmethod
    def get(seed, i):
        return o.haar_philox_port(seed, i)


@pytest.mark.parametrize("k,gate", [(6, SQ), (7, WEAK)])
def test_long_template_runs_follow_the_cpu_port(hip_ctx, k, gate):
    """The wavefront-per-item kernels run the SAME quasi-Newton iteration as the quad kernels (fp32 metric, Armijo backtracking, cautious
    update, scaling of the first metric, periodic restart): against oracle/bfgs_port.py item by item -- same converged loss, and, rounding
    of the float32 metric aside, the same number of evaluations."""
    from oracle.bfgs_port import minimize_port

    n_t, R = 4, 3
    T = o.haar_batch(n_t, seed0=1300 + k)
    hip_ctx.set_targets(T)
    hip_ctx.set_gates(gate[None])
    out = hip_ctx.minimize_stage([0] * k, _ffi.OptParams(restarts=R, seed=21))
    same = 0
    for t in range(n_t):
        for r in range(R):
            f, x, it, st, nev = minimize_port(o.x0_philox(21, t, r, k), [gate] * k, T[t])
            assert out["item_status"][t, r] in (0, 4) and st in (0, 4)
            assert abs(out["item_loss"][t, r] - f) < 1e-6, (k, t, r, out["item_loss"][t, r], f)
            same += int(out["item_evals"][t, r] == nev)
            assert abs(int(out["item_evals"][t, r]) - nev) <= max(10, nev // 3), (k, t, r, out["item_evals"][t, r], nev)
    assert same >= (n_t * R) // 3, same


def test_per_iteration_traces_of_a_long_template(hip_ctx):
    """use_callback for templates beyond five gates (src/slam/optimizer.py:217-224): the wavefront-per-item kernels record loss and point
    after every accepted step like the quad kernels -- the recorded losses are the oracle's at the recorded points, decrease, end at the
    item's result; through the API the reference's training_loss layout [-1, k, losses ...] comes out for a 6-gate template."""
    k, R, cap = 6, 2, 400
    T = o.haar_batch(3, seed0=4600)
    hip_ctx.set_targets(T)
    hip_ctx.set_gates(SQ[None])
    prm = _ffi.OptParams(restarts=R, seed=9)
    out = hip_ctx.minimize_stage_trace([0] * k, prm, 1e-10, cap)
    ref = hip_ctx.minimize_stage([0] * k, prm)  # the same launch without a trace
    assert np.array_equal(out["item_loss"], ref["item_loss"]) and np.array_equal(out["item_iters"], ref["item_iters"])
    n = 6 * (k + 1)
    for t in range(3):
        for r in range(R):
            it = int(out["item_iters"][t, r])
            assert 0 < it <= cap
            tl = out["trace_loss"][t, r]
            assert np.all(np.isfinite(tl[:it])) and np.all(np.isnan(tl[it:]))
            assert np.all(np.diff(tl[:it]) <= 1e-15)  # accepted steps only: Armijo decrease
            assert tl[it - 1] == out["item_loss"][t, r]
            for j in (0, it // 2, it - 1):
                x = out["trace_x"][t, r, j]
                assert x.shape == (n,) and abs(o.basic_cost(o.template_eval(x, [SQ] * k), T[t]) - tl[j]) < 1e-12
            assert np.all(np.isnan(out["trace_x"][t, r, it:]))
    # the API: one target, the callback's bookkeeping
    basis = CircuitTemplate(base_gates=[RiSwapGate(0.25)], maximum_span_guess=6)
    basis.spanning_range = range(6, 7)
    optm = TemplateOptimizer(basis, BasicCost(), use_callback=True, override_fail=True, training_restarts=2, seed=3)
    data = optm.approximate_target_U(o.haar_unitary(77))
    loss = list(optm.training_loss[0])  # (one list per target, optimizer.py:307-311)
    assert loss[0] == -1 and loss[1] == 6 and len(loss) > 10 and data.cycles == 6
    assert len(optm.coordinate_list) > 0
