"""GPU parity: the in-kernel quasi-Newton loop (HIP, through the C ABI) vs the oracle.

* Philox seeds are bit-exact against oracle.x0_philox.
* The first iterations follow oracle/bfgs_port.py (same algorithm on the CPU) closely.
* Converged results: same loss as SciPy-BFGS on the oracle (the reference's optimizer,
  src/slam/optimizer.py:270-278) from the same x0 whenever both reach the global minimum;
  best-of-restarts loss per target within 1e-6 (BASELINE.json north_star).
"""
import numpy as np
import pytest
import scipy.optimize as opt

from oracle import slam_oracle as o
from oracle.bfgs_port import minimize_port
from slam_decomposition_amd import _ffi

pytestmark = pytest.mark.gpu

CX = o.cx_matrix()
SQ = o.riswap_matrix(0.5)


def _params(restarts, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=7, flags=0):
    return _ffi.OptParams(restarts=restarts, maxiter=maxiter, gtol=gtol, stop_loss=stop_loss, seed=seed, flags=flags)


@pytest.mark.parametrize("k", [1, 2, 3])
def test_philox_x0_bit_exact(hip_ctx, k):
    targets = o.haar_batch(3, seed0=11)
    hip_ctx.set_targets(targets)
    hip_ctx.set_gates(SQ[None])
    seed = 0x1234_5678_9ABC_DEF1
    R = 5
    out = hip_ctx.minimize_stage([0] * k, _params(R, maxiter=0, seed=seed), active=np.array([2, 0], np.int32))
    # maxiter = 0: the kernel evaluates x0 once and returns it
    for slot, tgt in enumerate([2, 0]):
        losses = []
        for r in range(R):
            losses.append(o.loss(o.x0_philox(seed, tgt, r, k), [SQ] * k, targets[tgt]))
        r_best = int(np.argmin(losses))
        assert out["best_restart"][slot] == r_best
        x_ref = o.x0_philox(seed, tgt, r_best, k)
        assert np.array_equal(out["best_x"][slot], x_ref)
        assert np.allclose(out["item_loss"][slot], losses, atol=1e-13, rtol=0)
        assert np.all(out["item_evals"][slot] == 1)


@pytest.mark.parametrize("k,gate", [(1, CX), (2, SQ), (3, CX)])
def test_first_iterations_follow_cpu_port(hip_ctx, k, gate):
    targets = o.haar_batch(4, seed0=2024)
    hip_ctx.set_targets(targets)
    hip_ctx.set_gates(gate[None])
    R = 3
    x0 = np.stack([[o.x0_philox(3, t, r, k) for r in range(R)] for t in range(4)])
    for maxiter in (1, 3, 8):
        out = hip_ctx.minimize_stage([0] * k, _params(R, maxiter=maxiter), x0=x0)
        for t in range(4):
            for r in range(R):
                f, x, it, st, nev = minimize_port(x0[t, r], [gate] * k, targets[t], maxiter=maxiter)
                assert out["item_iters"][t, r] == it
                assert out["item_evals"][t, r] == nev
                # the inverse Hessian is float32 on both sides but summed in a different order
                assert abs(out["item_loss"][t, r] - f) < (1e-11 if maxiter == 1 else 1e-5)


# (target, restart) pairs of 24 whose HIP and SciPy runs end in the same minimum: what was measured on MI355X minus ONE pair (round 5;
# until round 4 the bar was 75 % = 18 for every case).  Pairs that differ sit in different local minima of a non-zero landscape (k below
# the span) or took another path to another zero.
MIN_AGREE = {("cx", 3): 23, ("sqiswap", 3): 23, ("cx", 2): 23, ("b", 2): 23}  # measured: 24 of 24 in all four cases


@pytest.mark.parametrize(
    "name,gate,k,expect_success",
    [("cx", CX, 3, True), ("sqiswap", SQ, 3, True), ("cx", CX, 2, False), ("b", o.berkeley_matrix(), 2, True)],
)
def test_converged_loss_matches_scipy_bfgs(hip_ctx, name, gate, k, expect_success):
    """Same x0 -> HIP quasi-Newton and SciPy BFGS (analytic jac on the oracle) end in the same
    minimum.  Trajectories differ (different line search), so a pair may land in different local
    minima; require agreement for >= 75 % of pairs and for every best-of-restarts value."""
    N, R = 6, 4
    targets = o.haar_batch(N, seed0=31337)
    hip_ctx.set_targets(targets)
    hip_ctx.set_gates(gate[None])
    x0 = np.stack([[o.x0_philox(9, t, r, k) for r in range(R)] for t in range(N)])
    out = hip_ctx.minimize_stage([0] * k, _params(R), x0=x0)
    ref = np.empty((N, R))
    for t in range(N):
        for r in range(R):
            res = opt.minimize(
                lambda xx: o.loss_and_grad(xx, [gate] * k, targets[t]), x0[t, r], jac=True, method="BFGS",
                options={"maxiter": 2500, "gtol": 1e-9},
            )
            ref[t, r] = res.fun
    agree = np.abs(out["item_loss"] - ref) < 1e-6
    print(f"AGREE {name} k={k}: {int(agree.sum())} of {agree.size}")
    assert int(agree.sum()) >= MIN_AGREE[(name, k)], (name, k, int(agree.sum()))
    assert np.all(np.abs(out["best_loss"] - ref.min(axis=1)) < 1e-6)
    # returned best_x really has the returned loss (re-evaluated by the oracle)
    for t in range(N):
        assert abs(o.loss(out["best_x"][t], [gate] * k, targets[t]) - out["best_loss"][t]) < 1e-12
    if expect_success:
        assert np.all(out["best_loss"] < 1e-10)
        # Weyl coordinates of the found circuit match the target's to 1e-6 (north_star)
        for t in range(N):
            W = o.template_eval(out["best_x"][t], [gate] * k)
            assert np.max(np.abs(o.c1c2c3_raw(W) - o.c1c2c3_raw(targets[t]))) < 1e-6
    else:
        assert np.all(out["best_loss"] > 1e-6)
    assert np.all(np.isin(out["item_status"], [0, 4]))
    assert np.all(out["item_evals"] >= out["item_iters"] + 1)


@pytest.mark.parametrize("k", [4, 5])
def test_long_spans_converge_like_scipy(hip_ctx, k):
    """Spans 4 and 5 (untuned instantiations of the same kernel): same minima as SciPy BFGS."""
    N, R = 3, 4
    targets = o.haar_batch(N, seed0=4100 + k)
    hip_ctx.set_targets(targets)
    hip_ctx.set_gates(SQ[None])
    x0 = np.stack([[o.x0_philox(13, t, r, k) for r in range(R)] for t in range(N)])
    out = hip_ctx.minimize_stage([0] * k, _params(R), x0=x0)
    assert np.all(out["best_loss"] < 1e-12)
    for t in range(N):
        assert abs(o.loss(out["best_x"][t], [SQ] * k, targets[t]) - out["best_loss"][t]) < 1e-12
        res = opt.minimize(lambda xx: o.loss_and_grad(xx, [SQ] * k, targets[t]), x0[t, 0], jac=True, method="BFGS",
                           options={"gtol": 1e-9})
        assert res.fun < 1e-10


def test_early_exit_preempts_siblings(hip_ctx):
    N, R, k = 8, 16, 3
    targets = o.haar_batch(N, seed0=5150)
    hip_ctx.set_targets(targets)
    hip_ctx.set_gates(SQ[None])
    full = hip_ctx.minimize_stage([0] * k, _params(R, seed=21))
    early = hip_ctx.minimize_stage([0] * k, _params(R, seed=21, flags=_ffi.FLAG_EARLY_EXIT))
    assert np.all(full["best_loss"] < 1e-12)
    assert np.all(early["best_loss"] < 1e-12)
    assert np.any(early["item_status"] == _ffi.ST_PREEMPTED)
    # pre-empted restarts did no more work than in the full run
    assert early["item_evals"].sum() < full["item_evals"].sum()
    # exactly-converged restarts are a subset of the full run's
    conv = early["item_status"] == 0
    assert np.all(np.abs(early["item_loss"][conv] - full["item_loss"][conv]) < 1e-12)


def test_decompose_span_loop_matches_oracle_cycles(hip_ctx):
    """Whole span loop (k = 1..3) on sqrt(iSWAP): best_cycles must equal the analytic rule
    |z| <= x - y (reference utils/transpiler_pass/weyl_decompose.py:348) and the losses of the
    solved targets must be < threshold; compare with the oracle's run_reference (SciPy, analytic
    jac) on a subset."""
    N, R = 32, 16
    targets = o.haar_batch(N, seed0=o.BENCH_TARGET_SEED0)
    hip_ctx.set_targets(targets)
    hip_ctx.set_gates(SQ[None])
    prm = _params(R, seed=1)
    best_loss, best_x, best_cycles = hip_ctx.decompose(1, 3, [[0], [0, 0], [0, 0, 0]], prm, 1e-10)
    assert np.all(best_loss < 1e-10)
    expect = []
    for t in range(N):
        c1, c2, c3 = o.c1c2c3_raw(targets[t])
        # mirror into the x >= y >= |z| chamber used by the sqrt(iSWAP) rule
        if c1 > 0.5:
            c1, c3 = 1 - c1, -c3
        expect.append(2 if abs(c3) <= c1 - c2 + 1e-9 else 3)
    assert np.array_equal(best_cycles, np.array(expect))
    for t in range(0, N, 8):
        k = best_cycles[t]
        W = o.template_eval(best_x[t, : 6 * (k + 1)], [SQ] * k)
        assert abs(o.basic_cost(W, targets[t]) - best_loss[t]) < 1e-12
        assert np.max(np.abs(o.c1c2c3_raw(W) - o.c1c2c3_raw(targets[t]))) < 1e-6
        ref_loss, _, ref_k, _ = o.run_reference(
            targets[t], [SQ], range(1, 4), R, 1e-8, x0_fn=lambda kk, r, t=t: o.x0_philox(1, t, r, kk), analytic_jac=True
        )  # 1e-8: SciPy's default gtol = 1e-5 stops at loss ~ 1e-10 (BASELINE.json metric: loss < 1e-8)
        assert ref_k == k
        assert abs(ref_loss - best_loss[t]) < 1e-6


@pytest.mark.parametrize(
    "name,gate,k",
    [
        ("cx", CX, 1), ("cx", CX, 2), ("sqiswap", SQ, 1), ("sqiswap", SQ, 3), ("b", o.berkeley_matrix(), 2),
        ("cg-phases", o.conversion_gain_matrix(0.3, -0.7, 0.9, 0.4, 1.0), 2),  # X-shaped, complex blocks (GC_XGEN)
        ("dense", o.haar_unitary(31337), 2),  # no structure (GC_DENSE)
    ],
)
def test_full_runs_follow_cpu_port(hip_ctx, name, gate, k):
    """Whole optimizer runs (default tolerances, incl. the step-growth rule) against the NumPy port of the
    same iteration, item by item: same converged loss, and -- rounding of the float32 inverse Hessian
    aside -- the same number of evaluations."""
    n_t, R = 6, 4
    targets = o.haar_batch(n_t, seed0=777)
    hip_ctx.set_targets(targets)
    hip_ctx.set_gates(gate[None])
    out = hip_ctx.minimize_stage([0] * k, _params(R, seed=11))
    same_evals = 0
    for t in range(n_t):
        for r in range(R):
            f, x, it, st, nev = minimize_port(o.x0_philox(11, t, r, k), [gate] * k, targets[t])
            assert out["item_status"][t, r] in (0, 4) and st in (0, 4)
            assert abs(out["item_loss"][t, r] - f) < 1e-6, (name, k, t, r, out["item_loss"][t, r], f)
            same_evals += int(out["item_evals"][t, r] == nev)
            assert abs(int(out["item_evals"][t, r]) - nev) <= max(8, nev // 4), (name, k, t, r, out["item_evals"][t, r], nev)
    assert same_evals >= (n_t * R) // 2, (name, k, same_evals)
