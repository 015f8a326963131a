"""GPU, round 5:
  * ``approximate_from_distribution`` on a sampler larger than one window: successive windows in flight, results bit for bit
    those of the single call (src/slam/optimizer.py:180-186);
  * ``CircuitTemplate(no_exterior_1q=True)`` on the fixed-gate path (src/slam/basis.py:57,154,165): SLAM_FLAG_NO_EXTERIOR pins
    layers 0 and k, the run is the 6 (k - 1)-parameter problem -- against SciPy BFGS on the oracle's reduced function;
  * the HIP path against the REFERENCE-FAITHFUL path (SciPy BFGS + finite differences, sequential restarts) for the bases of
    BASELINE configs[3] / configs[4] and a CircuitTemplateV2 case -- fixture tests/golden/fd_reference_r5.npz
    (tools/make_fd_reference_r5.py);
  * SLAM_FLAG_OVERLAP only where the batch's coverage says the last span is needed (a basis that solves at k = 2 runs staged).
"""
import os

import numpy as np
import pytest
import scipy.optimize as opt

from oracle import slam_oracle as o
from slam_decomposition_amd import _ffi, span_rules
from slam_decomposition_amd import gates as G
from slam_decomposition_amd.basis import CircuitTemplate
from slam_decomposition_amd.cost_function import BasicCost
from slam_decomposition_amd.optimizer import TemplateOptimizer
from slam_decomposition_amd.sampler import DeviceHaarBatch, GateSample

pytestmark = pytest.mark.gpu

GOLDEN5 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fd_reference_r5.npz")
SQ = o.riswap_matrix(0.5)


def _fold(c):
    return span_rules._fold(c)


# ------------------------------------------------------------------------------------------------------------------
# item 2: windows in flight
# ------------------------------------------------------------------------------------------------------------------
def test_windows_in_flight_equal_the_single_call():
    """A sampler of several windows through ``approximate_from_distribution``: contiguous shares on helper contexts, each run as
    windows of at most WINDOW_TARGETS targets, several in flight; ``target_data`` (lazy, per-window blocks) equals the single call's entry by entry, bit for bit -- seeds are
    keyed on the global target index and the ordered early exit makes a target's result independent of its neighbours."""
    n, R = 5000, 8
    basis = CircuitTemplate(base_gates=[G.RiSwapGate(0.5)], maximum_span_guess=3)

    def run(window, in_flight):
        optm = TemplateOptimizer(basis, BasicCost(), training_restarts=R, seed=77, override_fail=True, windows_in_flight=in_flight)
        optm.WINDOW_TARGETS = window
        loss, _, data = optm.approximate_from_distribution(DeviceHaarBatch(seed=4242, n_samples=n))
        return np.asarray(loss), data, optm

    l1, d1, o1 = run(1 << 30, 1)          # one call
    l2, d2, o2 = run(1024, 4)             # four helpers, two windows of 625 each
    l3, d3, o3 = run(1999, 3)             # three helpers with shares 1667 / 1667 / 1666, one window each
    assert np.array_equal(l1, l2) and np.array_equal(l1, l3)
    assert len(d2) == n and len(d3) == n
    for i in list(range(0, n, 97)) + [624, 625, 1249, 1250, 1666, 1667, 3333, 3334, n - 1]:
        for d in (d2, d3):
            assert d[i].cycles == d1[i].cycles and d[i].loss_result == d1[i].loss_result and d[i].success_label == d1[i].success_label
            assert np.array_equal(np.asarray(d[i].Xk), np.asarray(d1[i].Xk))
    assert o2.best_cycle_list == o1.best_cycle_list
    # stats are summed over the windows
    assert o2.last_stats["evals"][1] == o1.last_stats["evals"][1] and len(o2.last_stats_per_device) == 8 and len(o3.last_stats_per_device) == 3
    assert np.all(l1 < 1e-10)


def test_windows_with_host_targets_and_logging(caplog):
    """Host-side targets (a plain sampler) take the same windowed path; with INFO logging on the blocks are joined and the per-target
    log lines come out as for the single call."""
    import logging

    n, R = 300, 6
    T = o.haar_batch(n, seed0=9100)
    basis = CircuitTemplate(base_gates=[G.CXGate()], maximum_span_guess=3)

    class ListSampler:
        def __iter__(self):
            return iter(T)

    a = TemplateOptimizer(basis, BasicCost(), training_restarts=R, seed=5, override_fail=True)
    la, _, da = a.approximate_from_distribution(ListSampler())
    b = TemplateOptimizer(basis, BasicCost(), training_restarts=R, seed=5, override_fail=True, windows_in_flight=3)
    b.WINDOW_TARGETS = 128
    with caplog.at_level(logging.INFO):
        lb, _, db = b.approximate_from_distribution(ListSampler())
    assert la == lb and len(db) == n
    assert all(np.array_equal(np.asarray(x.Xk), np.asarray(y.Xk)) for x, y in zip(da, db))
    assert sum(1 for r in caplog.records if r.getMessage().startswith("Starting sample iter")) == n


# ------------------------------------------------------------------------------------------------------------------
# item 6: no_exterior_1q on the fixed-gate template
# ------------------------------------------------------------------------------------------------------------------
def _pad(xr, k):
    x = np.zeros(6 * (k + 1))
    x[6 : 6 * k] = xr
    return x


@pytest.mark.parametrize("k,gate", [(2, SQ), (3, SQ), (3, o.cx_matrix())])
def test_no_exterior_stage_is_the_reduced_problem(hip_ctx, k, gate):
    """SLAM_FLAG_NO_EXTERIOR through slam_minimize_stage: the exterior parameters of every result are exactly zero, the loss is
    the oracle's for the padded vector, and from the same interior start values SciPy BFGS on the REDUCED function
    xr -> loss(pad(xr)) ends in the same minimum for most items and for every best-of-restarts value."""
    N, R = 5, 6
    rng = np.random.default_rng(100 + k)
    # targets inside the reduced template's reach (the template itself at random interior angles) and generic ones
    T = np.stack([o.template_eval(_pad(rng.uniform(0, 2 * np.pi, 6 * (k - 1)), k), [gate] * k) for _ in range(3)] + [o.haar_unitary(50 + i) for i in range(N - 3)])
    hip_ctx.set_targets(T)
    hip_ctx.set_gates(gate[None])
    x0 = rng.uniform(0, 2 * np.pi, (N, R, 6 * (k + 1)))
    prm = _ffi.OptParams(restarts=R, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=3, flags=_ffi.FLAG_NO_EXTERIOR)
    out = hip_ctx.minimize_stage([0] * k, prm, x0=x0)
    assert np.all(out["best_x"][:, :6] == 0.0) and np.all(out["best_x"][:, 6 * k :] == 0.0)
    ref = np.empty((N, R))
    for t in range(N):
        assert abs(o.loss(out["best_x"][t], [gate] * k, T[t]) - out["best_loss"][t]) < 1e-12
        for r in range(R):
            def fun(xr, t=t):
                f, g = o.loss_and_grad(_pad(xr, k), [gate] * k, T[t])
                return f, g[6 : 6 * k]
            ref[t, r] = opt.minimize(fun, x0[t, r, 6 : 6 * k], jac=True, method="BFGS", options={"maxiter": 2500, "gtol": 1e-9}).fun
    assert (np.abs(out["item_loss"] - ref) < 1e-6).mean() >= 0.7
    assert np.all(np.abs(out["best_loss"] - ref.min(axis=1)) < 1e-6)
    assert np.all(out["best_loss"][:3] < 1e-10)  # the reachable targets are reached
    # the same items WITHOUT the flag move the exterior parameters (the flag is what pins them)
    free = hip_ctx.minimize_stage([0] * k, _ffi.OptParams(restarts=R, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=3), x0=x0)
    assert np.any(free["best_x"][:, :6] != 0.0)


def test_no_exterior_template_through_the_api():
    """CircuitTemplate(no_exterior_1q=True) (basis.py:57,154,165) end to end: ``Xk`` has 6 (cycles - 1) entries, ``eval`` /
    ``to_gate_list`` take them, the recorded loss is the oracle's for G_k K_{k-1} ... K_1 G_1, reachable targets are solved at
    the size they were built with, and Philox start points leave the exterior layers at the identity."""
    basis = CircuitTemplate(base_gates=[G.RiSwapGate(0.5)], no_exterior_1q=True, maximum_span_guess=3)
    rng = np.random.default_rng(8)
    Ts, sizes = [], []
    for k in (2, 3, 3, 2):
        Ts.append(o.template_eval(_pad(rng.uniform(0, 2 * np.pi, 6 * (k - 1)), k), [SQ] * k))
        sizes.append(k)
    Ts.append(o.haar_unitary(3))  # generic: out of reach without exterior gates
    optm = TemplateOptimizer(basis, BasicCost(), training_restarts=24, seed=31, override_fail=True)

    class S:
        def __iter__(self):
            return iter(Ts)

    loss, _, data = optm.approximate_from_distribution(S())
    for t, td in enumerate(data):
        k = td.cycles
        assert len(td.Xk) == 6 * (k - 1)
        W = o.template_eval(_pad(np.asarray(td.Xk), k), [SQ] * k)
        assert abs(o.basic_cost(W, Ts[t]) - td.loss_result) < 1e-12
        basis.build(k)
        assert basis.n_params == 6 * (k - 1)
        assert np.max(np.abs(basis.eval(td.Xk) - W)) < 1e-13
        assert len(basis.to_gate_list(td.Xk)) == k + 2 * (k - 1)
        if t < 4:
            assert td.success_label == 1 and td.loss_result <= 1e-10 and k <= sizes[t]
    assert data[4].success_label == 0 and data[4].loss_result > 1e-6
    with pytest.raises(NotImplementedError):
        TemplateOptimizer(basis, BasicCost(), use_callback=True)


# ------------------------------------------------------------------------------------------------------------------
# SLAM_FLAG_OVERLAP from the batch's coverage
# ------------------------------------------------------------------------------------------------------------------
def test_overlap_flag_follows_the_coverage_of_the_batch():
    """A big blocking call runs its spans side by side only when a good share of the batch needs the last span: sqrt(iSWAP) (21 %
    of Haar targets need three gates) and CNOT (all) do, the B gate (every target at two) does not -- its k = 3 stage would be the
    most expensive one of the call and pure waste.  Results are bit-equal either way."""
    n, R = 6000, 32  # n R > 2^17: beyond the library's own choice
    for gate, want in ((G.RiSwapGate(0.5), True), (G.CXGate(), True), (G.BerkeleyGate(), False)):
        basis = CircuitTemplate(base_gates=[gate], maximum_span_guess=3)
        optm = TemplateOptimizer(basis, BasicCost(), training_restarts=R, seed=1, override_fail=True)
        ctx = __import__("slam_decomposition_amd.runtime", fromlist=["x"]).get_context(0)
        ctx.sample_haar(606, n)
        assert optm._overlap_pays(ctx, n, [1, 2, 3]) is want
        assert optm._overlap_pays(ctx, 1000, [1, 2, 3]) is False  # medium calls: the library decides
    basis = CircuitTemplate(base_gates=[G.BerkeleyGate()], maximum_span_guess=3)
    a = TemplateOptimizer(basis, BasicCost(), training_restarts=R, seed=1, override_fail=True)
    la, _, da = a.approximate_from_distribution(DeviceHaarBatch(seed=606, n_samples=n))
    assert a.last_stats["evals"][3] == 0  # staged: nobody reaches the third span
    b = TemplateOptimizer(basis, BasicCost(), training_restarts=R, seed=1, override_fail=True)
    b.OVERLAP_MIN_TOP_SHARE = 0.0  # force the side-by-side form
    lb, _, db = b.approximate_from_distribution(DeviceHaarBatch(seed=606, n_samples=n))
    assert la == lb and b.last_stats["evals"][3] > 0
    assert all(np.array_equal(np.asarray(da[i].Xk), np.asarray(db[i].Xk)) for i in range(0, n, 211))


# ------------------------------------------------------------------------------------------------------------------
# item 3: reference-faithful parity for configs[3], configs[4] and V2
# ------------------------------------------------------------------------------------------------------------------
def _gates_of(basis):
    if basis == "iswap+b":
        return np.stack([G.RiSwapGate(1.0).to_matrix(), G.BerkeleyGate().to_matrix()])
    if basis == "b":
        return np.stack([G.BerkeleyGate().to_matrix()])
    from bench import sweep_gate

    return np.stack([sweep_gate(int(basis[5:]))])


@pytest.mark.parametrize("basis", ["iswap+b", "b", "sweep0", "sweep24", "sweep64", "sweep100"])
def test_hip_path_matches_the_finite_difference_reference_path_configs34(basis):
    """The fixture holds what ``run_reference(analytic_jac=False)`` -- SciPy BFGS with its own finite differences, sequential
    restarts with early break: the reference's real path, optimizer.py:233-303 -- ends with on 64 counter-based Haar targets x 16
    restarts for the mixed sequence of BASELINE configs[3], the B gate and four bases of the configs[4] sweep (one that reaches
    nothing, one partial, two full).  Solved on both sides: EQUAL best_cycles, |loss difference| <= 1e-6, HIP coordinates within
    1e-6 of the target's.  Unsolved on the reference side: unsolved here too, and the HIP best loss is not worse than the
    reference's by more than 1e-6 (both sit in local minima of a non-zero landscape; the lower one is the better answer)."""
    ref = np.load(GOLDEN5)
    n, R, level = int(ref["n"]), int(ref["restarts"]), float(ref["level"])
    table = _gates_of(basis)
    with _ffi.Context(0) as ctx:
        ctx.sample_haar(int(ref["target_seed"]), n)
        ctx.set_gates(table)
        prm = _ffi.OptParams(restarts=R, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=int(ref["opt_seed"]), flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED)
        seqs = [[i % len(table) for i in range(k)] for k in (1, 2, 3)]
        loss, x, cyc = ctx.decompose_range(0, n, 1, 3, seqs, prm, level)
        found = np.zeros((n, 3))
        for k in np.unique(cyc):
            sel = np.nonzero(cyc == k)[0]
            found[sel] = ctx.eval_c1c2c3(seqs[int(k) - 1], np.ascontiguousarray(x[sel, : 6 * (int(k) + 1)]), ndigits=-1)
        tgt = ctx.targets_c1c2c3(0, n, ndigits=-1)
    r_loss, r_cyc = ref[f"{basis}_loss"], ref[f"{basis}_cycles"]
    r_ok, g_ok = r_loss < level, loss < level
    assert np.abs(_fold(tgt) - _fold(ref[f"{basis}_target_coords"])).max() <= 1e-9  # the device's targets ARE the fixture's
    # the analytic-gradient path may solve a target the finite-difference path gave up on at its noise floor, never the reverse
    assert np.all(g_ok[r_ok]), np.nonzero(r_ok & ~g_ok)[0]
    both = r_ok & g_ok
    if both.any():
        assert np.array_equal(cyc[both], r_cyc[both]), (np.nonzero(both & (cyc != r_cyc))[0],)
        assert np.max(np.abs(loss[both] - r_loss[both])) <= 1e-6
        d_tgt = np.abs(_fold(found[both]) - _fold(tgt[both])).max(axis=1)
        d_ref = np.abs(_fold(found[both]) - _fold(ref[f"{basis}_found_coords"][both])).max(axis=1)
        assert d_tgt.max() <= 1e-6
        assert np.all(d_ref <= 1e-6 + 4.0 * np.sqrt(r_loss[both]))
    un = ~r_ok
    if un.any():
        extra = un & g_ok  # (allowed only right at the reference's finite-difference floor)
        assert np.all(r_loss[extra] < 1e-6), (np.nonzero(extra)[0], r_loss[extra])
        both_un = un & ~g_ok
        assert np.all(loss[both_un] <= r_loss[both_un] + 1e-6), (np.nonzero(both_un & (loss > r_loss + 1e-6))[0],)
    expect_solved = {"iswap+b": n, "b": n, "sweep0": 0, "sweep64": n, "sweep100": n}.get(basis)
    if expect_solved is not None:
        assert int(g_ok.sum()) == expect_solved
    else:
        assert 0 < int(g_ok.sum()) < n  # the partial basis


def test_v2_hip_path_matches_the_finite_difference_reference_path():
    """CircuitTemplateV2(base_gates=[RiSwapGate]) -- a free alpha per gate instance --, SquareCost, spans 1..2, from the fixture's
    explicit start points: v2_oracle.run_reference (SciPy BFGS + finite differences, sequential restarts, optimizer.py:255-303)
    against the device's stages driven like the span loop.  Same template sizes, losses within 1e-6."""
    from slam_decomposition_amd.basisv2 import CircuitTemplateV2
    from slam_decomposition_amd.gates import RiSwapGate

    ref = np.load(GOLDEN5)
    n, R, level = int(ref["v2_n"]), int(ref["v2_restarts"]), float(ref["level"])
    basis = CircuitTemplateV2(base_gates=[RiSwapGate], maximum_span_guess=2)
    T = np.stack([o.haar_philox_port(int(ref["target_seed"]), i) for i in range(n)])
    best = np.full(n, np.inf)
    cyc = np.full(n, -1)
    with _ffi.Context(0) as ctx:
        ctx.set_targets(T)
        ctx.v2_set_gates(basis._gate_maps)
        ctx.set_cost(_ffi.COST_SQUARE)
        for k in (1, 2):
            act = np.nonzero(~(best < level))[0].astype(np.int32)
            if not len(act):
                break
            basis.build(k)
            n_dev, idx, ilo, ihi, blo, bhi = basis.device_layout(k)
            assert np.array_equal(idx, np.arange(n_dev))  # RiSwapGate: user order == device order (1Q angles, then the alphas)
            prm = _ffi.OptParams(restarts=R, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=1, flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED)
            out = ctx.v2_minimize_stage([0] * k, prm, level, ilo, ihi, blo, bhi, active=act, x0=np.ascontiguousarray(ref[f"v2_x0_k{k}"][act]))
            better = out["best_loss"] < best[act]
            best[act[better]] = out["best_loss"][better]
            cyc[act[better]] = k
    r_loss, r_cyc = ref["v2_loss"], ref["v2_cycles"]
    assert np.all(r_loss < level) and np.all(best < level)
    assert np.array_equal(cyc, r_cyc)
    assert np.max(np.abs(best - r_loss)) <= 1e-6


# ------------------------------------------------------------------------------------------------------------------
# item 6: the polytope mode without a host step
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("carry", [False, True])
def test_decompose_predicted_equals_the_host_driven_polytope_mode(carry):
    """slam_decompose_predicted (lookup + per-size lists + one span loop, all on the device) against the round-4 form -- predict_spans,
    ``np.nonzero`` per size, one slam_decompose_list per size -- on the mixed sequence of BASELINE configs[3] and on a weak
    conversion-gain gate that leaves targets out of reach: identical losses, cycles and parameters bit for bit (a target's result
    does not depend on its place in a list), the same local / unreachable counts."""
    from bench import sweep_gate
    from slam_decomposition_amd.weyl import c1c2c3

    n, R = 3000, 8
    for table in (np.stack([G.RiSwapGate(1.0).to_matrix(), G.BerkeleyGate().to_matrix()]), np.stack([sweep_gate(24)])):
        seqs = [[i % len(table) for i in range(k)] for k in (1, 2, 3)]
        coords = [c1c2c3(table[i]) for i in seqs[2]]
        prm = _ffi.OptParams(restarts=R, seed=9, flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED)
        tol = 5e-4 if carry else 2e-8
        with _ffi.Context(0) as a, _ffi.Context(0) as b:
            for c in (a, b):
                c.sample_haar(31415, n)
                c.set_gates(table)
            # identity-like targets cannot come out of the Haar sampler: a local one is planted through set_targets below
            lb = a.predict_spans(coords, 3, 0, n, tol=tol)
            for k in np.unique(lb):
                if 1 <= k <= 3:
                    a.decompose_list(np.nonzero(lb == k)[0], int(k), 3 if carry else int(k), seqs[int(k) - 1 : (3 if carry else int(k))], prm, 1e-10, k_layout=3)
            la, xa, ca = a.fetch_results_range(3, 0, n)
            n_loc, n_unr = b.decompose_predicted(coords, 3, seqs, prm, 1e-10, 0, n, carry=carry, tol=tol)
            lb_, xb, cb = b.fetch_results_range(3, 0, n)
            assert n_loc == int((lb == 0).sum()) and n_unr == int((lb > 3).sum())
            ran = (lb >= 1) & (lb <= 3)
            assert np.array_equal(la[ran], lb_[ran]) and np.array_equal(ca[ran], cb[ran])
            for t in np.nonzero(ran)[0][::37]:
                w = 6 * (int(ca[t]) + 1)
                assert np.array_equal(xa[t, :w], xb[t, :w])
            assert np.all(np.isinf(lb_[lb > 3])) and np.all(cb[lb > 3] == -1)
            assert b.stats()["items"][1] == int((lb == 1).sum()) * R


def test_use_polytopes_on_a_device_sampler_runs_without_a_host_list():
    """TemplateOptimizer(CircuitTemplate(use_polytopes=True)) on a DeviceHaarBatch goes through slam_decompose_predicted: same
    target_data as the host-side polytope mode (a plain sampler of the same targets)."""
    n, R = 2000, 8
    basis = CircuitTemplate(base_gates=[G.RiSwapGate(0.5)], maximum_span_guess=3, use_polytopes=True)
    s = DeviceHaarBatch(seed=777, n_samples=n)
    a = TemplateOptimizer(basis, BasicCost(), training_restarts=R, seed=2, override_fail=True)
    la, _, da = a.approximate_from_distribution(s)
    T = DeviceHaarBatch(seed=777, n_samples=n).as_array()

    class L:
        def __iter__(self):
            return iter(T)

    b = TemplateOptimizer(basis, BasicCost(), training_restarts=R, seed=2, override_fail=True)
    lb, _, db = b.approximate_from_distribution(L())
    assert la == lb
    for i in range(0, n, 53):
        assert da[i].cycles == db[i].cycles and np.array_equal(np.asarray(da[i].Xk), np.asarray(db[i].Xk))
    assert np.mean(np.asarray(la) < 1e-10) > 0.99  # (exact size only, 8 restarts: a few targets next to a region's face miss it)


def test_v2_template_with_polytopes_runs_at_the_predicted_size():
    """CircuitTemplateV2(use_polytopes=True) (basisv2.py:77-85 -> polytope_wrap.py:39-94): with every gate parameter bounded to a
    point the circuit is one of fixed gates and coverage.py gives its regions -- sqrt(iSWAP) here: every target runs ONLY at the
    size the rule |z| <= x - y assigns and is solved there; a template whose gates are free raises; so does a target out of reach."""
    from slam_decomposition_amd.basisv2 import CircuitTemplateV2
    from slam_decomposition_amd.cost_function import SquareCost
    from slam_decomposition_amd.gates import RiSwapGate
    from slam_decomposition_amd.weyl import c1c2c3

    basis = CircuitTemplateV2(base_gates=[RiSwapGate], maximum_span_guess=3, use_polytopes=True)
    assert basis.spanning_range is None
    with pytest.raises(NotImplementedError):
        basis.get_spanning_range(o.haar_unitary(0))  # alpha free: no single polytope per size
    basis.build(3)
    for name in basis.parameter_names():
        if name.startswith("Q"):
            basis.add_bound(name, max=0.5, min=0.5)
    T = [o.haar_unitary(300 + i) for i in range(10)]
    want = []
    for t in T:
        c1, c2, c3 = c1c2c3(t)
        if c1 > 0.5:
            c1, c3 = 1 - c1, -c3
        want.append(2 if abs(c3) <= c1 - c2 + 1e-9 else 3)
    assert [list(basis.get_spanning_range(t)) for t in T] == [[k] for k in want]
    optm = TemplateOptimizer(basis, SquareCost(), training_restarts=16, seed=8, override_fail=True)

    class S:
        def __iter__(self):
            return iter(T)

    loss, _, data = optm.approximate_from_distribution(S())
    for t, td in enumerate(data):
        assert td.cycles == want[t] and td.success_label == 1 and td.loss_result <= 1e-10
        basis.build(td.cycles)
        assert abs(o.square_cost(basis.eval(td.Xk), T[t]) - td.loss_result) < 1e-12
        assert np.allclose(np.asarray(td.Xk)[-td.cycles:], 0.5)
    assert optm.last_stats["items"][1] == 0  # nobody ran the one-gate template
    weak = CircuitTemplateV2(base_gates=[RiSwapGate], maximum_span_guess=2, use_polytopes=True)
    weak.build(2)
    for name in ("Q0", "Q1"):
        weak.add_bound(name, max=0.25, min=0.25)
    with pytest.raises(ValueError, match="Monodromy did not find"):
        weak.get_spanning_range(G.SwapGate().to_matrix())


# ------------------------------------------------------------------------------------------------------------------
# result arrays of big windows: page-locked blocks, recycled (slam_host_alloc / _ffi.result_pool)
# ------------------------------------------------------------------------------------------------------------------
def test_big_result_arrays_come_from_the_pinned_pool_and_are_recycled():
    """``pinned=True`` (what TemplateOptimizer passes for one blocking call alone on the device): the fetch of a big window writes into
    page-locked blocks of ``_ffi.result_pool`` (DMA, no staging; nothing handed back to the C allocator between calls); the values are those of a second fetch of the resident results; a block goes back to the pool when the last
    view of it dies and is handed out again."""
    ctx = _ffi.Context(0)
    try:
        n = 20000
        ctx.sample_haar(991, n)
        ctx.set_gates(np.stack([SQ]))
        ctx.set_cost(0)
        prm = _ffi.OptParams(restarts=6, seed=3, flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED)
        seqs = [[0], [0, 0], [0, 0, 0]]
        before = _ffi.result_pool.allocated
        loss, x, cyc = ctx.decompose_range(0, n, 1, 3, seqs, prm, 1e-10, pinned=True)
        assert x.shape == (n, 24) and x.base is not None and _ffi.result_pool.allocated > before  # 3.8 MB: from the pool
        assert loss.base is None or loss.nbytes >= _ffi.ResultPool.MIN_BYTES
        pinned_blocks = _ffi.result_pool.allocated
        l2, x2, c2 = ctx.fetch_results_range(3, 0, n)  # pageable arrays (recycled too: _ffi.pageable_pool)
        assert _ffi.result_pool.allocated == pinned_blocks and np.array_equal(loss, l2) and np.array_equal(x, x2) and np.array_equal(cyc, c2)
        assert np.mean(loss < 1e-8) > 0.99
        assert np.all(x[cyc == 2][:, 18:] == 0.0)  # rows are zero-padded behind 6 (cycles + 1) parameters
        addr, row = x.ctypes.data, x[7].copy()
        keep = x[7:8]
        del x
        l3, x3, c3 = ctx.fetch_results_range(3, 0, n, pinned=True)
        assert x3.ctypes.data != addr and np.array_equal(keep[0], row)  # the block is still out: a view is alive
        del keep
        count = _ffi.result_pool.allocated
        l4, x4, c4 = ctx.fetch_results_range(3, 0, n, pinned=True)
        assert _ffi.result_pool.allocated == count and np.array_equal(x4, x2)  # served from the pool's idle blocks
    finally:
        ctx.close()


def test_results_of_an_earlier_call_survive_later_calls():
    """Result arrays are views of recycled blocks: a block must stay out of the pools for as long as anything of the call's results is
    alive -- ``target_data`` of a first call (single-call path: page-locked blocks; windowed path: pageable ones) reads the same after
    later calls of the same sizes have come and gone."""
    basis = CircuitTemplate(base_gates=[G.RiSwapGate(0.5)], maximum_span_guess=3)
    for n, window in ((12000, 1 << 30), (12000, 2048)):
        first = TemplateOptimizer(basis, BasicCost(), training_restarts=6, seed=11, override_fail=True)
        first.WINDOW_TARGETS = window
        loss1, _, data1 = first.approximate_from_distribution(DeviceHaarBatch(seed=500, n_samples=n))
        picks = list(range(0, n, 499)) + [n - 1]
        saved = [(data1[i].loss_result, data1[i].cycles, np.array(data1[i].Xk, copy=True)) for i in picks]
        loss1_copy = np.array(loss1, copy=True)
        for s in (501, 502, 503):
            again = TemplateOptimizer(basis, BasicCost(), training_restarts=6, seed=12, override_fail=True)
            again.WINDOW_TARGETS = window
            loss_s, _, data_s = again.approximate_from_distribution(DeviceHaarBatch(seed=s, n_samples=n))
            del loss_s, data_s, again  # (their blocks go back to the pools and are handed out again by the next call)
        assert np.array_equal(np.asarray(loss1), loss1_copy)
        for i, (l, c, x) in zip(picks, saved):
            assert data1[i].loss_result == l and data1[i].cycles == c and np.array_equal(np.asarray(data1[i].Xk), x)
