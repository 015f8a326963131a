"""GPU, round 4:
  * the HIP path against the REFERENCE-FAITHFUL optimizer path (SciPy BFGS with its own finite differences, sequential restarts:
    src/slam/optimizer.py:233-303, :270-278) on 64 targets per basis -- fixture tests/golden/fd_reference.npz, made by
    tools/make_fd_reference.py from the oracle: equal template sizes, losses and recovered Weyl coordinates within 1e-6;
  * exact coverage of two-gate products (span_rules.two_gate_region) against the brute-force span loop;
  * ``target_data`` as a lazy sequence; ``last_stats`` as one dict for any number of devices;
  * per-item results through the packed 32-byte records.
"""
import os

import numpy as np
import pytest

from oracle import slam_oracle as o
from slam_decomposition_amd import _ffi, span_rules
from slam_decomposition_amd import gates as G

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fd_reference.npz")


def _fold(c):
    return span_rules._fold(c)


_SPEC_ON = os.environ.get("SLAM_SPECULATE", "1")[:1] != "0"  # (SLAM_SPECULATE=0: small batches through the one-wavefront loop only)


@pytest.mark.parametrize("basis", ["cx", "sqiswap"])
def test_hip_path_matches_the_finite_difference_reference_path_on_64_targets(basis):
    """north_star: "match the reference NumPy/SciPy path's converged loss and recovered Weyl coordinates to 1e-6 on identical Haar
    targets".  The fixture holds what ``run_reference(analytic_jac=False)`` -- the reference's real optimizer path -- ends with on
    64 counter-based Haar targets from the same Philox start points; the device generates the same targets (slam_sample_haar) and
    start points.  Required: every target solved below the metric's 1e-8 on both sides, EQUAL best_cycles, |loss difference| <= 1e-6,
    Weyl coordinates of the HIP path's circuits within 1e-6 of the target's, and within the reference path's own accuracy of its."""
    ref = np.load(GOLDEN)
    n, R = int(ref["n"]), int(ref["restarts"])
    gate = G.CXGate().to_matrix() if basis == "cx" else G.RiSwapGate(0.5).to_matrix()
    with _ffi.Context(0) as ctx:
        ctx.sample_haar(int(ref["target_seed"]), n)
        T = ctx.get_targets(0, n)
        for i in (0, n - 1):  # the device's targets ARE the fixture's
            assert np.max(np.abs(T[i] - o.haar_philox_port(int(ref["target_seed"]), i))) < 1e-13
        ctx.set_gates(gate[None])
        prm = _ffi.OptParams(restarts=R, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=int(ref["opt_seed"]), flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED)
        seqs = [[0] * k for k in (1, 2, 3)]
        loss, x, cyc = ctx.decompose_range(0, n, 1, 3, seqs, prm, float(ref["level"]))
        found = np.zeros((n, 3))
        for k in np.unique(cyc):
            sel = np.nonzero(cyc == k)[0]
            found[sel] = ctx.eval_c1c2c3(seqs[int(k) - 1], np.ascontiguousarray(x[sel, : 6 * (int(k) + 1)]), ndigits=-1)
        tgt = ctx.targets_c1c2c3(0, n, ndigits=-1)
    r_loss, r_cyc = ref[f"{basis}_loss"], ref[f"{basis}_cycles"]
    assert np.all(r_loss < float(ref["level"])) and np.all(loss < float(ref["level"]))
    assert np.array_equal(cyc, r_cyc), (np.nonzero(cyc != r_cyc)[0], cyc[cyc != r_cyc], r_cyc[cyc != r_cyc])
    assert np.max(np.abs(loss - r_loss)) <= 1e-6
    # coordinates: the HIP path recovers the target's to 1e-6; the reference path's own circuits sit sqrt(loss) ~ 3e-5 off the target
    # (its finite-difference floor, SURVEY.md Appendix C-9: loss ~ 1e-9), so the two paths agree to 1e-6 + 4 sqrt(reference loss)
    d_tgt = np.abs(_fold(found) - _fold(tgt)).max(axis=1)
    d_ref = np.abs(_fold(found) - _fold(ref[f"{basis}_found_coords"])).max(axis=1)
    d_fix = np.abs(_fold(tgt) - _fold(ref[f"{basis}_target_coords"])).max()
    assert d_tgt.max() <= 1e-6 and d_fix <= 1e-9, (d_tgt.max(), d_fix)
    assert np.all(d_ref <= 1e-6 + 4.0 * np.sqrt(r_loss)), (d_ref.max(), r_loss.max())


def test_exact_two_gate_regions_match_the_brute_force_span_loop():
    """span_rules.two_gate_region: for the mixed sequence of BASELINE configs[3], [iSWAP, B, iSWAP][:k], the
    predicted template size equals the size the brute-force loop ends with for every Haar target at least 1e-4 off a boundary of the
    region -- 97.75 % need two gates; for two XY-type gates g = RiSwapGate(0.4) the region decides exactly which targets a
    two-gate template reaches."""
    N, R = 6000, 12
    isw, b = G.RiSwapGate(1.0).to_matrix(), G.BerkeleyGate().to_matrix()
    with _ffi.Context(0) as ctx:
        ctx.sample_haar(515, N)
        coords = ctx.targets_c1c2c3(0, N)
        prm = _ffi.OptParams(restarts=R, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=77, flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED)
        # (a) iSWAP . L . B
        ctx.set_gates(np.stack([isw, b]))
        loss, _, cyc = ctx.decompose_range(0, N, 1, 3, [[0], [0, 1], [0, 1, 0]], prm, 1e-10)
        assert np.all(loss < 1e-8)
        gc = [ctx.c1c2c3(isw[None])[0], ctx.c1c2c3(b[None])[0]]
        seq = [gc[0], gc[1], gc[0]]
        assert span_rules.sequence_is_exact(seq, 3)
        pred = span_rules.sequence_minimal_span(coords, seq, 3)
        f = _fold(coords)
        off = (np.abs(f[:, 0] - 0.25) > 1e-4) & (np.abs(np.abs(f[:, 2]) - 0.25) > 1e-4)
        assert off.mean() > 0.99
        assert np.array_equal(pred[off], cyc[off])
        assert abs((pred == 2).mean() - 0.9775) < 0.01
        # the lower bound now uses the exact region at k = 2: it IS the template size here
        assert np.array_equal(span_rules.span_lower_bound(coords, seq, 3)[off], cyc[off])
        # (b) RiSwapGate(0.4) twice: (a, a, 0) with a = 0.2
        g = G.RiSwapGate(0.4).to_matrix()
        ctx.set_gates(g[None])
        loss2, _, cyc2 = ctx.decompose_range(0, N, 1, 2, [[0], [0, 0]], prm, 1e-10)
        gcoord = ctx.c1c2c3(g[None])[0]
        inside = span_rules.two_gate_region(gcoord, gcoord)(f[:, 0], f[:, 1], f[:, 2], 0.0)
        margin = np.minimum.reduce([np.abs(np.abs(f[:, 2]) - (f[:, 0] - f[:, 1])), np.abs(f[:, 0] + f[:, 1] + np.abs(f[:, 2]) - 0.8), np.abs(f[:, 0] - 0.4)])
        clear = margin > 2e-3
        assert inside.sum() > 50 and (~inside).sum() > 50
        assert np.all(loss2[inside & clear] < 1e-8) and np.all(cyc2[inside & clear] == 2)
        assert np.all(loss2[~inside & clear] > 1e-7)


def test_use_polytopes_with_a_mixed_basis_is_exact_now():
    """CircuitTemplate(base_gates=[iSWAP, B], use_polytopes=True): every target optimised only at the size the exact rule assigns
    (``span_rules_exact``), same cycles as the brute-force template."""
    from slam_decomposition_amd.basis import CircuitTemplate
    from slam_decomposition_amd.cost_function import BasicCost
    from slam_decomposition_amd.optimizer import TemplateOptimizer
    from slam_decomposition_amd.sampler import DeviceHaarBatch

    gates = [G.RiSwapGate(1.0), G.BerkeleyGate()]
    poly = CircuitTemplate(base_gates=gates, maximum_span_guess=3, use_polytopes=True)
    assert poly.span_rules_exact
    brute = CircuitTemplate(base_gates=gates, maximum_span_guess=3)
    res = {}
    for name, basis in (("poly", poly), ("brute", brute)):
        opt = TemplateOptimizer(basis, BasicCost(), training_restarts=12, seed=5)
        _, _, data = opt.approximate_from_distribution(DeviceHaarBatch(seed=99, n_samples=600))
        res[name] = ([d.cycles for d in data], [d.success_label for d in data], opt.last_stats)
    assert res["poly"][0] == res["brute"][0] and all(res["poly"][1])
    # ... with less work: no k = 1 stage, the k = 3 stage only for the 2 % that need it
    assert sum(res["poly"][2]["evals"]) < 0.8 * sum(res["brute"][2]["evals"])


def test_target_data_is_a_lazy_list_and_last_stats_one_dict():
    """approximate_from_distribution returns ``target_data`` as a sequence that builds its DataDictEntry objects on access (VERDICT r3
    item 5) -- list semantics kept -- and ``last_stats`` is ONE dict whatever the number of devices."""
    from slam_decomposition_amd.basis import CircuitTemplate
    from slam_decomposition_amd.basis_abc import DataDictEntry, TargetDataList
    from slam_decomposition_amd.cost_function import BasicCost
    from slam_decomposition_amd.optimizer import TemplateOptimizer
    from slam_decomposition_amd.sampler import DeviceHaarBatch

    basis = CircuitTemplate(base_gates=[G.RiSwapGate(0.5)], maximum_span_guess=3)
    opt = TemplateOptimizer(basis, BasicCost(), training_restarts=8, seed=3)
    tl, cl, data = opt.approximate_from_distribution(DeviceHaarBatch(seed=11, n_samples=300))
    assert isinstance(data, TargetDataList) and len(data) == 300 == len(tl) and cl == []
    assert isinstance(data[0], DataDictEntry) and data[7] is data[7] and data[-1] is data[299]
    assert [d.loss_result for d in data] == tl and [d.cycles for d in data] == opt.best_cycle_list
    assert all(len(d.Xk) == 6 * (d.cycles + 1) for d in data) and all(d.success_label == 1 for d in data[10:20])
    as_list = list(data)
    assert data == as_list and as_list == data and len(data[5:9]) == 4 and data[5:9] == as_list[5:9]
    with pytest.raises(IndexError):
        data[300]
    # every entry re-evaluates to its loss on the CPU oracle
    T = DeviceHaarBatch(seed=11, n_samples=300).as_array()
    g = o.riswap_matrix(0.5)
    for t in (0, 150, 299):
        W = o.template_eval(data[t].Xk, [g] * data[t].cycles)
        assert abs(o.basic_cost(W, T[t]) - data[t].loss_result) < 1e-12
    assert isinstance(opt.last_stats, dict) and opt.last_stats["evals"][1] > 0 and len(opt.last_stats_per_device or [opt.last_stats]) >= 1
    # two shards on the same GPU: the same results, ONE merged statistics dict, the shards' own dicts beside it
    opt2 = TemplateOptimizer(basis, BasicCost(), training_restarts=8, seed=3, devices=[0, 0])
    tl2, _, data2 = opt2.approximate_from_distribution(DeviceHaarBatch(seed=11, n_samples=300))
    assert tl2 == tl and data2 == data
    assert isinstance(opt2.last_stats, dict) and len(opt2.last_stats_per_device) == 2
    assert opt2.last_stats["items"] == [a + b for a, b in zip(opt2.last_stats_per_device[0]["items"], opt2.last_stats_per_device[1]["items"])]
    assert opt2.last_stats["items"][1] == opt.last_stats["items"][1]


def test_per_item_results_come_through_the_packed_records():
    """slam_minimize_stage's per-item arrays (loss, iterations, status, evaluations) are unpacked from the kernels' 32-byte records:
    every item's loss re-evaluates on the oracle, statuses and counters are sane, the stage winner is the lowest loss."""
    N, R, k = 24, 6, 2
    T = o.haar_batch(N, seed0=321)
    g = o.riswap_matrix(0.5)
    with _ffi.Context(0) as ctx:
        ctx.set_targets(T)
        ctx.set_gates(g[None])
        prm = _ffi.OptParams(restarts=R, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=9, flags=0)
        out = ctx.minimize_stage([0] * k, prm)
    il, it, st, ev = out["item_loss"], out["item_iters"], out["item_status"], out["item_evals"]
    assert il.shape == (N, R) and np.all(np.isfinite(il)) and np.all((st >= 0) & (st <= 4)) and np.all(it >= 0) and np.all(ev > it)
    assert np.array_equal(out["best_loss"], il.min(axis=1)) and np.array_equal(out["best_restart"], il.argmin(axis=1))
    X = out["item_x"] if "item_x" in out else None
    for t in (0, N - 1):
        W = o.template_eval(out["best_x"][t], [g] * k)
        assert abs(o.basic_cost(W, T[t]) - out["best_loss"][t]) < 1e-12
    assert X is None or X.shape[0] == N


def test_decompose_multi_equals_one_call_per_context():
    """slam_decompose_multi: the span loops of several contexts -- one basis gate each, the same targets -- as
    ONE chain of kernels (a multi-queue optimizer launch per span) leave in every context exactly what its own slam_decompose_range
    call leaves: losses, parameters and cycles bit for bit (ordered early exit), the same items per stage.  Contexts whose
    gates fall into different structure classes get one launch per class (the same item through two instantiations ends 1e-14
    apart); windows and repeated calls work."""
    import bench

    N, R = 700, 8
    bases = [bench.sweep_gate(b) for b in (16, 40, 71, 96, 127)] + [G.RiSwapGate(0.5).to_matrix(), G.BerkeleyGate().to_matrix()]
    prm = _ffi.OptParams(restarts=R, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=31, flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED)
    seqs = [[0], [0, 0], [0, 0, 0]]
    ctxs = [_ffi.Context(0) for _ in bases]
    try:
        for c, g in zip(ctxs, bases):
            c.sample_haar(808, N)
            c.set_gates(g[None])
        solo = []
        for c in ctxs:
            c.reset_stats()
            solo.append(c.decompose_range(0, N, 1, 3, seqs, prm, 1e-10) + (c.stats(),))
        for c in ctxs:
            c.reset_stats()
        _ffi.decompose_multi(ctxs, 0, N, 1, 3, seqs, prm, 1e-10)
        for c, (l0, x0, c0, st0) in zip(ctxs, solo):
            l1, x1, c1 = c.fetch_results_range(3, 0, N)
            assert np.array_equal(l0, l1) and np.array_equal(c0, c1) and np.array_equal(x0, x1)
            st1 = c.stats()
            assert st1["items"] == st0["items"] and st1["evals"][1] == st0["evals"][1]  # (k >= 2: pre-empted evaluations depend on timing)
        assert ctxs[0].stats()["kernel_launches"] == 6 and ctxs[1].stats()["kernel_launches"] == 0  # two gate classes x three spans
        # a window, three of the contexts, spans 2..3 only, twice in a row
        sub = [ctxs[4], ctxs[1], ctxs[6]]
        want = [c.decompose_range(100, 333, 2, 3, seqs[1:], prm, 1e-10) for c in sub]
        for _ in range(2):
            _ffi.decompose_multi(sub, 100, 333, 2, 3, seqs[1:], prm, 1e-10)
            for c, (l0, x0, c0) in zip(sub, want):
                l1, x1, c1 = c.fetch_results_range(3, 100, 333)
                assert np.array_equal(l0, l1) and np.array_equal(c0, c1) and np.array_equal(x0, x1)
        # refused: the same context twice, no ordered early exit, spans beyond 3
        with pytest.raises(_ffi.SlamHipError):
            _ffi.decompose_multi([ctxs[0], ctxs[0]], 0, N, 1, 3, seqs, prm, 1e-10)
        with pytest.raises(_ffi.SlamHipError):
            _ffi.decompose_multi(ctxs[:2], 0, N, 1, 3, seqs, _ffi.OptParams(restarts=R, seed=31, flags=_ffi.FLAG_EARLY_EXIT), 1e-10)
        # the contexts are usable on their own afterwards
        l2, _, c2 = ctxs[0].decompose_range(0, N, 1, 3, seqs, prm, 1e-10)
        assert np.array_equal(l2, solo[0][0]) and np.array_equal(c2, solo[0][2])
    finally:
        for c in ctxs:
            c.close()


def test_override_method_nelder_mead_runs_scipys_simplex_on_the_device_objective():
    """TemplateOptimizer(override_method="Nelder-Mead") (optimizer.py:266-268; cost_function_comparison.ipynb): the simplex method
    driven from the host in lock-step over all restarts, objective values from the device.  From the same start points SciPy's own
    Nelder-Mead on the CPU oracle takes the same path: equal losses to 1e-9 (the objective values differ in the last bits only) for
    nearly every restart; "BFGS" is accepted; a method that is not implemented raises."""
    import scipy.optimize as sopt

    from slam_decomposition_amd.basis import CircuitTemplate
    from slam_decomposition_amd.cost_function import BasicCost
    from slam_decomposition_amd.optimizer import TemplateOptimizer
    from slam_decomposition_amd.sampler import HaarBatch

    gate = G.RiSwapGate(0.5)
    basis = CircuitTemplate(base_gates=[gate], maximum_span_guess=1)
    T = HaarBatch(seed0=77, n_samples=3).as_array()
    opt = TemplateOptimizer(basis, BasicCost(), training_restarts=3, seed=4, override_fail=True, override_method="Nelder-Mead")
    tl, _, data = opt.approximate_from_distribution(HaarBatch(seed0=77, n_samples=3))
    assert len(data) == 3 and all(d.cycles == 1 and d.success_label == 0 and len(d.Xk) == 12 for d in data)
    x0 = np.random.default_rng([4, 1]).random((3, 3, 12)) * 2 * np.pi
    g = o.riswap_matrix(0.5)
    agree = 0
    for t in range(3):
        ref = [sopt.minimize(lambda xx: o.loss(xx, [g], T[t]), x0[t, r], method="Nelder-Mead", options={"maxiter": 2500}) for r in range(3)]
        best = min(r.fun for r in ref)
        agree += abs(best - data[t].loss_result) < 1e-9
        assert abs(o.loss(np.asarray(data[t].Xk), [g], T[t]) - data[t].loss_result) < 1e-12  # the returned point has the returned loss
        assert data[t].loss_result < best + 1e-3
    assert agree >= 2
    assert opt.last_stats["evals"][1] > 3 * 3 * 100
    TemplateOptimizer(basis, BasicCost(), override_method="BFGS")
    with pytest.raises(NotImplementedError):
        TemplateOptimizer(basis, BasicCost(), override_method="Powell")


@pytest.mark.parametrize("basis,R", [("sqiswap", 16), ("cx", 5), ("b", 40)])
def test_one_wavefront_per_target_span_loop_equals_the_staged_launches(basis, R):
    """Small batches run the whole span loop of a target in ONE wavefront (span_wave_kernel: one launch, no stage barrier, the winner
    reduced as restarts finish); SLAM_FLAG_STAGED forces one optimizer + one bookkeeping launch per span.  Same items, same seeds, same
    quasi-Newton loop: losses, parameters, cycles and the per-span running best are equal bit for bit -- also with more restarts than
    quads (R = 40: wave-local refill) and fewer (R = 5) -- and the wave path is ONE kernel launch."""
    # (more than 16 restarts: the wave path is taken only when targets x restarts fill the chip; at most 2 targets per CU: the
    # speculative form -- one launch per span side by side + the merge, span_spec_kernel -- instead of the one-wavefront loop)
    N = 900 if R > 16 else (300 if basis == "sqiswap" else 700)
    gate = {"sqiswap": G.RiSwapGate(0.5), "cx": G.CXGate(), "b": G.BerkeleyGate()}[basis].to_matrix()
    seqs = [[0], [0, 0], [0, 0, 0]]
    flags = _ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED
    with _ffi.Context(0) as ctx:
        ctx.sample_haar(2024, 1200)
        out = {}
        # (more than 16 restarts: the library prefers the overlapped spans where they may run -- NO_OVERLAP keeps the call on the wave kernel)
        wave_flags = flags | (_ffi.FLAG_NO_OVERLAP if R > 16 else 0)
        for name, fl in (("wave", wave_flags), ("staged", flags | _ffi.FLAG_STAGED)):
            # (first another basis on the same window, so that the resident records hold an earlier call's values: the spans a call
            # does not run must read "not run" afterwards on either path)
            other = G.CXGate().to_matrix() if basis != "cx" else G.BerkeleyGate().to_matrix()
            ctx.set_gates(other[None])
            ctx.decompose_range(200, N, 1, 3, seqs, _ffi.OptParams(restarts=3, maxiter=200, seed=1, flags=fl), 1e-10, fetch=False)
            ctx.set_gates(gate[None])
            prm = _ffi.OptParams(restarts=R, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=12, flags=fl)
            ctx.reset_stats()
            res = ctx.decompose_range(200, N, 1, 3, seqs, prm, 1e-10)
            out[name] = res + (ctx.fetch_span_losses(200, N), ctx.stats())
        (l0, x0, c0, s0, st0), (l1, x1, c1, s1, st1) = out["wave"], out["staged"]
        assert np.array_equal(l0, l1) and np.array_equal(c0, c1) and np.array_equal(x0, x1)
        assert np.array_equal(np.isnan(s0), np.isnan(s1)) and np.array_equal(np.nan_to_num(s0), np.nan_to_num(s1))
        assert st0["kernel_launches"] == (4 if (N <= 512 and _SPEC_ON) else 1) and st1["kernel_launches"] == (2 if basis == "b" else 3)  # (B reaches everything in two)
        assert st0["items"] == st1["items"] and st0["evals"][1] == st1["evals"][1]  # (k = 1: nothing is pre-empted)
        if N <= 512 and _SPEC_ON:
            # speculative spans: the k = 2 stage ran for every target as in the loop; of the k = 3 stage (run for ALL targets) only the
            # share of the targets the loop would have taken there is booked as accepted, the rest as pre-empted
            # (which later restarts an early exit cuts short depends on timing, so only the totals' order is compared)
            assert st0["evals"][3] > st1["evals"][3] and st0["evals_preempted"][3] > st1["evals_preempted"][3]
            assert st0["evals_preempted"][3] >= 0.5 * st0["evals"][3]  # sqrt(iSWAP): 79 % of the targets stop at k = 2
        assert np.all(l0 < 1e-8)
        # spans 2..3 only, and a batch too big for the wave path (falls back by itself)
        prm = _ffi.OptParams(restarts=R, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=12, flags=flags)
        a = ctx.decompose_range(0, 64, 2, 3, seqs[1:], prm, 1e-10)
        b = ctx.decompose_range(0, 64, 2, 3, seqs[1:], _ffi.OptParams(restarts=R, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=12, flags=flags | _ffi.FLAG_STAGED), 1e-10)
        assert all(np.array_equal(u, v) for u, v in zip(a, b))


def test_wave_loop_edge_cases_square_cost_dense_gate_single_target():
    """The one-wavefront-per-target span loop on the corners: ONE target with ONE restart, spans 3..3 only, SquareCost, a dense
    (unstructured) basis gate, a mixed-class sequence (falls back to the per-span launches by itself) -- each equal to the staged path."""
    rng = np.random.default_rng(5)
    q, _ = np.linalg.qr(rng.normal(size=(4, 4)) + 1j * rng.normal(size=(4, 4)))  # a dense 4x4 unitary as basis gate
    flags = _ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED
    cases = [
        ("one target, one restart", G.RiSwapGate(0.5).to_matrix()[None], 1, 1, 1, 3, [[0], [0, 0], [0, 0, 0]], _ffi.COST_BASIC),
        ("span 3 only", G.CXGate().to_matrix()[None], 40, 6, 3, 3, [[0, 0, 0]], _ffi.COST_BASIC),
        ("SquareCost", G.RiSwapGate(0.5).to_matrix()[None], 50, 7, 1, 3, [[0], [0, 0], [0, 0, 0]], _ffi.COST_SQUARE),
        ("dense gate", q[None], 30, 4, 1, 2, [[0], [0, 0]], _ffi.COST_BASIC),
        ("mixed classes", np.stack([G.RiSwapGate(1.0).to_matrix(), G.BerkeleyGate().to_matrix()]), 60, 8, 1, 3, [[0], [0, 1], [0, 1, 0]], _ffi.COST_BASIC),
    ]
    for name, table, N, R, k0, k1, seqs, cost in cases:
        with _ffi.Context(0) as ctx:  # (a context's resident results have ONE row width: a fresh one per template size)
            ctx.sample_haar(11, 64)
            ctx.set_gates(table)
            ctx.set_cost(cost)
            res = []
            for extra in (0, _ffi.FLAG_STAGED):
                prm = _ffi.OptParams(restarts=R, maxiter=300, gtol=1e-9, stop_loss=1e-13, seed=2, flags=flags | extra)
                ctx.reset_stats()
                res.append(ctx.decompose_range(3, N, k0, k1, seqs, prm, 1e-10) + (ctx.stats()["kernel_launches"],))
            (l0, x0, c0, n0), (l1, x1, c1, n1) = res
            assert np.array_equal(l0, l1) and np.array_equal(x0, x1) and np.array_equal(c0, c1), name
            want = (k1 - k0 + 2) if (k1 > k0 and _SPEC_ON) else 1  # speculative spans: one launch per span + the merge
            if name == "mixed classes":  # not for the wave kernels: the overlapped spans of medium batches take it (2 launches per span + merge)
                want = 2 * (k1 - k0 + 1) + 1 if os.environ.get("SLAM_OVERLAP", "1")[:1] != "0" else n1
            assert n0 == want, (name, n0, n1)


def test_mixed_order_template_binds_the_cheapest_circuit_that_reaches_each_target():
    """MixedOrderBasisCircuitTemplate (basis.py:213-359, polytope_wrap.py:39-94) over {sqrt(iSWAP), iSWAP} as conversion-gain gates:
    every target ends with the cheapest gate multiset that reaches it.  Checked against an exhaustive run: EVERY coverage entry
    optimised over ALL targets (no region test, no outer bound) -- the entry TemplateOptimizer bound must be the first one, in cost
    order, that the exhaustive run solves; in particular no target is reachable by an entry the rules excluded for it."""
    from slam_decomposition_amd.basis import MixedOrderBasisCircuitTemplate
    from slam_decomposition_amd.cost_function import BasicCost
    from slam_decomposition_amd.optimizer import TemplateOptimizer

    pi = np.pi
    basis = MixedOrderBasisCircuitTemplate([G.ConversionGainGate(0, 0, pi / 4, 0, 1), G.ConversionGainGate(0, 0, pi / 2, 0, 1)], maximum_span_guess=3)
    n = 400
    with _ffi.Context(0) as ctx:
        ctx.sample_haar(4242, n - 4)
        T = ctx.get_targets(0, n - 4)
    # a few special targets among the Haar ones: the gates themselves, CNOT, SWAP
    special = [basis.gate_matrices[0], basis.gate_matrices[1], G.CXGate().to_matrix(), G.SwapGate().to_matrix()]
    T = np.concatenate([T, np.stack(special)])
    opt = TemplateOptimizer(basis, BasicCost(), training_restarts=16, seed=3)
    data = [opt.approximate_target_U(T[0])]
    assert data[0].success_label == 1 and basis.circuit_polytope is opt.circuit_polytopes[-1] and basis.cycles == data[0].cycles
    opt = TemplateOptimizer(basis, BasicCost(), training_restarts=16, seed=3)

    class _S:
        def __iter__(self):
            return iter(T)

    _, _, data = opt.approximate_from_distribution(_S())
    assert all(d.success_label == 1 for d in data)
    bound = opt.circuit_polytopes
    assert [d.cycles for d in data] == [len(e) for e in bound]
    # exhaustive: every entry over every target
    with _ffi.Context(0) as ctx:
        ctx.set_targets(T)
        ctx.set_gates(basis.gate_matrices)
        prm = _ffi.OptParams(restarts=16, stop_loss=1e-11, seed=77, flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED)
        solved = []
        for e in basis.coverage:
            k = len(e)
            ctx.decompose_list(np.arange(n), k, k, [e.gate_indices], prm, 1e-10, k_layout=3)
            loss, _, _ = ctx.fetch_results_range(3, 0, n)
            solved.append(loss < 1e-10)
        tgt = ctx.c1c2c3(T)
    solved = np.array(solved)  # [entries, n]
    first = np.argmax(solved, axis=0)
    assert solved.any(axis=0).all()
    idx_of = {id(e): j for j, e in enumerate(basis.coverage)}
    got = np.array([idx_of[id(e)] for e in bound])
    # targets within 1e-4 of a region boundary may fall either way (the optimiser's success is not sharp there): everything else equal
    masks = np.array([m for _, m, _ in basis.candidate_entries(tgt)])
    wide = np.array([e.inside(tgt, slack=2e-4)[0] for e in basis.coverage])
    narrow = np.array([e.inside(tgt, slack=-2e-4)[0] for e in basis.coverage])
    clear = ~np.any(wide & ~narrow, axis=0)
    assert clear.sum() > 0.95 * n
    assert np.array_equal(got[clear], first[clear]), np.nonzero(got[clear] != first[clear])
    # soundness of the rules: an excluded (entry, target) pair is never solved by the exhaustive run
    assert not np.any(solved & ~masks & clear[None, :])
    # the special targets: sqrt(iSWAP) -> one sqrt(iSWAP); iSWAP -> one iSWAP or two sqrt(iSWAP) (both cost 1: the shorter first);
    # CNOT -> two sqrt(iSWAP); SWAP -> three sqrt(iSWAP) (cost 1.5; [sqrt(iSWAP), iSWAP], also 1.5 and tried first, cannot reach it)
    assert [tuple(e.gate_indices) for e in bound[-4:]] == [(0,), (1,), (0, 0), (0, 0, 0)]
    costs = np.array([e.cost for e in bound])
    assert costs[:-4].mean() < 1.5 and set(np.unique(costs)) <= {0.5, 1.0, 1.5}
    # get_spanning_range (the reference's per-target lookup) binds the same entries; the inexact one is decided on the device
    for t in (n - 1, n - 2, int(np.nonzero(got == 3)[0][0]) if np.any(got == 3) else n - 3):
        r = basis.get_spanning_range(T[t])
        assert basis.circuit_polytope is bound[t] and list(r) == [len(bound[t])]
        basis.build(len(bound[t]))
        assert basis.gate_sequence() == list(bound[t].gate_indices)


def test_mixed_order_homogeneous_equals_use_polytopes_template():
    """One basis gate: MixedOrderBasisCircuitTemplate is CircuitTemplate(use_polytopes=True) on the (gc <= gg)-ordered gate -- same
    template sizes, and every target solved at exactly that size."""
    from slam_decomposition_amd.basis import CircuitTemplate, MixedOrderBasisCircuitTemplate
    from slam_decomposition_amd.cost_function import BasicCost
    from slam_decomposition_amd.optimizer import TemplateOptimizer
    from slam_decomposition_amd.sampler import DeviceHaarBatch

    g = G.ConversionGainGate(0, 0, np.pi / 4, 0, 1)
    mixed = MixedOrderBasisCircuitTemplate([g])
    assert mixed.span_rules_exact
    plain = CircuitTemplate(base_gates=[G.RiSwapGate(0.5)], maximum_span_guess=3, use_polytopes=True)
    out = {}
    for name, basis in (("mixed", mixed), ("plain", plain)):
        opt = TemplateOptimizer(basis, BasicCost(), training_restarts=16, seed=11)
        _, _, data = opt.approximate_from_distribution(DeviceHaarBatch(seed=5, n_samples=500))
        assert all(d.success_label == 1 for d in data)
        out[name] = [d.cycles for d in data]
    assert out["mixed"] == out["plain"]
    assert 0.7 < np.mean(np.array(out["mixed"]) == 2) < 0.88  # KAT-4: 79 % of Haar targets in two sqrt(iSWAP)


def test_exact_coverage_of_conversion_gain_gates_equals_the_brute_force_span_loop():
    """coverage.py against the device: for bases of BASELINE configs[4]'s conversion-gain sweep (weak to strong, conversion- and
    gain-heavy, none in a class with a closed-form rule) and a mixed three-gate sequence, the template size predicted from the
    monodromy inequalities IS the size the brute-force span loop solves each target at, and the targets predicted out of reach of
    three gates are exactly the ones it does not solve -- for every target at least 2e-4 away from a region boundary (the optimiser
    accepts loss < 1e-10, i.e. coordinates ~1e-5 outside)."""
    import bench
    from slam_decomposition_amd import coverage
    from slam_decomposition_amd.weyl import c1c2c3

    N = 3000
    prm = _ffi.OptParams(restarts=24, seed=8, flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED)
    cases = [(f"cg{b}", np.stack([bench.sweep_gate(b)]), [[0], [0, 0], [0, 0, 0]]) for b in (9, 27, 44, 52, 66, 77, 100, 125)]
    cases.append(("cg40+cg100+sqiswap", np.stack([bench.sweep_gate(40), bench.sweep_gate(100), G.RiSwapGate(0.5).to_matrix()]), [[0], [0, 1], [0, 1, 2]]))
    stats = {}
    with _ffi.Context(0) as ctx:
        ctx.sample_haar(515151, N)
        coords = ctx.targets_c1c2c3(0, N)
        for name, table, seqs in cases:
            ctx.set_gates(table)
            loss, _, cyc = ctx.decompose(1, 3, seqs, prm, 1e-10)
            solved = loss < 1e-10
            g = [c1c2c3(table[i]) for i in seqs[2]]
            pred = coverage.minimal_prefix(coords, g, 3, tol=0.0)
            clear = (coverage.minimal_prefix(coords, g, 3, tol=2e-4) == pred) & (coverage.minimal_prefix(coords, g, 3, tol=-2e-4) == pred)
            assert clear.mean() > 0.97, name
            reach = pred <= 3
            assert np.array_equal(solved[clear], reach[clear]), (name, np.nonzero(solved[clear] != reach[clear])[0][:5])
            ok = clear & reach
            assert np.array_equal(cyc[ok], pred[ok]), (name, np.nonzero(cyc[ok] != pred[ok])[0][:5])
            stats[name] = (float(reach.mean()), np.bincount(pred, minlength=5).tolist())
    # the sweep covers the regimes: a weak gate reaches almost nothing with three applications, strong ones everything, most in two
    fr = [s[0] for s in stats.values()]
    assert min(fr) < 0.05 and max(fr) > 0.99 and any(0.2 < v < 0.95 for v in fr), stats


def test_device_span_predictor_equals_the_host_coverage_test():
    """slam_predict_spans (the coverage half-spaces evaluated on the resident targets) against coverage.minimal_prefix on the host, for
    closed-form classes, general conversion-gain gates and a mixed sequence, with special targets (identity, the gates themselves,
    CNOT, SWAP) among 20 000 Haar ones; then TemplateOptimizer(use_polytopes=True) on a DeviceHaarBatch -- the path that uses it --
    against the host-side lookup on the same targets."""
    import bench
    from slam_decomposition_amd import coverage
    from slam_decomposition_amd.basis import CircuitTemplate
    from slam_decomposition_amd.cost_function import BasicCost
    from slam_decomposition_amd.optimizer import TemplateOptimizer
    from slam_decomposition_amd.sampler import DeviceHaarBatch
    from slam_decomposition_amd.weyl import c1c2c3

    n = 20000
    cases = {
        "cx": [G.CXGate().to_matrix()] * 3,
        "sqiswap": [G.RiSwapGate(0.5).to_matrix()] * 3,
        "iswap,b,iswap": [G.RiSwapGate(1.0).to_matrix(), G.BerkeleyGate().to_matrix(), G.RiSwapGate(1.0).to_matrix()],
        "cg52 x 5": [bench.sweep_gate(52)] * 5,
        "cg100,cg27,sqiswap": [bench.sweep_gate(100), bench.sweep_gate(27), G.RiSwapGate(0.5).to_matrix()],
    }
    with _ffi.Context(0) as ctx:
        ctx.sample_haar(31337, n)
        T = ctx.get_targets(0, n)
        special = [np.eye(4), G.CXGate().to_matrix(), G.SwapGate().to_matrix(), G.RiSwapGate(0.5).to_matrix(), bench.sweep_gate(52), bench.sweep_gate(100),
                   np.kron(o.u3(0.3, 0.2, 0.1), o.u3(1.0, 0.5, 0.2))]
        T[: len(special)] = np.stack(special)
        ctx.set_targets(T)
        coords = ctx.targets_c1c2c3(0, n)
        for name, mats in cases.items():
            g = [c1c2c3(m) for m in mats]
            for tol in (2e-8, 5e-4):
                dev = ctx.predict_spans(g, len(g), 0, n, tol=tol)
                host = coverage.minimal_prefix(coords, g, len(g), tol=tol)
                diff = np.nonzero(dev != host)[0]
                # (a target within rounding of a face may fall either way: the two sides add the same numbers in a different order)
                assert len(diff) <= 2 and np.all(diff >= len(special)), (name, tol, diff[:5], dev[diff[:5]], host[diff[:5]])
            assert dev[0] == 0 and dev[6] == 0  # identity, a local gate
        w = ctx.predict_spans([c1c2c3(cases["cx"][0])] * 3, 3, 100, 50)  # a window
        assert np.array_equal(w, coverage.minimal_prefix(coords[100:150], [c1c2c3(cases["cx"][0])] * 3, 3, tol=2e-8))
        with pytest.raises(_ffi.SlamHipError):
            ctx.predict_spans([c1c2c3(cases["cx"][0])] * 3, 3, n - 10, 50)
    # the API path: device-generated targets, use_polytopes=True -> spans predicted on the device, nothing copied back
    basis = CircuitTemplate(base_gates=[G.ConversionGainGate(0, 0, 0.3 * np.pi, 0.12 * np.pi, 1)], maximum_span_guess=3, use_polytopes=True)
    opt = TemplateOptimizer(basis, BasicCost(), training_restarts=16, seed=4, override_fail=True)
    sampler = DeviceHaarBatch(seed=77, n_samples=3000)
    loss, _, data = opt.approximate_from_distribution(sampler)
    assert sampler._cache is None  # (the batch was not copied back)
    spans_host = basis.minimal_spans(runtime_coords(sampler))
    cyc = np.array([d.cycles for d in data])
    assert np.mean(cyc == spans_host) > 0.999 and np.mean(np.asarray(loss) < 1e-8) > 0.995


def runtime_coords(sampler):
    from slam_decomposition_amd import runtime

    return runtime.get_context(0).c1c2c3(sampler.as_array())


def test_mixed_order_weak_gate_on_cphase_targets_like_the_reference_notebook():
    """scripts/haar_improvements.ipynb cell 1 at a size the kernels take: a weak gain-only ConversionGainGate as the one basis gate of a
    MixedOrderBasisCircuitTemplate, CPhase(pi / 2 / t) targets.  Every target is solved at exactly the template size the coverage set
    assigns (one gate fewer fails in a brute-force run), the sizes fall as the phase shrinks, and a target beyond five gates raises the
    lookup's error."""
    from slam_decomposition_amd.basis import CircuitTemplate, MixedOrderBasisCircuitTemplate
    from slam_decomposition_amd.cost_function import BasicCost
    from slam_decomposition_amd.optimizer import TemplateOptimizer

    gate = G.ConversionGainGate(0, 0, 0, np.pi / 16, 1)  # Weyl coordinates (1/16, 1/16, 0): CPhase(pi/2) = (1/4, 0, 0) takes four
    basis = MixedOrderBasisCircuitTemplate([gate], chatty_build=False, use_smush_polytope=0)
    cphase = lambda th: np.diag([1, 1, 1, np.exp(1j * th)]).astype(np.complex128)  # noqa: E731  (qiskit CPhaseGate)
    sizes = []
    for t in range(1, 9):
        U = cphase(np.pi / 2 / t)
        opt = TemplateOptimizer(basis, BasicCost(), training_restarts=16, seed=t)
        d = opt.approximate_target_U(U)
        assert d.success_label == 1 and d.cycles == len(basis.circuit_polytope.operations) == basis.cycles
        assert abs(basis.unit_cost(d.cycles) - d.cycles * gate.cost()) < 1e-12
        sizes.append(d.cycles)
        if d.cycles > 1:  # one gate fewer does not reach it
            brute = CircuitTemplate(base_gates=[gate], maximum_span_guess=d.cycles - 1)
            b = TemplateOptimizer(brute, BasicCost(), training_restarts=16, seed=t, override_fail=True).approximate_target_U(U)
            assert b.success_label == 0 and b.loss_result > 1e-6
    assert sizes == sorted(sizes, reverse=True) and sizes[0] >= 3 and sizes[-1] <= 2, sizes
    with pytest.raises(ValueError, match="did not find a polytope"):
        TemplateOptimizer(basis, BasicCost()).approximate_target_U(G.SwapGate().to_matrix())  # 24 applications of this gate


@pytest.mark.parametrize("basis,N,R,extra", [("cx", 3000, 16, 0), ("sqiswap", 2500, 32, 0), ("b", 1500, 24, 0), ("sqiswap", 12000, 32, "overlap")])
def test_overlapped_spans_equal_the_staged_launches(basis, N, R, extra):
    """Medium batches (more than the one-wavefront loop takes, at most 2^17 work items per span -- or any size with SLAM_FLAG_OVERLAP):
    the spans of the loop run side by side for all targets, each as the ordinary per-span pipeline on a helper context, and the
    loop's bookkeeping is applied afterwards in span order (span_merge_kernel).  Losses, parameters, cycles and the per-span running
    best equal the span-by-span launches bit for bit; the stages the loop would not have reached show up as pre-empted evaluations."""
    gate = {"sqiswap": G.RiSwapGate(0.5), "cx": G.CXGate(), "b": G.BerkeleyGate()}[basis].to_matrix()
    seqs = [[0], [0, 0], [0, 0, 0]]
    flags = _ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED
    with _ffi.Context(0) as ctx:
        ctx.sample_haar(909, N + 300)
        ctx.set_gates(gate[None])
        out = {}
        for name, fl in (("overlap", flags | (_ffi.FLAG_OVERLAP if extra == "overlap" else 0)), ("staged", flags | _ffi.FLAG_STAGED)):
            prm = _ffi.OptParams(restarts=R, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=21, flags=fl)
            ctx.reset_stats()
            res = ctx.decompose_range(100, N, 1, 3, seqs, prm, 1e-10)
            out[name] = res + (ctx.fetch_span_losses(100, N), ctx.stats())
        (l0, x0, c0, s0, st0), (l1, x1, c1, s1, st1) = out["overlap"], out["staged"]
        assert np.array_equal(l0, l1) and np.array_equal(c0, c1) and np.array_equal(x0, x1)
        assert np.array_equal(np.isnan(s0), np.isnan(s1)) and np.array_equal(np.nan_to_num(s0), np.nan_to_num(s1))
        assert st0["kernel_launches"] == 7 and st1["kernel_launches"] == (2 if basis == "b" else 3)
        assert st0["items"] == st1["items"] and st0["evals"][1] == st1["evals"][1]  # same targets per stage; k = 1: nothing pre-empted
        # every stage ran for ALL targets: more evaluations at the later spans than the loop needs, the surplus booked as pre-empted
        if basis != "cx":  # (CNOT: every target needs all three spans -- nothing is wasted)
            assert st0["evals"][3] > st1["evals"][3] and st0["evals_preempted"][3] > st1["evals_preempted"][3]
        else:
            assert st0["evals"][2] == st1["evals"][2]  # (all restarts of every target run at k = 2 on either path: none succeeds)
        assert np.all(l0 < 1e-8)
        # a window of two spans, and a second call on the same context (helpers reused, another gate table)
        ctx.set_gates(G.RiSwapGate(0.5).to_matrix()[None] if basis != "sqiswap" else G.CXGate().to_matrix()[None])
        prm = _ffi.OptParams(restarts=8, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=5, flags=flags | (_ffi.FLAG_OVERLAP if extra == "overlap" else 0))
        a = ctx.decompose_range(0, min(2000, N), 2, 3, seqs[1:], prm, 1e-10)
        prm.flags = flags | _ffi.FLAG_STAGED
        b = ctx.decompose_range(0, min(2000, N), 2, 3, seqs[1:], prm, 1e-10)
        assert all(np.array_equal(u, v) for u, v in zip(a, b))


def test_randomized_calls_equal_the_staged_launches_whatever_path_they_take():
    """40 random calls -- basis, window, restarts, span range, cost function, flags -- through whatever form the library picks (speculative
    spans, the one-wavefront loop, overlapped spans, or the staged launches themselves) against SLAM_FLAG_STAGED: results bit for bit."""
    rng = np.random.default_rng(20261004)
    gates = {"cx": G.CXGate().to_matrix(), "sqiswap": G.RiSwapGate(0.5).to_matrix(), "b": G.BerkeleyGate().to_matrix(),
             "cg": G.ConversionGainGate(0, 0, 0.3 * np.pi, 0.12 * np.pi, 1).to_matrix(), "iswap": G.RiSwapGate(1.0).to_matrix()}
    paths = set()
    with _ffi.Context(0) as c1, _ffi.Context(0) as c2, _ffi.Context(0) as c3:
        by_kmax = {1: c1, 2: c2, 3: c3}  # (a context's resident results have ONE row width)
        for c in by_kmax.values():
            c.sample_haar(4711, 7000)
        for case in range(40):
            names = list(rng.choice(list(gates), size=int(rng.integers(1, 3)), replace=False))
            table = np.stack([gates[n] for n in names])
            k0 = int(rng.integers(1, 4))
            k1 = int(rng.integers(k0, 4))
            ctx = by_kmax[k1]
            seqs = [[i % len(table) for i in range(k)] for k in range(k0, k1 + 1)]
            N = int(rng.choice([1, 3, 17, 200, 513, 900, 1025, 2500, 6000]))
            first = int(rng.integers(0, 7000 - N + 1))
            R = int(rng.choice([1, 2, 5, 16, 17, 32]))
            if N * R > 120000:
                R = 16
            cost = int(rng.choice([_ffi.COST_BASIC, _ffi.COST_SQUARE]))
            extra = int(rng.choice([0, 0, _ffi.FLAG_OVERLAP, _ffi.FLAG_NO_OVERLAP]))
            ctx.set_gates(table)
            ctx.set_cost(cost)
            res = []
            for fl in (extra, _ffi.FLAG_STAGED):
                prm = _ffi.OptParams(restarts=R, maxiter=400, gtol=1e-9, stop_loss=1e-13, seed=1000 + case,
                                     flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED | fl)
                ctx.reset_stats()
                out = ctx.decompose_range(first, N, k0, k1, seqs, prm, 1e-10)
                res.append(out + (ctx.fetch_span_losses(first, N), ctx.stats()["kernel_launches"]))
            (l0, x0, c0, s0, n0), (l1, x1, c1, s1, n1) = res
            tag = (names, N, R, k0, k1, cost, extra)
            assert np.array_equal(l0, l1) and np.array_equal(x0, x1) and np.array_equal(c0, c1), tag
            assert np.array_equal(np.isnan(s0), np.isnan(s1)) and np.array_equal(np.nan_to_num(s0), np.nan_to_num(s1)), tag
            nk = k1 - k0 + 1
            paths.add("staged" if n0 == n1 and n0 != 1 else ("wave" if n0 == 1 else ("spec" if n0 == nk + 1 else ("overlap" if n0 == 2 * nk + 1 else "?"))))
    assert "?" not in paths and {"overlap", "staged"} <= paths and ("spec" in paths or "wave" in paths), paths


def test_device_sampler_and_predictor_reproduce_the_recorded_haar_volumes():
    """The measure on the device: 2^22 targets from the device Haar sampler through slam_predict_spans -- the fraction that needs at
    most k gates must be the Haar volume the reference recorded for k applications of the gate (extended_results.json via
    tests/golden/reference_haar_volumes.json: sqrt(iSWAP) x 2 = 0.790117, sqrt(B) x 3 = 0.995810, sqrt(CNOT) x 4 = 0.959883, ...), within
    five standard errors (2.5e-4 at most) -- a joint check of the sampler's distribution, the device lookup and coverage.py."""
    import json

    from slam_decomposition_amd.weyl import c1c2c3

    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_haar_volumes.json")))
    n = 1 << 22
    with _ffi.Context(0) as ctx:
        ctx.sample_haar(20261004, n)
        for name, v in ref.items():
            g = c1c2c3(G.ConversionGainGate(0, 0, v["gc"], v["gg"], v["t"]).to_matrix())
            kmax = min(max(int(k) for k in v["base_vol"]), 5)
            spans = ctx.predict_spans([g] * kmax, kmax, 0, n, tol=0.0)
            for k, vol in v["base_vol"].items():
                k = int(k)
                if k > kmax:
                    continue
                frac = float(np.mean((spans <= k) & (spans >= 1)))
                se = np.sqrt(max(vol * (1 - vol), 1e-9) / n)
                assert abs(frac - vol) <= 5 * se + 1e-5, (name, k, frac, vol)
