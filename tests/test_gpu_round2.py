"""GPU: ordered (reference-order) early exit, evaluation accounting, resident-result initialisation, the RCCL
communicator through the C ABI, sharded devices = single device bit for bit."""
import numpy as np
import pytest

from oracle import slam_oracle as o
from slam_decomposition_amd import _ffi

pytestmark = pytest.mark.gpu

SQ = o.riswap_matrix(0.5)
SEQS = [[0], [0, 0], [0, 0, 0]]
ORDERED = _ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED


def test_ordered_early_exit_picks_the_restart_the_sequential_loop_stops_at(hip_ctx):
    """optimizer.py:253-295 runs the restarts one after the other and breaks at the first one below the threshold.
    Restarts are independent here, so the run WITHOUT early exit tells what each restart ends with; with
    SLAM_FLAG_ORDERED the stage result must be the lowest-index restart below stop_loss of that run, bit for bit,
    whatever the launch shape."""
    N, R, k = 24, 12, 2
    targets = o.haar_batch(N, seed0=6100)
    hip_ctx.set_targets(targets)
    hip_ctx.set_gates(SQ[None])
    stop = 1e-13
    full = hip_ctx.minimize_stage([0] * k, _ffi.OptParams(restarts=R, seed=5, flags=0, stop_loss=stop))
    want_r = np.empty(N, dtype=np.int32)
    for t in range(N):
        below = np.nonzero(full["item_loss"][t] < stop)[0]
        want_r[t] = below[0] if len(below) else int(np.argmin(full["item_loss"][t]))
    assert np.any(want_r > 0), "test needs targets whose first restart fails"
    for ipq in (0, 3):
        got = hip_ctx.minimize_stage([0] * k, _ffi.OptParams(restarts=R, seed=5, flags=ORDERED, stop_loss=stop, items_per_quad=ipq))
        assert np.array_equal(got["best_restart"], want_r)
        assert np.array_equal(got["best_loss"], full["item_loss"][np.arange(N), want_r])
        # restarts below the winner are never pre-empted; restarts above it may be
        for t in range(N):
            assert np.all(got["item_status"][t, : want_r[t] + 1] != _ffi.ST_PREEMPTED)
            assert np.array_equal(got["item_loss"][t, : want_r[t] + 1], full["item_loss"][t, : want_r[t] + 1])
    assert np.any(got["item_status"] == _ffi.ST_PREEMPTED)


def test_ordered_span_loop_is_bitwise_reproducible_over_runs_windows_and_launch_shapes(hip_ctx):
    N = 40
    hip_ctx.set_targets(o.haar_batch(N, seed0=777))
    hip_ctx.set_gates(SQ[None])
    prm = _ffi.OptParams(restarts=8, seed=42, flags=ORDERED)
    whole = hip_ctx.decompose(1, 3, SEQS, prm, 1e-10)
    again = hip_ctx.decompose(1, 3, SEQS, _ffi.OptParams(restarts=8, seed=42, flags=ORDERED, items_per_quad=4), 1e-10)
    a = hip_ctx.decompose_range(0, 13, 1, 3, SEQS, prm, 1e-10)
    b = hip_ctx.decompose_range(13, 27, 1, 3, SEQS, prm, 1e-10)
    for i in range(3):
        assert np.array_equal(whole[i], again[i])
        assert np.array_equal(np.concatenate([a[i], b[i]]), whole[i])
    assert np.all(whole[0] < 1e-10)
    # the same targets as a shard of a larger job: seeds keyed on target_base + local index
    shard = o.haar_batch(N, seed0=777)[10:25]
    hip_ctx.set_targets(shard)
    sh = hip_ctx.decompose(1, 3, SEQS, _ffi.OptParams(restarts=8, seed=42, flags=ORDERED, target_base=10), 1e-10)
    for i in range(3):
        assert np.array_equal(sh[i], whole[i][10:25])


def test_evaluation_accounting_adds_up(hip_ctx):
    N, R = 64, 8
    hip_ctx.set_targets(o.haar_batch(N, seed0=4400))
    hip_ctx.set_gates(SQ[None])
    for flags in (0, _ffi.FLAG_EARLY_EXIT, ORDERED):
        hip_ctx.reset_stats()
        hip_ctx.decompose(1, 3, SEQS, _ffi.OptParams(restarts=R, seed=9, flags=flags), 1e-10)
        st = hip_ctx.stats()
        for k in (1, 2, 3):
            ev, acc, pre = st["evals"][k], st["evals_accepted"][k], st["evals_preempted"][k]
            if st["items"][k] == 0:
                continue
            assert ev > 0 and 0 < acc <= ev and 0 <= pre <= ev - acc
            if flags == 0:
                assert pre == 0
            # at least one accepted evaluation (the initial point) per restart that ran to its own end
            assert acc >= 1
        if flags:
            assert sum(st["evals_preempted"]) > 0
    # per-item view of one stage: accepted <= evaluations, the packed counters come apart correctly
    out = hip_ctx.minimize_stage([0, 0], _ffi.OptParams(restarts=R, seed=9, flags=0))
    assert np.all(out["item_evals"] >= out["item_iters"] + 1)
    assert np.all(out["item_evals"] < 1 << 20)


def test_results_of_windows_never_decomposed_read_as_nothing_found(hip_ctx):
    """slam_fetch_results_range of a window no call has decomposed must not return uninitialised memory."""
    N = 50
    hip_ctx.set_targets(o.haar_batch(N, seed0=12))
    hip_ctx.set_gates(SQ[None])
    prm = _ffi.OptParams(restarts=4, seed=1, flags=_ffi.FLAG_EARLY_EXIT)
    hip_ctx.decompose_range(20, 10, 1, 3, SEQS, prm, 1e-10)
    loss, x, cyc = hip_ctx.fetch_results_range(3, 0, N)
    inside = np.zeros(N, bool)
    inside[20:30] = True
    assert np.all(np.isinf(loss[~inside])) and np.all(cyc[~inside] == -1)
    assert np.all(np.isfinite(loss[inside])) and np.all(cyc[inside] >= 1)
    # a list call leaves the other targets alone as well
    hip_ctx.set_targets(o.haar_batch(N, seed0=12))
    hip_ctx.decompose_list(np.array([3, 7, 40]), 2, 2, [[0, 0]], prm, 1e-10, k_layout=3)
    loss, x, cyc = hip_ctx.fetch_results_range(3, 0, N)
    assert sorted(np.nonzero(np.isfinite(loss))[0].tolist()) == [3, 7, 40]
    assert np.all(cyc[np.isinf(loss)] == -1)


def test_a_failing_call_leaves_the_context_usable(hip_ctx):
    hip_ctx.set_targets(o.haar_batch(8, seed0=3))
    hip_ctx.set_gates(SQ[None])
    prm = _ffi.OptParams(restarts=4, seed=1, flags=_ffi.FLAG_EARLY_EXIT)
    good = hip_ctx.decompose(1, 3, SEQS, prm, 1e-10)
    with pytest.raises(_ffi.SlamHipError):
        hip_ctx.decompose(1, 3, [[0], [0, 5], [0, 0, 0]], prm, 1e-10)  # gate index outside the table
    with pytest.raises(_ffi.SlamHipError):
        hip_ctx.decompose(1, 3, SEQS, _ffi.OptParams(restarts=4, maxiter=10**6), 1e-10)  # beyond SLAM_MAX_MAXITER
    again = hip_ctx.decompose(1, 3, SEQS, prm, 1e-10)
    assert np.array_equal(good[2], again[2]) and np.all(again[0] < 1e-10)


def test_rccl_communicator_through_the_c_abi_single_rank(hip_ctx, tmp_path):
    """slam_comm_* on a world of one rank: librccl is dlopen'ed, ncclCommInitRank / ncclAllReduce run on the GPU,
    and the final best-loss all-reduce takes the context's RESIDENT buffer device to device.  (Several ranks need
    several GPUs: RCCL refuses duplicate devices; the N > 1 host logic is covered by tests/test_parallel_gloo.py.)"""
    from slam_decomposition_amd.parallel import RcclComm, merge_results

    comm = RcclComm(0, 0, 1, str(tmp_path / "id"))
    try:
        a = np.array([3.0, -1.5, 7.25])
        for f in (comm.allreduce_min, comm.allreduce_max, comm.allreduce_sum):
            b = a.copy()
            f(b)
            assert np.array_equal(a, b)
        comm.barrier()
        N = 48
        hip_ctx.set_targets(o.haar_batch(N, seed0=91))
        hip_ctx.set_gates(SQ[None])
        best_loss, best_x, best_cycles = hip_ctx.decompose(1, 3, SEQS, _ffi.OptParams(restarts=6, seed=2, flags=ORDERED), 1e-10)
        # two windows of the resident results into a larger job vector, the rest stays +inf
        comm.raw.merge_begin(N + 10)
        comm.raw.merge_add(hip_ctx, 0, 20, 5)
        comm.raw.merge_add(hip_ctx, 20, N - 20, 25)
        n_below, merged = comm.raw.allreduce_min_merged(1e-8, want_merged=True)
        assert len(merged) == N + 10 and comm.raw.rccl_rank_world() == (0, 1) and (comm.rccl_rank, comm.rccl_world) == (0, 1)
        assert np.array_equal(merged[5 : 5 + N], best_loss)
        assert np.all(np.isinf(merged[:5])) and np.all(np.isinf(merged[5 + N :]))
        assert n_below == int((best_loss < 1e-8).sum())
        # host windows min-merge into the same vector
        comm.raw.merge_begin(8)
        comm.raw.merge_add_host(np.array([1.0, 2.0, 3.0]), 2)
        comm.raw.merge_add_host(np.array([5.0, 0.5]), 3)
        n_below, merged = comm.raw.allreduce_min_merged(1.5, want_merged=True)
        assert np.array_equal(merged, [np.inf, np.inf, 1.0, 2.0, 0.5, np.inf, np.inf, np.inf]) and n_below == 2
        with pytest.raises(_ffi.SlamHipError):
            comm.raw.merge_add(hip_ctx, 0, N, 7)  # window beyond the merge vector
        # the library checks the capacity of the host copy itself (a short buffer used to be a heap overflow)
        lib = _ffi.load_library()
        short = np.empty(3)
        nb = _ffi.C.c_int64(0)
        assert lib.slam_allreduce_min(comm.raw._h, 1.0, _ffi.C.byref(nb), short.ctypes.data_as(_ffi.C.c_void_p), 3) == -1  # SLAM_ERR_INVALID
        # in place means in place: a non-contiguous view is refused instead of silently reducing a copy
        with pytest.raises(ValueError):
            comm.allreduce_min(np.zeros((4, 4))[:, 1])
        L, X, C = merge_results(comm, N, 0, best_loss, best_x, best_cycles)
        assert np.array_equal(L, best_loss) and np.array_equal(X, best_x) and np.array_equal(C, best_cycles)
    finally:
        comm.close()


def test_sharded_devices_equal_the_single_device_run_bit_for_bit():
    """TemplateOptimizer(devices=[...]): every shard uploads only its own targets and keys its seeds on the global
    target index; with the default deterministic=True the sharded job returns exactly the single-device results."""
    from slam_decomposition_amd.basis import CircuitTemplate
    from slam_decomposition_amd.cost_function import BasicCost
    from slam_decomposition_amd.gates import RiSwapGate
    from slam_decomposition_amd.optimizer import TemplateOptimizer
    from slam_decomposition_amd.sampler import DeviceHaarBatch, HaarBatch

    for sampler in (HaarBatch(seed0=4321, n_samples=13), DeviceHaarBatch(seed=5, n_samples=13)):
        def run(**extra):
            basis = CircuitTemplate(base_gates=[RiSwapGate(0.5)], maximum_span_guess=3)
            opt = TemplateOptimizer(basis, BasicCost(), training_restarts=6, seed=77, **extra)
            return opt.approximate_from_distribution(sampler)[2]

        one, two, three = run(), run(devices=[0, 0]), run(devices=[0, 0, 0])
        for a, b, c in zip(one, two, three):
            assert a.cycles == b.cycles == c.cycles and a.success_label == b.success_label == c.success_label == 1
            assert a.loss_result == b.loss_result == c.loss_result
            assert np.array_equal(a.Xk, b.Xk) and np.array_equal(a.Xk, c.Xk)


def test_use_callback_records_the_sequential_restart_loop(caplog):
    """use_callback=True (optimizer.py:217-224,238,287-292): training_loss entries are [-1, k, loss after every
    iteration of every restart the sequential loop runs, -1, k + 1, ...], coordinate_list the Weyl coordinates of the
    same points (reset per span).  Checked item by item against the NumPy port of the kernel's iteration
    (oracle/bfgs_port.py) run sequentially with the reference's break rule."""
    import logging

    from oracle import bfgs_port
    from slam_decomposition_amd.basis import CircuitTemplate
    from slam_decomposition_amd.cost_function import BasicCost
    from slam_decomposition_amd.gates import RiSwapGate
    from slam_decomposition_amd.optimizer import TemplateOptimizer
    from slam_decomposition_amd.sampler import HaarBatch

    N, R, seed = 3, 4, 11
    sampler = HaarBatch(seed0=2024, n_samples=N)
    targets = list(sampler)
    basis = CircuitTemplate(base_gates=[RiSwapGate(0.5)], maximum_span_guess=3)
    opt = TemplateOptimizer(basis, BasicCost(), use_callback=True, override_fail=True, training_restarts=R, seed=seed)
    with caplog.at_level(logging.INFO):
        training_loss, coordinate_list, data = opt.approximate_from_distribution(sampler)
    msgs = [r.getMessage() for r in caplog.records]
    assert sum(m.startswith("Starting opt on template size") for m in msgs) == sum(td.cycles for td in data)
    assert sum(m.startswith("Break on cycle") for m in msgs) == sum(td.success_label for td in data)
    assert sum(m.startswith("Cycle (k =") for m in msgs) == sum(td.cycles for td in data)

    # sequential emulation with the CPU port: same seeds (Philox keyed on seed, target, restart, k)
    entry = 0
    for t in range(N):
        temp, best = [], None
        fired = []  # (entry index, span, losses recorded in that span so far) each time the break condition fires
        for k in (1, 2, 3):
            temp.extend([-1, k])
            n_span = 0
            for r in range(R):
                tr = []
                f, x, it, status, nev = bfgs_port.minimize_port(o.x0_philox(seed, t, r, k), [SQ] * k, targets[t], trace=tr)
                temp.extend(tr)
                n_span += len(tr)
                best = f if best is None or f < best else best
                if best < 1e-10 or r == R - 1:
                    fired.append((entry, k, n_span))
                    entry += 1
                if best < 1e-10:
                    break
            if best < 1e-10:
                break
        assert abs(best - data[t].loss_result) < 1e-9
        # the reference appends the SAME growing list every time the condition fires (optimizer.py:289-292)
        got = training_loss[fired[-1][0]]
        assert all(training_loss[e] is got for e, _, _ in fired)
        flags = [i for i, v in enumerate(got) if v == -1]
        assert [got[i + 1] for i in flags] == list(range(1, fired[-1][1] + 1))  # [-1, 1, ..., -1, 2, ...] span markers
        body = np.array([v for i, v in enumerate(got) if i not in flags and i - 1 not in flags])
        ref = np.array([v for i, v in enumerate(temp) if not (v == -1 or (i > 0 and temp[i - 1] == -1))])
        assert np.all(np.isfinite(body)) and np.all(body >= 0)
        # the kernel's iteration follows the port; the fp32 metric's summation order can shift an iteration count by
        # one here and there, which shifts everything recorded after it
        assert abs(len(body) - len(ref)) <= 2 + 0.03 * len(ref), (t, len(body), len(ref))
        if len(body) == len(ref):
            # same trajectory: identical to many digits at first, within a factor of a few at the fp32 metric's noise level
            # towards the end of a restart (losses fall by a decade per iteration there)
            assert np.max(np.abs(np.log10(body + 1e-15) - np.log10(ref + 1e-15))) < 1.0, t
        m = min(8, len(body), len(ref))
        assert np.allclose(body[:m], ref[:m], rtol=1e-7, atol=1e-12)  # restart 0 of span 1: first iterations
        # coordinate_list: one list per firing, reset per span, one Weyl-coordinate triple per recorded loss
        for e, k, n_span in fired:
            assert abs(len(coordinate_list[e]) - n_span) <= 2 + 0.03 * n_span
            assert all(len(c) == 3 for c in coordinate_list[e])
        span_counts = np.diff(flags + [len(got)]) - 2
        assert len(coordinate_list[fired[-1][0]]) == span_counts[-1]
    assert entry == len(training_loss) == len(coordinate_list)
    assert all(td.success_label == 1 for td in data)


def test_span_log_lines_without_callback(caplog):
    import logging

    from slam_decomposition_amd.basis import CircuitTemplate
    from slam_decomposition_amd.cost_function import BasicCost
    from slam_decomposition_amd.gates import CXGate
    from slam_decomposition_amd.optimizer import TemplateOptimizer
    from slam_decomposition_amd.sampler import HaarBatch

    basis = CircuitTemplate(base_gates=[CXGate()], maximum_span_guess=3)
    opt = TemplateOptimizer(basis, BasicCost(), training_restarts=6, seed=5)
    with caplog.at_level(logging.INFO):
        _, _, data = opt.approximate_from_distribution(HaarBatch(seed0=1, n_samples=2))
    msgs = [r.getMessage() for r in caplog.records]
    # a Haar target needs three CNOTs: every target logs the three template sizes, improving losses, one break
    for k in (1, 2, 3):
        assert msgs.count(f"Starting opt on template size {k}") == 2
    cyc = [float(m.split("Best Loss=")[1]) for m in msgs if m.startswith("Cycle (k =")]
    assert len(cyc) == 6 and all(cyc[3 * i] >= cyc[3 * i + 1] >= cyc[3 * i + 2] for i in range(2))
    assert msgs.count("Break on cycle 3") == 2
    assert [td.loss_result for td in data] == [cyc[2], cyc[5]]


def test_non_contiguous_spanning_range():
    """basis.spanning_range = [1, 3] (the reference accepts any iterable, optimizer.py:233): sizes visited in order, the
    template size 2 never tried."""
    from slam_decomposition_amd.basis import CircuitTemplate
    from slam_decomposition_amd.cost_function import BasicCost
    from slam_decomposition_amd.gates import RiSwapGate
    from slam_decomposition_amd.optimizer import TemplateOptimizer
    from slam_decomposition_amd.sampler import HaarBatch

    basis = CircuitTemplate(base_gates=[RiSwapGate(0.5)], maximum_span_guess=3)
    basis.spanning_range = [1, 3]
    opt = TemplateOptimizer(basis, BasicCost(), training_restarts=6, seed=3)
    sampler = HaarBatch(seed0=60, n_samples=5)
    _, _, data = opt.approximate_from_distribution(sampler)
    targets = list(sampler)
    for t, td in enumerate(data):
        assert td.cycles == 3 and td.success_label == 1
        W = o.template_eval(td.Xk, [SQ] * 3)
        assert abs(o.basic_cost(W, targets[t]) - td.loss_result) < 1e-12


def test_callback_failure_raises_like_the_reference_and_vz_only_v2_runs():
    from slam_decomposition_amd.basis import CircuitTemplate
    from slam_decomposition_amd.basisv2 import CircuitTemplateV2
    from slam_decomposition_amd.cost_function import BasicCost
    from slam_decomposition_amd.gates import ConversionGainGate, CXGate
    from slam_decomposition_amd.optimizer import TemplateOptimizer

    # configs[0]-like failure with the callback on: two CNOTs cannot reach a Haar target -> ValueError (optimizer.py:89-93)
    basis = CircuitTemplate(base_gates=[CXGate()], maximum_span_guess=2)
    opt = TemplateOptimizer(basis, BasicCost(), use_callback=True, training_restarts=2, seed=1)
    with pytest.raises(ValueError, match="Failed to converge"):
        opt.approximate_target_U(o.haar_unitary(3))
    assert opt.training_loss == []  # without override_fail the break condition never fires (optimizer.py:287-292)
    # utils/gates/family_extend.py:40-56: phase lambdas, rz-only layers, one cycle, SquareCost-free variant with BasicCost:
    # the template reaches targets that are the gate itself up to rz layers
    g1, g2 = 0.9, 0.4
    fn = lambda p1, p2: ConversionGainGate(p1, p2, g1, g2, 1.0)
    tmpl = CircuitTemplateV2(base_gates=[fn], maximum_span_guess=1, vz_only=True)
    tmpl.spanning_range = range(1, 2)
    target = ConversionGainGate(0.7, -1.3, g1, g2, 1.0).to_matrix()
    opt2 = TemplateOptimizer(tmpl, BasicCost(), override_fail=True, training_restarts=6, seed=2)
    td = opt2.approximate_target_U(target)
    assert td.success_label == 1 and td.cycles == 1 and len(td.Xk) == 4 + 2 and td.loss_result < 1e-10
    tmpl.build(1)
    assert abs(o.basic_cost(tmpl.eval(td.Xk), target) - td.loss_result) < 1e-12


def test_span_losses_entry_point(hip_ctx):
    N = 12
    hip_ctx.set_targets(o.haar_batch(N, seed0=8))
    hip_ctx.set_gates(SQ[None])
    best_loss, _, best_cycles = hip_ctx.decompose(1, 3, SEQS, _ffi.OptParams(restarts=5, seed=6, flags=ORDERED), 1e-10)
    sl = hip_ctx.fetch_span_losses(0, N)
    assert sl.shape == (N, _ffi.MAX_SPAN_EVAL) and np.all(np.isnan(sl[:, 3:]))
    for t in range(N):
        k = int(best_cycles[t])
        assert np.all(np.isfinite(sl[t, :k])) and np.all(np.isnan(sl[t, k:3]))   # spans after the solving one never ran
        assert sl[t, k - 1] == best_loss[t] and np.all(np.diff(sl[t, :k]) <= 0)  # running best: non-increasing


def test_fast_and_logged_paths_of_the_python_api_agree(caplog):
    """Without INFO logging TemplateOptimizer skips the per-target log lines (and the coordinates only they show); the
    results, the bookkeeping lists and the failure behaviour must be the same as with logging on."""
    import logging

    from slam_decomposition_amd.basis import CircuitTemplate
    from slam_decomposition_amd.cost_function import BasicCost
    from slam_decomposition_amd.gates import CXGate, RiSwapGate
    from slam_decomposition_amd.optimizer import TemplateOptimizer
    from slam_decomposition_amd.sampler import DeviceHaarBatch, HaarBatch

    def run(level, sampler, gate, kmax, **kw):
        basis = CircuitTemplate(base_gates=[gate], maximum_span_guess=kmax)
        opt = TemplateOptimizer(basis, BasicCost(), training_restarts=5, seed=2, **kw)
        with caplog.at_level(level):
            try:
                tl, cl, data = opt.approximate_from_distribution(sampler)
                err = None
            except ValueError as e:
                data, err = None, str(e)
        return opt, data, err

    for sampler in (HaarBatch(seed0=70, n_samples=9), DeviceHaarBatch(seed=70, n_samples=9)):
        a_opt, a, _ = run(logging.WARNING, sampler, RiSwapGate(0.5), 3)
        b_opt, b, _ = run(logging.INFO, sampler, RiSwapGate(0.5), 3)
        assert [(x.success_label, x.loss_result, x.cycles) for x in a] == [(x.success_label, x.loss_result, x.cycles) for x in b]
        assert all(np.array_equal(x.Xk, y.Xk) for x, y in zip(a, b))
        assert a_opt.training_loss == b_opt.training_loss and a_opt.best_cycle_list == b_opt.best_cycle_list
    # failure without override_fail: two CNOTs cannot reach a Haar target -- the first target raises, its loss is recorded
    for level in (logging.WARNING, logging.INFO):
        f_opt, data, err = run(level, HaarBatch(seed0=70, n_samples=4), CXGate(), 2)
        assert data is None and err.startswith("Failed to converge") and len(f_opt.training_loss) == 1 and f_opt.best_cycle_list == [2]
    o_opt, data, err = run(logging.WARNING, HaarBatch(seed0=70, n_samples=4), CXGate(), 2, override_fail=True)
    assert err is None and [d.success_label for d in data] == [0] * 4 and len(o_opt.training_loss) == 4


def test_unknown_flag_bits_are_ignored_and_shared_seeds_equal_explicit_ones(hip_ctx):
    """slam_opt_params.flags: bits above SLAM_FLAG_ORDERED are internal (masked at the boundary).  And the start points a
    refill deals over the wave (Philox blocks spread over the 64 lanes, handed out through LDS) are the ones the oracle
    generates item by item: a run from explicit x0 = the oracle's Philox seeds ends bit for bit where the in-kernel run does."""
    N, R, k = 40, 8, 2
    hip_ctx.set_targets(o.haar_batch(N, seed0=6200))
    hip_ctx.set_gates(SQ[None])
    base = hip_ctx.minimize_stage([0] * k, _ffi.OptParams(restarts=R, seed=11, flags=ORDERED))
    junk = hip_ctx.minimize_stage([0] * k, _ffi.OptParams(restarts=R, seed=11, flags=ORDERED | 0x100 | 0x8000))
    for key in ("best_loss", "best_x", "best_restart"):  # (restarts above the winner are pre-empted at a timing-dependent point)
        assert np.array_equal(base[key], junk[key]), key
    plain = hip_ctx.minimize_stage([0] * k, _ffi.OptParams(restarts=R, seed=11, flags=0))
    x0 = np.stack([np.stack([o.x0_philox(11, t, r, k) for r in range(R)]) for t in range(N)])
    explicit = hip_ctx.minimize_stage([0] * k, _ffi.OptParams(restarts=R, seed=999, flags=0), x0=x0)
    assert np.array_equal(plain["item_loss"], explicit["item_loss"])
    assert np.array_equal(plain["item_iters"], explicit["item_iters"])
