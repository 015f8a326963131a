"""``TemplateOptimizer`` (reference: src/slam/optimizer.py) on the HIP library.

Same constructor, same ``approximate_target_U`` / ``approximate_from_distribution`` results
(``training_loss``, ``coordinate_list``, ``[DataDictEntry]``), same log lines and the same
``ValueError`` on failure -- but the triple loop of ``_run`` (spans x restarts x BFGS iterations,
optimizer.py:233-303) runs for the whole batch of targets at once inside libslamhip:
every (target, restart) pair is one quasi-Newton minimisation owned by a quad of GPU lanes.

Differences a caller can observe:
  * restarts of one target run concurrently.  With ``deterministic=True`` (default) the result of a span is
    the LOWEST-INDEX restart that ends below the threshold -- the restart the reference's sequential loop
    breaks at (optimizer.py:287-295) -- and results are bitwise reproducible for a fixed ``seed``, whatever the
    scheduling, ``devices`` sharding or batch composition.  ``deterministic=False`` lets the first restart
    to FINISH win (fewer evaluations, timing-dependent winner among equally successful restarts);
  * multi-start seeds come from Philox keyed on ``seed`` instead of NumPy's global generator
    (``seed=None`` draws the key from that generator, keeping "unseeded" behaviour);
  * gradients are analytic, so converged losses sit near 1e-14 instead of the reference's
    finite-difference floor of about 1e-10.
"""
from __future__ import annotations

import logging
from typing import List, Sequence

import numpy as np

from . import _ffi, runtime
from .basis import CircuitTemplate
from .basisv2 import CircuitTemplateV2
from .basis_abc import DataDictEntry, LazyList, RowBlocks, TargetDataList, VariationalTemplate
from .cost_function import BasicCost, SquareCost, UnitaryCostFunction
from .sampler import SampleFunction

SUCCESS_THRESHOLD = 1e-10  # optimizer.py:18
TRAINING_RESTARTS = 5  # optimizer.py:19
MAXITER = 2500  # optimizer.py:275
DEFAULT_GTOL = 1e-9
DEFAULT_STOP_LOSS = 1e-13

_FAIL_MSG = (
    "Failed to converge within error threshold. Try increasing restart attempts or increasing "
    "temperature scaling on preseed."
)


class TemplateOptimizer:
    def __init__(
        self,
        basis: VariationalTemplate,
        objective: UnitaryCostFunction,
        use_callback=False,
        override_fail=False,
        success_threshold=None,
        training_restarts=None,
        override_method=None,
        device=None,
        devices=None,
        seed=None,
        gtol=DEFAULT_GTOL,
        stop_loss=None,
        deterministic=True,
        auto_shards=1,
        windows_in_flight=None,
    ):
        self.basis = basis
        self.objective = objective
        self.preseeding = self.basis.preseeded
        self.use_callback = use_callback
        self.training_loss = LazyList()  # per target: final loss (optimizer.py:307-309); a list whose big stretches stay arrays
        self.coordinate_list = []
        self.best_cycle_list = LazyList()
        self.override_fail = override_fail
        self.override_method = override_method
        self.success_threshold = SUCCESS_THRESHOLD if success_threshold is None else success_threshold
        self.training_restarts = TRAINING_RESTARTS if training_restarts is None else training_restarts
        assert not (self.preseeding and self.override_fail)
        assert not (self.preseeding and self.basis.n_qubits != 2)

        if not isinstance(basis, (CircuitTemplate, CircuitTemplateV2)):
            raise NotImplementedError("the HIP optimizer needs a slam_decomposition_amd CircuitTemplate / CircuitTemplateV2")
        self._v2 = isinstance(basis, CircuitTemplateV2)
        if isinstance(self.objective, SquareCost):
            self._cost_kind = _ffi.COST_SQUARE
        elif isinstance(self.objective, BasicCost):
            self._cost_kind = _ffi.COST_BASIC
        else:
            # the reference raises this for objectives its objective_func does not know (optimizer.py:211)
            raise ValueError("Unrecognized Cost Function")
        if use_callback and not deterministic:
            raise ValueError("use_callback=True records the reference's sequential restart loop: it needs deterministic=True")
        # optimizer.py:266-268: override_method goes to scipy.optimize.minimize.  "BFGS" (and, for V2 templates, the methods the
        # reference selects itself for bounds / constraints) is the loop the kernels run; "Nelder-Mead" (cost_function_comparison.ipynb)
        # is driven from the host with the objective evaluated on the device (host_methods.py); anything else is not implemented
        from . import host_methods

        self._host_method = None
        if override_method in host_methods.SUPPORTED and not self._v2 and not use_callback:
            self._host_method = override_method
        elif override_method not in (None, "BFGS") and not (self._v2 and override_method in ("L-BFGS-B", "SLSQP")):
            raise NotImplementedError(f"override_method={override_method!r}: implemented are BFGS (device), Nelder-Mead (fixed-gate "
                                      "templates without callback; host-driven, device objective), L-BFGS-B / SLSQP for V2 templates")
        if getattr(basis, "mixed_order", False) and (use_callback or self._host_method is not None):
            raise NotImplementedError("MixedOrderBasisCircuitTemplate: use_callback / override_method are not implemented")
        self._no_exterior = bool(getattr(basis, "no_exterior_1q", False)) and not self._v2
        if self._no_exterior and (use_callback or self._host_method is not None or getattr(basis, "mixed_order", False)):
            raise NotImplementedError("CircuitTemplate(no_exterior_1q=True): use_callback / override_method / mixed-order templates "
                                      "are not implemented")
        if self.training_restarts <= 0:
            raise ValueError("training_restarts must be positive")
        self.device = basis.device if device is None else device
        # several GPUs in one process: targets are sharded contiguously over `devices` (one host thread and
        # one libslamhip context per entry), results concatenated on the host -- no collective needed
        self.devices = [self.device] if devices is None else list(devices)
        if not self.devices:
            raise ValueError("devices must not be empty")
        self.seed = seed
        self.deterministic = bool(deterministic)
        self.auto_shards = int(auto_shards)
        if windows_in_flight is not None:
            self.windows_in_flight = int(windows_in_flight)
        self.gtol = float(gtol)
        if stop_loss is None:
            stop_loss = min(DEFAULT_STOP_LOSS, 0.1 * self.success_threshold)
        self.stop_loss = float(stop_loss)
        self.last_stats = None

    # ------------------------------------------------------------------------------------------
    AUTO_SHARD_MIN_TARGETS = 16384
    _device_sampler = None
    _want_span_losses = True
    last_stats_per_device = None

    def _set_stats(self, per_device) -> None:
        """``last_stats`` is ONE dict whatever the number of devices (sums over the shards; ``total_ms`` = the slowest
        shard's); the shards' own dicts are in ``last_stats_per_device``."""
        self.last_stats_per_device = list(per_device)
        tot = {}
        for st in per_device:
            for key, v in st.items():
                if isinstance(v, list):
                    tot[key] = [a + b for a, b in zip(tot[key], v)] if key in tot else list(v)
                elif key == "total_ms":
                    tot[key] = max(tot.get(key, 0.0), v)
                else:
                    tot[key] = tot.get(key, 0) + v
        self.last_stats = tot

    def _opt_params(self) -> "_ffi.OptParams":
        seed = self.seed
        if seed is None:
            # reference: unseeded np.random.random (basis.py:111) -> draw the Philox key from it
            seed = int(np.random.randint(0, 2**63 - 1, dtype=np.int64))
        return _ffi.OptParams(
            restarts=int(self.training_restarts),
            maxiter=MAXITER,
            gtol=self.gtol,
            stop_loss=self.stop_loss,
            seed=int(seed) & 0xFFFFFFFFFFFFFFFF,
            # one blocking call at a time on the device leaves the library's own choice (medium calls: spans side by side; big
            # calls: _overlap_pays below decides from the batch's coverage); several shards in flight fill the chip by themselves
            flags=_ffi.FLAG_EARLY_EXIT | (_ffi.FLAG_ORDERED if self.deterministic else 0)
            | (0 if (self.deterministic and len(self.devices) == 1 and self.auto_shards <= 1) else _ffi.FLAG_NO_OVERLAP)
            | (_ffi.FLAG_NO_EXTERIOR if self._no_exterior else 0),  # basis.py:154,165: layers 0 and k pinned at the identity
        )

    # a span loop side by side for ALL targets (SLAM_FLAG_OVERLAP) runs the LAST span for every target of the call: it pays when
    # that span is needed by a good share of them anyway (CNOT: all; sqrt(iSWAP): 21 % -- 18.1 -> 16.8 ms for 65 536 x 32) and is
    # pure waste when the basis covers the chamber earlier (B: every Haar target at two gates -- the most expensive stage for nothing)
    OVERLAP_MIN_TOP_SHARE = 0.15
    OVERLAP_AUTO_ITEMS = 1 << 17  # up to here the library overlaps by itself (slam_hip.h: SLAM_FLAG_OVERLAP)

    def _overlap_pays(self, ctx, count: int, ks) -> bool:
        """Big single call, resident targets: the share of THIS batch that needs the loop's last span, from the exact coverage
        regions of the gate sequence evaluated on the device (``slam_predict_spans``; 0.2 ms for 65 536 targets)."""
        if not (self.deterministic and len(self.devices) == 1 and self.auto_shards <= 1):
            return False
        if count * int(self.training_restarts) <= self.OVERLAP_AUTO_ITEMS or len(ks) < 2 or ks[0] != 1 or ks[-1] > 3:
            return False
        # the share is a property of the basis and of the targets' distribution: measured on the first big batch a basis sees and kept
        # with the basis (a heuristic for speed only -- results are bit-equal either way)
        cache = self.basis.__dict__.setdefault("_top_span_share", {})
        share = cache.get(ks[-1])
        if share is None:
            try:
                coords = getattr(self.basis, "_gate_coords_all", None)
                if coords is None:
                    from .weyl import c1c2c3

                    coords = self.basis._gate_coords_all = [c1c2c3(m) for m in self.basis.gate_matrices]
                spans = ctx.predict_spans([coords[i] for i in self.basis.gate_sequence(ks[-1])], ks[-1], 0, count)
            except (NotImplementedError, ValueError):
                return False
            share = cache[ks[-1]] = float(np.mean(spans >= ks[-1]))
        return share >= self.OVERLAP_MIN_TOP_SHARE

    def _run_batch(self, targets: np.ndarray, spanning_range: Sequence[int]):
        """``_run`` (optimizer.py:188-313) for all targets at once.  Returns
        (best_result[N], best_Xk list, best_cycles[N])."""
        ks = list(spanning_range)
        if not ks:
            raise ValueError("empty spanning range")
        if ks[0] <= 0:
            raise ValueError()  # CircuitTemplate.build(n_repetitions <= 0), basis.py:127-128
        if max(ks) > _ffi.MAX_SPAN_MINIMIZE:
            raise NotImplementedError(
                f"template spans up to {_ffi.MAX_SPAN_MINIMIZE} are implemented on the HIP path "
                f"(got {max(ks)})"
            )
        if ks != list(range(ks[0], ks[-1] + 1)):
            return self._run_batch_any_order(targets, ks)
        if ks[-1] > _ffi.MAX_SPAN_MINIMIZE:
            raise NotImplementedError(
                f"template spans up to {_ffi.MAX_SPAN_MINIMIZE} are implemented on the HIP path "
                f"(got maximum_span_guess = {ks[-1]})"
            )
        gate_seqs = [self.basis.gate_sequence(k) for k in ks]
        prm = self._opt_params()
        n = len(targets)

        if len(self.devices) == 1 and self.auto_shards <= 1 and n > self.WINDOW_TARGETS and self.windows_in_flight > 1:
            return self._run_batch_windows(n, targets, ks, gate_seqs, prm)
        devices = self.devices
        if len(devices) == 1 and self.auto_shards > 1 and n >= self.AUTO_SHARD_MIN_TARGETS:
            # opt-in (auto_shards > 1): a big batch on one GPU as target shards side by side (one cached context + stream + host
            # thread each), bit-equal with the unsharded run.  Measured on MI355X, 65 536 x 32 sqrt(iSWAP) through
            # approximate_from_distribution (tools/r4_api_probe.py): 1 / 2 / 4 / 8 shards -> 21.7 / 21.5 / 22.8 / 26.4 ms: ONE call
            # gains nothing from being split -- what several batches in flight buy is the overlap of SUCCESSIVE
            # batches; the shards of one batch still end with their own tails.  Hence the default of 1.
            devices = devices * int(self.auto_shards)
        def run_shard(device, first, count, r=0):
            # a context per shard holding only the shard's targets (kept in runtime's cache: shard r uses slot r of its device);
            # the seeds are keyed on the GLOBAL target index (target_base), so a sharded run draws exactly the seeds of the unsharded one
            single = len(devices) == 1
            ctx = runtime.get_context(device, 0 if single else r)
            try:
                if self._device_sampler is not None:
                    self._device_sampler.fill(ctx, first, count)  # generated in place, nothing crosses PCIe
                else:
                    ctx.set_targets(targets[first : first + count])
                ctx.set_gates(self.basis.gate_matrices)
                ctx.set_cost(self._cost_kind)
                ctx.reset_stats()
                flags = prm.flags | (_ffi.FLAG_OVERLAP if (single and self._overlap_pays(ctx, count, ks)) else 0)
                sp = _ffi.OptParams(restarts=prm.restarts, maxiter=prm.maxiter, gtol=prm.gtol, stop_loss=prm.stop_loss, seed=prm.seed,
                                    flags=flags, gtol_far=prm.gtol_far, far_loss=prm.far_loss, items_per_quad=prm.items_per_quad,
                                    target_base=first)
                # (one blocking call alone on the device: results into recycled page-locked blocks, _ffi.result_pool)
                out = ctx.decompose_range(0, count, ks[0], ks[-1], gate_seqs, sp, self.success_threshold, pinned=single)
                # the running best loss per span feeds the "Cycle (k =...)" log lines only (optimizer.py:297)
                sl = ctx.fetch_span_losses(0, count) if self._want_span_losses else None
                return out + (sl,), ctx.stats()
            finally:
                pass

        if len(devices) == 1 or n < len(devices):
            (best_loss, best_x, best_cycles, self._span_losses), st = run_shard(devices[0], 0, n)
            self._set_stats([st])
        else:
            import threading

            from .parallel import shard_range

            parts = [None] * len(devices)
            errors = []
            def work(r):
                try:
                    first, count = shard_range(n, r, len(devices))
                    parts[r] = run_shard(devices[r], first, count, r)
                except Exception as exc:  # surfaced below, in rank order
                    errors.append(exc)

            threads = [threading.Thread(target=work, args=(r,)) for r in range(len(devices))]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            if errors:
                raise errors[0]
            best_loss = np.concatenate([p[0][0] for p in parts])
            best_x = np.concatenate([p[0][1] for p in parts])
            best_cycles = np.concatenate([p[0][2] for p in parts])
            self._span_losses = np.concatenate([p[0][3] for p in parts]) if self._want_span_losses else None
            self._set_stats([p[1] for p in parts])
        # per target the first 6 (cycles + 1) parameters of its row: the padded block goes back as it is (rows are cut when an
        # entry is looked at: 65 536 per-row slices cost more than the span loop on the GPU)
        return best_loss, best_x, best_cycles

    # ------------------------------------------------------------------------------------------
    # A sampler larger than one window: the targets are dealt in contiguous shares to windows_in_flight helpers -- cached contexts of
    # the one GPU, each with its own stream, stage buffers and host thread -- and every helper runs its share as SUCCESSIVE windows of
    # at most WINDOW_TARGETS targets, as bench.py runs its batches.  What several calls in flight buy is the overlap of one window's
    # stage tails, its k = 3 stage and its launch gaps with the other helpers' work; target shards of ONE window do
    # not get it (round 4, negative).  A helper's whole share is made resident before the first window starts: a generator launch
    # behind running optimizer kernels waits for a free wave slot, i.e. for the end of a stage (measured: 20-27 ms, profiles/
    # r5_api_timeline.txt), and a helper that starts late leaves a window to run alone at the end.  Seeds are keyed on the global
    # target index and the ordered early exit makes a target's result independent of what runs beside it, so the result equals the
    # single call's bit for bit.  Reference: the sequential loop over the sampler, optimizer.py:180-186.
    WINDOW_TARGETS = 65536
    windows_in_flight = 4  # measured (tools/r5_api_large_probe.py): 327 680 targets 4 / 5 / 6 helpers 82.1 / 81.8 / 81.2 ms, 655 360: 156 / 167 / 160
    window_stagger = False  # (measured: no gain, 81-83 ms either way; starting the helpers 2-9 ms apart: none either)

    def _window_plan(self, n: int):
        """Per helper the list [(first, count)] of its windows: helper j owns the contiguous share j of the n targets (shares differ by
        at most one target), cut into equal windows of at most WINDOW_TARGETS.  ``window_stagger``: a helper's first window is
        (j + 1) / f of a window and its last one the rest, so that the helpers reach their stage boundaries at different times."""
        W = max(1, int(self.WINDOW_TARGETS))
        f = max(1, min(int(self.windows_in_flight), -(-n // W)))
        plan, base = [], 0
        for j in range(f):
            T = n // f + (1 if j < n % f else 0)
            units = max(1, -(-T // W))
            sizes = [T // units + (1 if u < T % units else 0) for u in range(units)]
            if self.window_stagger and f > 1 and T >= 2 * f:
                a = max(1, (sizes[0] * (j + 1)) // f)
                if a < sizes[0]:
                    sizes = [a] + sizes[1:] + [sizes[0] - a]
            wins, first = [], base
            for c in sizes:
                if c > 0:
                    wins.append((first, c))
                    first += c
            plan.append(wins)
            base += T
        return plan

    def _run_batch_windows(self, n, targets, ks, gate_seqs, prm):
        import threading

        plan = self._window_plan(n)
        n_thr = len(plan)
        device = self.devices[0]
        flags = (prm.flags & ~_ffi.FLAG_OVERLAP) | _ffi.FLAG_NO_OVERLAP  # several calls in flight fill the chip by themselves
        parts = [[None] * len(w) for w in plan]
        stats = [[None] * len(w) for w in plan]
        errors = []
        resident = threading.Barrier(n_thr)

        def work(slot):
            try:
                wins = plan[slot]
                base = wins[0][0]
                share = wins[-1][0] + wins[-1][1] - base
                try:
                    ctx = runtime.get_context(device, slot)
                    ctx.set_gates(self.basis.gate_matrices)
                    ctx.set_cost(self._cost_kind)
                    if self._device_sampler is not None:
                        self._device_sampler.fill(ctx, base, share)
                    else:
                        ctx.set_targets(targets[base : base + share])
                except Exception:
                    resident.abort()  # the other helpers must not wait for this one
                    raise
                resident.wait()
                sp = _ffi.OptParams(restarts=prm.restarts, maxiter=prm.maxiter, gtol=prm.gtol, stop_loss=prm.stop_loss, seed=prm.seed,
                                    flags=flags, gtol_far=prm.gtol_far, far_loss=prm.far_loss, items_per_quad=prm.items_per_quad,
                                    target_base=base)
                for i, (first, count) in enumerate(wins):
                    if errors:
                        return
                    ctx.reset_stats()
                    out = ctx.decompose_range(first - base, count, ks[0], ks[-1], gate_seqs, sp, self.success_threshold)
                    sl = ctx.fetch_span_losses(first - base, count) if self._want_span_losses else None
                    parts[slot][i] = out + (sl,)
                    stats[slot][i] = ctx.stats()
            except threading.BrokenBarrierError:
                pass  # another helper failed while making its share resident: its exception is the one to surface
            except Exception as exc:  # surfaced below
                errors.append(exc)

        threads = [threading.Thread(target=work, args=(slot,)) for slot in range(n_thr)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        parts = [p for per in parts for p in per]  # in target order: shares are contiguous, windows ascending
        best_loss = np.concatenate([p[0] for p in parts])
        best_cycles = np.concatenate([p[2] for p in parts])
        self._span_losses = np.concatenate([p[3] for p in parts]) if self._want_span_losses else None
        self._set_stats([s for per in stats for s in per])
        # the parameter blocks stay per window (63 MB for 327 680 x 24 parameters would be copied once more): rows are looked up
        # through RowBlocks when an entry of target_data is built
        return best_loss, RowBlocks([p[1] for p in parts]), best_cycles

    def _run_batch_host_method(self, targets: np.ndarray, spanning_range):
        """``_run`` (optimizer.py:233-303) with ``override_method="Nelder-Mead"`` (:266-268): the simplex iterations of all
        (target, restart) items of a span advance in lock-step on the host (host_methods.nelder_mead_batch: SciPy's algorithm and
        defaults, ``maxiter=2500``), every objective value -- CircuitTemplate.eval + the cost -- comes from the device
        (``slam_eval_loss_grad``, one launch per simplex operation).  Start points: ``parameter_guess`` (basis.py:106-111), i.e.
        NumPy's global generator when ``seed`` is None, else a generator keyed on (seed, span).  Restart semantics as everywhere:
        the result of a span is the lowest-index restart below the threshold, else the lowest loss."""
        from . import host_methods

        ks = [int(k) for k in spanning_range]
        if not ks:
            raise ValueError("empty spanning range")
        n, R = len(targets), int(self.training_restarts)
        ctx = runtime.get_context(self.devices[0])
        ctx.set_targets(np.asarray(targets, dtype=np.complex128))
        ctx.set_gates(self.basis.gate_matrices)
        ctx.set_cost(self._cost_kind)
        best_loss = np.full(n, np.inf)
        best_x = [None] * n
        best_cycles = np.full(n, -1, dtype=np.int32)
        self._span_losses = np.full((n, _ffi.MAX_SPAN_EVAL), np.nan)
        evals = [0] * (_ffi.MAX_SPAN_EVAL + 1)
        for k in ks:
            if k <= 0:
                raise ValueError()  # CircuitTemplate.build(n_repetitions <= 0), basis.py:127-128
            if k > _ffi.MAX_SPAN_EVAL:
                raise NotImplementedError(f"template spans up to {_ffi.MAX_SPAN_EVAL} are implemented on the HIP path (got {k})")
            act = np.nonzero(~(best_loss < self.success_threshold))[0]
            if len(act) == 0:
                break
            npar = 6 * (k + 1)
            if self.seed is None:
                x0 = np.random.random((len(act), R, npar)) * 2 * np.pi
            else:
                x0 = np.random.default_rng([int(self.seed) & 0xFFFFFFFF, k]).random((n, R, npar))[act] * 2 * np.pi
            seq = self.basis.gate_sequence(k)
            tof = np.repeat(act, R).astype(np.int32)

            def fun(items, X, tof=tof, seq=seq):
                return ctx.eval_loss_grad(seq, X, tof[items], want_grad=False)[0]

            x, f, _, nfev = host_methods.nelder_mead_batch(fun, x0.reshape(len(act) * R, npar), maxiter=MAXITER)
            evals[k] += int(nfev.sum())
            f, x = f.reshape(len(act), R), x.reshape(len(act), R, npar)
            for j, t in enumerate(act):
                below = np.nonzero(f[j] < self.success_threshold)[0]
                r = int(below[0]) if len(below) else int(np.argmin(f[j]))  # optimizer.py:281-295: sequential restarts, first success wins
                if best_cycles[t] < 0 or f[j, r] < best_loss[t]:
                    best_loss[t], best_x[t], best_cycles[t] = float(f[j, r]), x[j, r].copy(), k
                self._span_losses[t, k - 1] = best_loss[t]
        self._set_stats([{"kernel_ms": 0.0, "kernel_launches": 0, "evals": evals, "items": [0] * len(evals), "total_ms": 0.0,
                          "kernel_ms_span": [0.0] * len(evals), "wave_rounds": [0] * len(evals), "evals_accepted": [0] * len(evals),
                          "evals_preempted": [0] * len(evals)}])
        return best_loss, best_x, best_cycles

    def _run_batch_any_order(self, targets: np.ndarray, ks):
        """A spanning range that is not a run k0, k0 + 1, ... (e.g. ``basis.spanning_range = [1, 3]``): the sizes are
        visited in the given order, one ``slam_decompose_list`` per size over the targets still unsolved -- the loop of
        optimizer.py:233-303 with the host between the spans."""
        n = len(targets)
        prm = self._opt_params()
        ctx = runtime.get_context(self.devices[0])
        if self._device_sampler is not None:
            self._device_sampler.fill(ctx)
        else:
            ctx.set_targets(targets)
        ctx.set_gates(self.basis.gate_matrices)
        ctx.set_cost(self._cost_kind)
        ctx.reset_stats()
        k_top = max(ks)
        best_loss = np.full(n, np.inf)
        best_x = np.zeros((n, 6 * (k_top + 1)))
        best_cycles = np.full(n, -1, dtype=np.int32)
        self._span_losses = np.full((n, _ffi.MAX_SPAN_EVAL), np.nan)
        for k in ks:
            if k <= 0:
                raise ValueError()  # CircuitTemplate.build(n_repetitions <= 0), basis.py:127-128
            todo = np.nonzero(~(best_loss < self.success_threshold))[0]
            if len(todo) == 0:
                break
            ctx.decompose_list(todo, k, k, [self.basis.gate_sequence(k)], prm, self.success_threshold, k_layout=k_top)
            loss, x, cyc = ctx.fetch_results_range(k_top, 0, n)
            better = np.zeros(n, bool)
            better[todo] = (best_cycles[todo] < 0) | (loss[todo] < best_loss[todo])  # optimizer.py:281-284
            best_loss[better], best_x[better], best_cycles[better] = loss[better], x[better], k
            self._span_losses[todo, k - 1] = best_loss[todo]
        self._set_stats([ctx.stats()])
        xs = [best_x[t, self.basis.param_slice(int(best_cycles[t]))].copy() for t in range(n)]
        return best_loss, xs, best_cycles

    def _run_batch_mixed_order(self, targets: np.ndarray, coords: np.ndarray):
        """``MixedOrderBasisCircuitTemplate``: the coverage entries (gate multisets) are visited in cost order; an entry is
        optimised over the targets still open that it can contain (polytope_wrap.py:78-90 with the membership test replaced by
        ``CircuitCoverage.inside`` -- and, where that is only an outer bound, by the optimisation itself).  One
        ``slam_decompose_list`` per entry; the batch stays resident.  ``self.circuit_polytopes[t]`` is the entry target t ended
        with (what the reference leaves bound in ``basis.circuit_polytope``)."""
        n = len(targets)
        local = np.max(np.abs(coords), axis=1) < 2e-8
        if np.any(local):
            raise ValueError()  # range(0, 1) -> CircuitTemplate.build(0), polytope_wrap.py:53-54, basis.py:127-128
        self.basis.minimal_spans(coords)  # raises for a target no entry can contain (polytope_wrap.py:91-93)
        entries = self.basis.candidate_entries(coords)
        k_top = max(len(e) for e, _, _ in entries)
        if k_top > _ffi.MAX_SPAN_MINIMIZE:
            raise NotImplementedError(f"template spans up to {_ffi.MAX_SPAN_MINIMIZE} are implemented on the HIP path (got {k_top})")
        prm = self._opt_params()
        ctx = runtime.get_context(self.devices[0])
        ctx.set_targets(targets)
        ctx.set_gates(self.basis.gate_matrices)
        ctx.set_cost(self._cost_kind)
        ctx.reset_stats()
        best_loss = np.full(n, np.inf)
        best_x = np.zeros((n, 6 * (k_top + 1)))
        best_cycles = np.full(n, -1, dtype=np.int32)
        best_entry = np.full(n, -1, dtype=np.int64)
        self._span_losses = None
        self._entries_tried = [[] for _ in range(n)]
        for j, (e, mask, _) in enumerate(entries):
            todo = np.nonzero(mask & ~(best_loss < self.success_threshold))[0]
            if len(todo) == 0:
                continue
            k = len(e)
            ctx.decompose_list(todo, k, k, [e.gate_indices], prm, self.success_threshold, k_layout=k_top)
            loss, x, _ = ctx.fetch_results_range(k_top, 0, n)
            better = np.zeros(n, bool)
            better[todo] = (best_cycles[todo] < 0) | (loss[todo] < best_loss[todo])  # optimizer.py:281-284
            best_loss[better], best_x[better], best_cycles[better], best_entry[better] = loss[better], x[better], k, j
            if self._want_span_losses:  # (only the log lines read it)
                for t in todo:
                    self._entries_tried[t].append((j, float(best_loss[t])))
        self._set_stats([ctx.stats()])
        self.circuit_polytopes = [entries[j][0] for j in best_entry]
        xs = [best_x[t, : 6 * (int(best_cycles[t]) + 1)].copy() for t in range(n)]
        return best_loss, xs, best_cycles

    def _run_batch_by_span(self, targets: np.ndarray, spans: np.ndarray):
        """Polytope mode: every target is optimised only at the template size ``spans[t]`` it needs.  The batch is
        resident once; each size is one ``slam_decompose_list`` call over the list of its targets."""
        if np.any(spans <= 0):
            raise ValueError()  # CircuitTemplate.build(n_repetitions <= 0), basis.py:127-128
        exact = getattr(self.basis, "span_rules_exact", True)
        # exact rules: every target runs at its own size only; lower bounds: the span loop runs from the bound to the template's maximum
        k_top = int(spans.max()) if exact else max(int(spans.max()), int(self.basis.maximum_span_guess))
        if k_top > _ffi.MAX_SPAN_MINIMIZE:
            raise NotImplementedError(f"template spans up to {_ffi.MAX_SPAN_MINIMIZE} are implemented on the HIP path (got {k_top})")
        prm = self._opt_params()
        ctx = runtime.get_context(self.devices[0])
        if self._device_sampler is not None:
            self._device_sampler.fill(ctx)
        else:
            ctx.set_targets(targets)
        ctx.set_gates(self.basis.gate_matrices)
        ctx.set_cost(self._cost_kind)
        ctx.reset_stats()
        for k in np.unique(spans):
            k = int(k)
            k_hi = k if exact else k_top
            ctx.decompose_list(np.nonzero(spans == k)[0], k, k_hi, [self.basis.gate_sequence(kk) for kk in range(k, k_hi + 1)], prm,
                               self.success_threshold, k_layout=k_top)
        best_loss, best_x, best_cycles = ctx.fetch_results_range(k_top, 0, len(targets))
        self._span_losses = ctx.fetch_span_losses(0, len(targets)) if self._want_span_losses else None
        self._set_stats([ctx.stats()])
        return best_loss, best_x, best_cycles  # (padded rows [n, 6 (k_top + 1)]: cut at 6 (cycles + 1) on access)

    def _run_batch_predicted(self, ctx, n: int):
        """Polytope mode for targets that were generated on the device (exact coverage regions): lookup, per-size lists and the span
        loop in one chain of kernels (``slam_decompose_predicted``) -- no index list is built on the host (round 4: ``np.nonzero`` per
        size + one ``slam_decompose_list`` each).  The reference's failures keep their exceptions: a target no prefix of the template
        reaches (polytope_wrap.py:91-93), a local target (``build(0)``, basis.py:127-128)."""
        kmax = int(self.basis.maximum_span_guess)
        self._device_sampler.fill(ctx)
        ctx.set_gates(self.basis.gate_matrices)
        ctx.set_cost(self._cost_kind)
        ctx.reset_stats()
        prm = self._opt_params()
        n_local, n_unreach = ctx.decompose_predicted([self.basis._gate_coords_all[i] for i in self.basis.gate_sequence(kmax)], kmax,
                                                     [self.basis.gate_sequence(k) for k in range(1, kmax + 1)], prm, self.success_threshold, 0, n)
        if n_unreach:
            raise ValueError("Monodromy did not find a polytope containing U")  # polytope_wrap.py:91-93
        if n_local:
            raise ValueError()  # CircuitTemplate.build(n_repetitions <= 0), basis.py:127-128
        best_loss, best_x, best_cycles = ctx.fetch_results_range(kmax, 0, n)
        self._span_losses = None
        self._set_stats([ctx.stats()])
        return best_loss, best_x, best_cycles

    def _run_batch_v2(self, targets: np.ndarray, spanning_range):
        """``_run`` for a CircuitTemplateV2 (optimizer.py:233-303 with method "L-BFGS-B" when the template has bounds, "BFGS"
        otherwise, :255-268): the span loop is driven from the host, one ``slam_v2_minimize_stage`` per template size over the
        targets still unsolved.  A stage's result is the restart the reference's sequential loop ends with (the first one
        below the threshold, else the lowest loss); restarts behind a successful one are not started (ordered early exit).
        Several ``devices``: contiguous target shards, one host thread + context each, seeds keyed on the global target index
        (``target_base``) -- the sharded job returns the single-device results bit for bit."""
        basis = self.basis
        n = len(targets)
        prm = self._opt_params()
        ks = [int(k) for k in spanning_range]
        for k in ks:
            if k <= 0:
                raise ValueError()  # build(n_repetitions <= 0), basisv2.py:221-222
            if k > _ffi.V2_MAX_SPAN:
                raise NotImplementedError(f"parametrised-gate templates run spans 1..{_ffi.V2_MAX_SPAN} on the HIP path (got {k})")
        layouts = {}
        for k in ks:
            basis.build(k)
            layouts[k] = (basis.gate_sequence(k),) + tuple(basis.device_layout(k))
        # set_constraint (optimizer.py:260-265 switches SciPy to SLSQP): the half-space of every span, for the device
        constraints = {k: basis.constraint_layout(k) for k in ks} if basis.using_constraints else {}

        def run_shard(device, first, count):
            single = len(self.devices) == 1
            ctx = runtime.get_context(device) if single else _ffi.Context(device)
            try:
                ctx.set_targets(targets[first : first + count])
                ctx.v2_set_gates(basis._gate_maps)
                for k, (w_dev, cost_max) in constraints.items():
                    ctx.v2_set_constraint(k, w_dev, cost_max)
                ctx.set_cost(self._cost_kind)
                ctx.reset_stats()
                sp = _ffi.OptParams(restarts=prm.restarts, maxiter=prm.maxiter, gtol=prm.gtol, stop_loss=prm.stop_loss, seed=prm.seed,
                                    flags=prm.flags, gtol_far=prm.gtol_far, far_loss=prm.far_loss, target_base=first)
                if ks == list(range(ks[0], ks[-1] + 1)):
                    # the usual case, a run of template sizes: the whole span loop as one chain of kernels on the device
                    try:
                        bl, bx, bc = ctx.v2_decompose_range(0, count, ks[0], ks[-1], [layouts[k][0] for k in ks],
                                                            [layouts[k][3:7] for k in ks], sp, self.success_threshold)
                    except _ffi.SlamHipError as exc:
                        if exc.code == -3:
                            raise NotImplementedError(str(exc)) from exc
                        raise
                    xs = [None] * count
                    for k in np.unique(bc):
                        idx = layouts[int(k)][2]
                        sel = np.nonzero(bc == k)[0]
                        block = np.ascontiguousarray(bx[sel][:, idx])  # user parameters (index order) of the span's device layout
                        for i, row in zip(sel.tolist(), block):
                            xs[i] = row
                    return (bl, xs, bc.astype(np.int32), ctx.fetch_span_losses(0, count)), ctx.stats()
                best = np.full(count, np.inf)
                best_x = [None] * count
                best_k = np.full(count, -1, dtype=np.int32)
                span_losses = np.full((count, _ffi.MAX_SPAN_EVAL), np.nan)
                for k in ks:
                    act = np.nonzero(~(best < self.success_threshold))[0].astype(np.int32)
                    if len(act) == 0:
                        break
                    seq, _, idx, init_lo, init_hi, blo, bhi = layouts[k]
                    try:
                        out = ctx.v2_minimize_stage(seq, sp, self.success_threshold, init_lo, init_hi, blo, bhi, active=act, want_items=False)
                    except _ffi.SlamHipError as exc:
                        if exc.code == -3:  # SLAM_ERR_UNSUPPORTED: span x parameters-per-gate beyond what the device kernels hold
                            raise NotImplementedError(str(exc)) from exc
                        raise
                    for j, t in enumerate(act):
                        if best_k[t] < 0 or out["best_loss"][j] < best[t]:  # optimizer.py:281-284
                            best[t], best_k[t] = out["best_loss"][j], k
                            best_x[t] = out["best_x"][j][idx].copy()  # user parameters (index order)
                        span_losses[t, k - 1] = best[t]
                return (best, best_x, best_k, span_losses), ctx.stats()
            finally:
                if not single:
                    ctx.close()

        if len(self.devices) == 1 or n < len(self.devices):
            (best, best_x, best_k, self._span_losses), st = run_shard(self.devices[0], 0, n)
            self._set_stats([st])
        else:
            import threading

            from .parallel import shard_range

            parts = [None] * len(self.devices)
            errors = []

            def work(r):
                try:
                    first, count = shard_range(n, r, len(self.devices))
                    parts[r] = run_shard(self.devices[r], first, count)
                except Exception as exc:  # surfaced below
                    errors.append(exc)

            threads = [threading.Thread(target=work, args=(r,)) for r in range(len(self.devices))]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            if errors:
                raise errors[0]
            best = np.concatenate([p[0][0] for p in parts])
            best_x = [x for p in parts for x in p[0][1]]
            best_k = np.concatenate([p[0][2] for p in parts])
            self._span_losses = np.concatenate([p[0][3] for p in parts])
            self._set_stats([p[1] for p in parts])
        if np.any(best_k < 0):
            raise ValueError("empty spanning range")
        return best, best_x, best_k

    def _run_batch_callback(self, targets: np.ndarray, spans_per_target):
        """``_run`` with ``use_callback=True`` (optimizer.py:217-224,238,287-292): the span loop is driven from the host,
        one ``slam_minimize_stage_trace`` per template size, so that the loss and the point after every quasi-Newton
        iteration of every restart come back.  With the ordered early exit every restart up to the first successful
        one runs to its end, exactly the restarts the reference's sequential loop executes; what it would never have
        started (restarts after the first success) is left out of the record.  Meant for a handful of targets.

        Returns (best_loss, best_xs, best_cycles); fills ``self._callback_records[t]`` with the list of
        (temp_training_loss, temp_coordinate_list) pairs the reference appends for target t."""
        n = len(targets)
        prm = self._opt_params()
        ctx = runtime.get_context(self.devices[0])
        ctx.set_targets(targets)
        if self._v2:
            ctx.v2_set_gates(self.basis._gate_maps)
            if self.basis.using_constraints:
                for k in sorted({int(k) for ks in spans_per_target for k in ks}):
                    ctx.v2_set_constraint(k, *self.basis.constraint_layout(k))
        else:
            ctx.set_gates(self.basis.gate_matrices)
        ctx.set_cost(self._cost_kind)
        ctx.reset_stats()
        R = int(self.training_restarts)
        best = [None] * n
        best_x = [None] * n
        best_k = [-1] * n
        temp_loss = [[] for _ in range(n)]  # one list per target, growing over the spans (optimizer.py:229,238)
        self._callback_records = [[] for _ in range(n)]
        self._span_losses = np.full((n, _ffi.MAX_SPAN_EVAL), np.nan)
        all_ks = sorted({int(k) for ks in spans_per_target for k in ks})
        for k in all_ks:
            if k <= 0:
                raise ValueError()  # CircuitTemplate.build(n_repetitions <= 0), basis.py:127-128
            k_lim = _ffi.V2_MAX_SPAN if self._v2 else _ffi.MAX_SPAN_MINIMIZE  # (per-iteration traces: both kernel families)
            if k > k_lim:
                raise NotImplementedError(f"template spans up to {k_lim} are implemented on the HIP path (got {k})")
            act = np.array([t for t in range(n) if k in spans_per_target[t] and not (best[t] is not None and best[t] < self.success_threshold)],
                           dtype=np.int32)
            if len(act) == 0:
                continue
            seq = self.basis.gate_sequence(k)
            v2_idx = None
            if self._v2:
                self.basis.build(k)
                _, v2_idx, v2_ilo, v2_ihi, v2_blo, v2_bhi = self.basis.device_layout(k)
            cap = 256
            while True:
                if self._v2:
                    out = ctx.v2_minimize_stage_trace(seq, prm, self.success_threshold, cap, v2_ilo, v2_ihi, v2_blo, v2_bhi, active=act)
                else:
                    out = ctx.minimize_stage_trace(seq, prm, self.success_threshold, cap, active=act)
                need = int(out["item_iters"].max())
                if need <= cap:
                    break
                cap = need  # deterministic: the same run again, now with room for the longest restart

            def coords_of(X):
                """c1c2c3 of the template unitaries of the traced points (optimizer.py:223-224)"""
                if not len(X):
                    return []
                if self._v2:
                    _, _, W = ctx.v2_eval(seq, X, want_grad=False, want_unitary=True)  # X is in device layout already
                    C3 = ctx.c1c2c3(W)
                else:
                    C3 = ctx.eval_c1c2c3(seq, X)
                return [tuple(float(v) for v in c) for c in C3]
            for j, t in enumerate(act):
                temp_loss[t].extend([-1, k])  # flags for the plotting function (optimizer.py:238)
                rows = []  # (restart, iterations) of the restarts the sequential loop runs
                for r in range(R):
                    it = int(out["item_iters"][j, r])
                    temp_loss[t].extend(float(v) for v in out["trace_loss"][j, r, :it])
                    rows.append((r, it))
                    result = float(out["item_loss"][j, r])
                    if best[t] is None or result < best[t]:  # optimizer.py:281-284
                        best[t], best_k[t] = result, k
                    hit = best[t] < self.success_threshold
                    if hit or (self.override_fail and r == R - 1):
                        X = np.concatenate([out["trace_x"][j, rr, :ii] for rr, ii in rows]) if rows else np.zeros((0, out["trace_x"].shape[-1]))
                        coords = coords_of(X)
                        # the reference appends the SAME growing list object every time (optimizer.py:289-292)
                        self._callback_records[t].append((temp_loss[t], coords))
                    if hit:
                        break
                if best_k[t] == k:
                    # the sequential best of this span is the stage's ordered winner: the first restart below the
                    # threshold, else the lowest loss (first occurrence)
                    assert best[t] == float(out["best_loss"][j])
                    best_x[t] = out["best_x"][j][v2_idx].copy() if self._v2 else out["best_x"][j].copy()
                self._span_losses[t, k - 1] = best[t]
        self._set_stats([ctx.stats()])
        if any(b is None for b in best):
            raise ValueError("empty spanning range")
        return np.array(best), best_x, np.array(best_k, dtype=np.int32)

    def _log_span_loop(self, i: int, spans) -> None:
        """The per-span log lines of ``_run`` (optimizer.py:234,297,302) for target i, from the running best loss
        the device recorded after every span."""
        if getattr(self.basis, "mixed_order", False):
            for j, v in self._entries_tried[i]:
                k = len(self.basis.coverage[j])
                logging.info(f"Starting opt on template size {k}")
                logging.info(f"Cycle (k ={k}), Best Loss={v}")
                if v < self.success_threshold:
                    logging.info(f"Break on cycle {k}")
                    break
            return
        sl = getattr(self, "_span_losses", None)
        if sl is None:
            return
        for k in spans:
            v = sl[i, k - 1] if 1 <= k <= sl.shape[1] else np.nan
            if np.isnan(v):
                continue
            logging.info(f"Starting opt on template size {k}")
            logging.info(f"Cycle (k ={k}), Best Loss={float(v)}")
            if v < self.success_threshold:
                logging.info(f"Break on cycle {k}")
                break

    def _finish_target(self, target_coordinates, best_result, best_Xk, best_cycles, found_coordinates, index=None) -> DataDictEntry:
        """Labelling / logging / exception of approximate_target_U (optimizer.py:80-119)."""
        logging.info(f"Overall Best Loss={best_result}")
        if self.use_callback:
            # optimizer.py:287-292: one (training_loss, coordinate_list) entry per firing of the break condition
            for tl, cl in self._callback_records[index]:
                self.training_loss.append(tl)
                self.coordinate_list.append(cl)
        else:
            self.training_loss.append(best_result)  # optimizer.py:307-309 (no callback)
        self.best_cycle_list.append(best_cycles)
        if best_result <= self.success_threshold:
            success_label = 1
            logging.info(f"Success: {target_coordinates}, Found: {found_coordinates}")
        else:
            if not self.override_fail:
                raise ValueError(_FAIL_MSG)
            success_label = 0
            logging.info(f"Fail: {target_coordinates}, Found: {found_coordinates}")
        return DataDictEntry(success_label, best_result, best_Xk, best_cycles)

    def _row_slice(self, cycles: int) -> slice:
        """The entry's ``Xk`` inside a padded device row [6 (k_max + 1)]: the first 6 (cycles + 1) values, or -- no_exterior_1q --
        the interior layers (basis.param_slice)."""
        if self._v2:
            return slice(0, 6 * (cycles + 1))
        return self.basis.param_slice(cycles)

    def _found_coordinates(self, best_xs, best_cycles) -> np.ndarray:
        """c1c2c3 of the found circuits (optimizer.py:85,103): one batched CircuitTemplate.eval on the GPU
        per distinct span, the Weyl coordinates of the template unitaries on the device too (``slam_eval_c1c2c3``)."""
        n = len(best_xs)
        found = np.zeros((n, 3))
        ctx = runtime.get_context(self.devices[0])
        if self._v2:
            for k in np.unique(best_cycles):
                idx = np.nonzero(best_cycles == k)[0]
                self.basis.build(int(k))
                X = self.basis.to_device_vector(np.stack([best_xs[i] for i in idx]), int(k))
                _, _, W = ctx.v2_eval(self.basis.gate_sequence(int(k)), X, want_grad=False, want_unitary=True)
                found[idx] = ctx.c1c2c3(W)
            return found
        ctx.set_gates(self.basis.gate_matrices)
        if ctx.n_targets == 0:
            ctx.set_targets(np.eye(4, dtype=np.complex128)[None])
        if getattr(self.basis, "mixed_order", False):
            ids = np.array([id(e) for e in self.circuit_polytopes])
            for v in np.unique(ids):
                idx = np.nonzero(ids == v)[0]
                found[idx] = ctx.eval_c1c2c3(self.circuit_polytopes[idx[0]].gate_indices, np.stack([best_xs[i] for i in idx]))
            return found
        for k in np.unique(best_cycles):
            idx = np.nonzero(best_cycles == k)[0]
            X = self.basis.device_vector(np.stack([best_xs[i] for i in idx]), int(k))
            # template unitary and its Weyl coordinates on the device: three doubles per circuit come back
            found[idx] = ctx.eval_c1c2c3(self.basis.gate_sequence(int(k)), X)
        return found

    # ------------------------------------------------------------------------------------------
    def approximate_target_U(self, target_U) -> DataDictEntry:
        """Atomic training function (optimizer.py:65-119)."""
        t = np.asarray(target_U, dtype=np.complex128)
        if t.shape != (4, 4):
            raise ValueError("targets must be 4x4 unitaries")
        return self._approximate_batch(t[None], log_index=False)[0]

    def approximate_from_distribution(self, sampler: SampleFunction):
        """optimizer.py:180-186; all targets of the sampler are optimised as one GPU batch."""
        self._device_sampler = sampler if hasattr(sampler, "fill") else None  # sampler.DeviceHaarBatch
        try:
            if self._device_sampler is not None:
                # the targets stay where they were generated; they come over (one array) only if something on the host
                # looks at them: log lines, polytope mode, callbacks, several devices' shards, V2 templates
                stacked = _ResidentTargets(sampler)
            else:
                targets = [np.asarray(t, dtype=np.complex128) for t in sampler]
                for t in targets:
                    if t.shape != (4, 4):
                        raise ValueError("targets must be 4x4 unitaries")
                stacked = np.stack(targets) if targets else np.zeros((0, 4, 4), dtype=np.complex128)
            target_data = self._approximate_batch(stacked, log_index=True) if len(stacked) else []
        finally:
            self._device_sampler = None
        return self.training_loss, self.coordinate_list, target_data

    def _approximate_batch(self, stacked, log_index: bool):
        n = len(stacked)
        # The reference logs per target (optimizer.py:77-106,183,234,297-305).  Formatting ~10 lines for each of 1e5
        # targets costs 50x the GPU time of the batch, so the log lines -- and the coordinates that only they show -- are
        # produced only when INFO logging is enabled.
        log_on = logging.getLogger().isEnabledFor(logging.INFO)
        self._want_span_losses = log_on
        ctx0 = runtime.get_context(self.devices[0])
        need_coords = log_on or (self.basis.use_polytopes and not self._v2)
        fast = (not need_coords and not self.use_callback and not self._v2 and not self.basis.use_polytopes and self._host_method is None)
        # use_polytopes with targets that were generated on the device: the template-size lookup runs there too (slam_predict_spans
        # evaluates the coverage half-spaces of coverage.py on the resident targets) -- neither targets nor coordinates come back
        poly_resident = (self.basis.use_polytopes and not self._v2 and not log_on and not self.use_callback and self._host_method is None
                         and not getattr(self.basis, "mixed_order", False) and isinstance(stacked, _ResidentTargets)
                         and len(self.devices) == 1 and getattr(self.basis, "span_rules_exact", False)
                         and int(self.basis.maximum_span_guess) <= _ffi.MAX_SPAN_EVAL)
        if poly_resident:
            need_coords = False
        elif not fast and isinstance(stacked, _ResidentTargets):
            stacked = stacked.as_array()
        # target_invariant (basis_abc.py:80-84) for the whole batch, on the device
        coords_arr = ctx0.c1c2c3(stacked) if need_coords else None
        self.basis.assign_seed(None)  # optimizer.py:150-152
        spans_of = None
        if self._host_method is not None:
            spanning_range = list(self.basis.get_spanning_range(stacked[0]))
            spans_of = [spanning_range] * n
            best_loss, best_xs, best_cycles = self._run_batch_host_method(stacked, spanning_range)
        elif self._v2 and not self.use_callback and self.basis.use_polytopes:
            # basisv2.py:77-85: every target at the size its coverage region assigns; one run per size over that size's targets
            spans = np.asarray(self.basis.minimal_spans(ctx0.c1c2c3(np.asarray(stacked))), dtype=np.int64)
            if np.any(spans <= 0):
                raise ValueError()  # build(n_repetitions <= 0), basisv2.py:221-222
            spans_of = [[int(k)] for k in spans]
            best_loss = np.full(n, np.inf)
            best_xs = [None] * n
            best_cycles = np.full(n, -1, dtype=np.int32)
            sl_all = np.full((n, _ffi.MAX_SPAN_EVAL), np.nan)
            stats = []
            for k in np.unique(spans):
                sel = np.nonzero(spans == k)[0]
                bl, bx, bc = self._run_batch_v2(np.asarray(stacked)[sel], [int(k)])
                best_loss[sel], best_cycles[sel] = bl, bc
                for i, x in zip(sel.tolist(), bx):
                    best_xs[i] = x
                if self._span_losses is not None:
                    sl_all[sel] = self._span_losses
                stats.extend(self.last_stats_per_device)
            self._span_losses = sl_all
            self._set_stats(stats)
        elif self._v2 and not self.use_callback:
            spanning_range = list(self.basis.get_spanning_range(stacked[0]))
            spans_of = [spanning_range] * n
            best_loss, best_xs, best_cycles = self._run_batch_v2(stacked, spanning_range)
        elif self.use_callback:
            if self.basis.use_polytopes:
                if coords_arr is None:
                    coords_arr = ctx0.c1c2c3(np.asarray(stacked))
                top = None if getattr(self.basis, "span_rules_exact", True) else int(self.basis.maximum_span_guess)
                spans_of = [list(range(int(k), (int(k) if top is None else top) + 1)) for k in self.basis.minimal_spans(coords_arr)]
            else:
                spans_of = [list(self.basis.get_spanning_range(stacked[0]))] * n
            best_loss, best_xs, best_cycles = self._run_batch_callback(stacked, spans_of)
        elif poly_resident:
            best_loss, best_xs, best_cycles = self._run_batch_predicted(ctx0, n)
        elif getattr(self.basis, "mixed_order", False):
            best_loss, best_xs, best_cycles = self._run_batch_mixed_order(stacked, coords_arr)
            spans_of = [None] * n  # (the log lines come from _entries_tried)
        elif self.basis.use_polytopes:
            # get_spanning_range per target (optimizer.py:233 with basis.py:95-100): only the template size
            # the target needs.  Targets are grouped by that size; each group is one batch.
            spans = self.basis.minimal_spans(coords_arr)
            if getattr(self.basis, "span_rules_exact", True):
                spans_of = [[int(k)] for k in spans]
            else:
                spans_of = [list(range(int(k), int(self.basis.maximum_span_guess) + 1)) for k in spans]
            best_loss, best_xs, best_cycles = self._run_batch_by_span(stacked, spans)
        else:
            spanning_range = self.basis.spanning_range  # (get_spanning_range without polytopes: the brute-force range, basis.py:95-100)
            if not fast:
                spans_of = [list(spanning_range)] * n
            best_loss, best_xs, best_cycles = self._run_batch(stacked, spanning_range)
        best_loss = np.asarray(best_loss, dtype=np.float64)
        best_cycles = np.asarray(best_cycles)
        if getattr(self.basis, "mixed_order", False):
            self.basis.set_polytope(self.circuit_polytopes[-1])
        self.basis.build(n_repetitions=int(best_cycles[-1]))  # the reference leaves the template at the last size
        if isinstance(best_xs, RowBlocks) and (log_on or self.use_callback):
            best_xs = best_xs.as_array()
        padded = isinstance(best_xs, (np.ndarray, RowBlocks)) and best_xs.ndim == 2  # [n, 6 (k_max + 1)] rows, cut at 6 (cycles + 1) on access
        if not log_on and not self.use_callback:
            # same bookkeeping as the per-target path below, without the log lines: every target up to (and including)
            # the first one that fails without override_fail is recorded, then the reference's ValueError (optimizer.py:89-93)
            ok = best_loss <= self.success_threshold
            fail = (not self.override_fail) and (not bool(ok.all()))
            stop = int(np.argmin(ok)) + 1 if fail else n
            self.training_loss.extend_array(best_loss[:stop])  # optimizer.py:307-309 (no callback): elements materialise on access
            self.best_cycle_list.extend_array(best_cycles[:stop])
            if fail:
                raise ValueError(_FAIL_MSG)
            # list of DataDictEntry (optimizer.py:113), entries built when they are looked at
            return TargetDataList(ok, best_loss, best_xs, best_cycles, self._row_slice if padded else None)
        if padded:
            best_xs = [best_xs[i, self._row_slice(int(best_cycles[i]))] for i in range(n)]
        found = self._found_coordinates(best_xs, best_cycles) if log_on else np.zeros((n, 3))
        coords = coords_arr if coords_arr is not None else np.zeros((n, 3))
        out = []
        for i in range(n):
            tc = tuple(float(v) for v in coords[i])
            if log_index:
                logging.info(f"Starting sample iter {i}")
            logging.info(f"Begin search: {tc}")
            self._log_span_loop(i, spans_of[i])
            fc = tuple(float(v) for v in found[i])
            out.append(self._finish_target(tc, float(best_loss[i]), best_xs[i], int(best_cycles[i]), fc, index=i))
        return out


class _ResidentTargets:
    """The targets of a device sampler (sampler.DeviceHaarBatch) as the optimizer sees them: a length, and the array only
    when somebody asks for it (``as_array`` copies the batch back once and caches it in the sampler)."""

    def __init__(self, sampler):
        self._sampler = sampler

    def __len__(self):
        return int(self._sampler.n_samples)

    def as_array(self) -> np.ndarray:
        return np.asarray(self._sampler.as_array(), dtype=np.complex128)

    def __getitem__(self, i):
        return self.as_array()[i]
