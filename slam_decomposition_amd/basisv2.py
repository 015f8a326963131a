"""``CircuitTemplateV2`` (reference: src/slam/basisv2.py:27-299) without qiskit.

The V2 template passes 2Q gate *classes / lambdas* instead of gate objects: every gate instance of the circuit gets
its own optimisable "Q" parameters next to the "P" parameters of the U gates (basisv2.py:262-287), optionally
box-bounded (``add_bound``, basisv2.py:174-190 -- the reference's optimizer then switches SciPy to L-BFGS-B,
optimizer.py:255-268).  The arithmetic -- template unitary, loss, gradient with respect to P *and* Q parameters,
the projected quasi-Newton loop -- runs in libslamhip (``slam_v2_*``, csrc/slam_v2.hpp).

Supported gate callables: anything that, called with its parameters, returns a member of the conversion-gain family
with every raw angle an affine function of at most one parameter -- ``RiSwapGate`` (alpha), lambdas over
``ConversionGainGate(phi_c, phi_g, gc, gg, t)`` with ``t`` fixed (e.g. ``lambda gc, gg: ConversionGainGate(0, 0, gc, gg, 1)``,
``lambda p1, p2: ConversionGainGate(p1, p2, g1, g2, t)`` of utils/gates/family_extend.py:40-47).  The map is found by
probing the callable; anything else raises ``NotImplementedError``.

Parameter order of ``Xk`` here is *index order*: ``P0 .. P{m-1}, Q0 .. Q{QN k - 1}`` (the reference zips ``Xk`` with
qiskit's name-sorted ``circuit.parameters``; ``to_qiskit_order`` / ``from_qiskit_order`` convert).  ``no_exterior_1q`` (no 1Q
layer before the first and after the last 2Q gate, basisv2.py:264,291) maps onto the device template by fixing those two
layers at ``U(0, 0, 0)`` = identity, like ``vz_only`` fixes theta and phi.  ``set_constraint(c)`` (basisv2.py:192-200:
``circuit_cost(x) <= c``, which the reference hands to SciPy's SLSQP, optimizer.py:260-265) is implemented for circuit costs that
are affine in the parameters over the box -- ``RiSwapGate`` (cost = alpha), ``ConversionGainGate`` lambdas whose drive strengths
are bounded to one sign -- as a projected quasi-Newton method on the box cut by the half-space (slam_v2_set_constraint).
Not implemented: ``param_vec_expand`` (time-sliced "smush" gates: not conversion-gain family members), polytopes.
"""
from __future__ import annotations

from inspect import signature
from typing import List, Sequence

import numpy as np

from . import _ffi, runtime
from .basis_abc import VariationalTemplate
from .gates import RiSwapGate

DEFAULT_BOUND = (-4 * np.pi, 4 * np.pi)  # parameter_guess default range, basisv2.py:160


def _raw_angles(gate) -> np.ndarray:
    """(a, phi_c, b, phi_g) of a conversion-gain family member."""
    name = type(gate).__name__
    prm = [float(v) for v in getattr(gate, "params", [])]
    if name in ("RiSwapGate", "iSwapGate") or getattr(gate, "label", None) == "riswap":
        return np.array([-0.5 * np.pi * prm[0], 0.0, 0.0, 0.0])  # RiSwap(alpha) = CG(0, 0, -pi alpha / 2, 0, 1)
    if name == "ConversionGainGate" and len(prm) == 5:
        p1, p2, g1, g2, t = prm
        return np.array([g1 * t, p1, g2 * t, p2])
    raise NotImplementedError(f"parametrised gate {name}: only conversion-gain family gates (RiSwapGate, ConversionGainGate) run on the HIP path")


def gate_map(gate_fn):
    """Probe ``gate_fn`` for raw[r] = scale[r] * q[sel[r]] + offset[r].  Returns (n_params, sel, scale, offset)."""
    qn = len(signature(gate_fn).parameters)  # basisv2.py:268-273
    if not 1 <= qn <= 4:
        raise NotImplementedError(f"parametrised gates with {qn} parameters: 1..4 are supported (fix the pulse time t with a lambda)")
    rng = np.random.default_rng(12345)
    q0 = rng.uniform(0.2, 1.2, qn)
    r0 = _raw_angles(gate_fn(*q0))
    M = np.zeros((4, qn))
    for m in range(qn):
        e = np.zeros(qn)
        e[m] = 0.37
        M[:, m] = (_raw_angles(gate_fn(*(q0 + e))) - r0) / 0.37
    q1 = rng.uniform(-2.0, 2.0, qn)
    if not np.allclose(_raw_angles(gate_fn(*q1)), r0 + M @ (q1 - q0), atol=1e-9):
        raise NotImplementedError("parametrised gate: the conversion-gain angles must be affine in the parameters (fix t with a lambda)")
    sel, scale = [-1] * 4, [0.0] * 4
    for r in range(4):
        nz = np.nonzero(np.abs(M[r]) > 1e-12)[0]
        if len(nz) > 1:
            raise NotImplementedError("parametrised gate: each conversion-gain angle may depend on one parameter only")
        if len(nz) == 1:
            sel[r], scale[r] = int(nz[0]), float(M[r, nz[0]])
    offset = r0 - M @ q0
    offset[np.abs(offset) < 1e-15] = 0.0
    return qn, sel, scale, [float(v) for v in offset]


class CircuitTemplateV2(VariationalTemplate):
    def __init__(
        self,
        n_qubits=2,
        base_gates=None,
        edge_params=None,
        no_exterior_1q=False,
        use_polytopes=False,
        maximum_span_guess=5,
        preseed=False,
        vz_only=False,
        param_vec_expand=None,
        device=0,
    ):
        if base_gates is None:
            base_gates = [RiSwapGate]  # basisv2.py:31
        if edge_params is None:
            edge_params = [[(0, 1)]]
        if n_qubits != 2:
            raise NotImplementedError("the HIP template optimizer handles 2-qubit templates only")
        if param_vec_expand is not None:
            raise NotImplementedError("param_vec_expand (time-sliced smush gates, basisv2.py:47-50) is not implemented on the HIP path")
        for el in edge_params:
            for e in el:
                if tuple(e) != (0, 1):
                    raise NotImplementedError("only edge (0, 1) is implemented on the HIP path")
        self.filename = None
        self.n_qubits = n_qubits
        self.no_exterior_1q = bool(no_exterior_1q)
        self.param_vec_expand = None
        self.vz_only = bool(vz_only)
        self.base_gates = list(base_gates)
        self.edge_params = edge_params
        self.device = device
        maps = [gate_map(g) for g in self.base_gates]
        qns = {m[0] for m in maps}
        if len(qns) != 1:
            raise NotImplementedError("all parametrised base gates of a template must take the same number of parameters")
        self.n_gate_params = qns.pop()
        # device gates take 1, 2 or 4 parameters: a 3-parameter gate gets a fixed dummy fourth
        self._dev_qn = 4 if self.n_gate_params == 3 else self.n_gate_params
        self._gate_maps = [_ffi.V2Gate(self._dev_qn, m[1], m[2], m[3]) for m in maps]
        # XXX (basisv2.py:60-65)
        self.bounds = {}  # parameter name -> (min, max)
        self.bounds_list = []
        self.constraint_func = None
        self.using_bounds = False
        self.using_constraints = False
        # basisv2.py:66-70: the brute-force range exists only without polytopes.  With them (basisv2.py:77-85 ->
        # monodromy_range_from_target) the template size comes from the coverage set: here coverage.py's regions of the circuit the
        # bounds FIX (every gate parameter bounded to a point) -- see get_spanning_range
        self.spanning_range = None if use_polytopes else range(1, maximum_span_guess + 1)
        self.maximum_span_guess = maximum_span_guess
        self.coverage = None
        super().__init__(preseed=preseed, use_polytopes=use_polytopes)
        self._reset()
        self.trotter = False

    # ---- structure ---------------------------------------------------------------------------
    def _reset(self):
        self.cycles = 0

    def build(self, n_repetitions):
        """basisv2.py:215-229 (the gate cycle restarts at every build, as in CircuitTemplate here)."""
        self._reset()
        if n_repetitions <= 0:
            raise ValueError()
        if n_repetitions > _ffi.V2_MAX_SPAN:
            raise NotImplementedError(f"parametrised-gate templates run spans 1..{_ffi.V2_MAX_SPAN} on the HIP path")
        self.cycles = int(n_repetitions)

    def gate_sequence(self, k=None) -> List[int]:
        k = self.cycles if k is None else k
        return [i % len(self.base_gates) for i in range(k)]

    def get_spanning_range(self, target_u):
        """basisv2.py:77-85.  Without polytopes: the brute-force range.  With them: ``range(k, k + 1)`` for the smallest template that
        reaches the target (polytope_wrap.py:39-94).  The reference reads that off a precomputed ``coverage`` list the caller has to
        attach; here the regions come from coverage.py, which describes circuits of FIXED gates -- so every gate parameter must be
        bounded to a point (``add_bound(name, max=v, min=v)``); a template whose gates are still free has no single monodromy
        polytope per size and raises."""
        if not self.use_polytopes:
            return self.spanning_range
        from .weyl import c1c2c3

        k = int(self.minimal_spans(np.array([c1c2c3(target_u)]))[0])
        return range(k, k + 1)

    def fixed_gate_coordinates(self, k_max=None):
        """Weyl coordinates of the template's gates G_1 .. G_kmax when the bounds fix them (min == max for every gate parameter)."""
        from .gates import gate_matrix
        from .weyl import c1c2c3

        k_max = int(self.maximum_span_guess) if k_max is None else int(k_max)
        qn = self.n_gate_params
        coords = []
        for j, gi in enumerate(self.gate_sequence(k_max)):
            vals = []
            for m in range(qn):
                b = self.bounds.get(f"Q{qn * j + m}")
                if b is None or b[0] is None or b[1] is None or float(b[0]) != float(b[1]):
                    raise NotImplementedError(
                        f"CircuitTemplateV2(use_polytopes=True): gate parameter Q{qn * j + m} is free -- the coverage regions (coverage.py) "
                        "describe circuits of fixed gates; bound every gate parameter to a point (add_bound(name, max=v, min=v)), or attach "
                        "nothing and run without polytopes")
                vals.append(float(b[0]))
            coords.append(c1c2c3(gate_matrix(self.base_gates[gi](*vals))))
        return coords

    def minimal_spans(self, target_coords) -> np.ndarray:
        """Template size per target from the exact coverage regions of the fixed-gate circuit; raises like the reference's lookup
        for a target out of reach (polytope_wrap.py:91-93)."""
        if not self.use_polytopes:
            raise ValueError("minimal_spans needs use_polytopes=True")
        from . import coverage

        kmax = int(self.maximum_span_guess)
        saved = self.cycles
        try:
            self.cycles = kmax  # (add_bound names refer to the longest template)
            k = coverage.minimal_prefix(np.asarray(target_coords, dtype=np.float64).reshape(-1, 3), self.fixed_gate_coordinates(kmax), kmax)
        finally:
            self.cycles = saved
        if np.any(k > kmax):
            raise ValueError("Monodromy did not find a polytope containing U")  # polytope_wrap.py:91-93
        return k

    def _n_p(self, k=None) -> int:
        k = self.cycles if k is None else k
        # rz: one parameter per qubit and layer (basisv2.py:256-259); no_exterior_1q: the layers before the first and after the
        # last 2Q gate do not exist (basisv2.py:264,291): k - 1 interior layers
        layers = (k - 1) if self.no_exterior_1q else (k + 1)
        return (2 if self.vz_only else 6) * max(layers, 0)

    def parameter_names(self, k=None) -> List[str]:
        """Index order: P0.., then Q0.. (gate 1's parameters first)."""
        k = self.cycles if k is None else k
        return [f"P{i}" for i in range(self._n_p(k))] + [f"Q{i}" for i in range(self.n_gate_params * k)]

    @property
    def n_params(self) -> int:
        return len(self.parameter_names())

    def to_qiskit_order(self, Xk, k=None):
        """Index order -> the order of qiskit's name-sorted ``circuit.parameters`` the reference zips ``Xk`` with."""
        names = self.parameter_names(k)
        order = sorted(range(len(names)), key=lambda i: names[i])
        return np.asarray(Xk, dtype=np.float64)[order]

    def from_qiskit_order(self, Xk, k=None):
        names = self.parameter_names(k)
        order = sorted(range(len(names)), key=lambda i: names[i])
        out = np.empty(len(names))
        out[order] = np.asarray(Xk, dtype=np.float64)
        return out

    # ---- bounds (basisv2.py:174-190) ------------------------------------------------------------
    def add_bound(self, parameter_name, max=None, min=None):
        if parameter_name not in self.parameter_names():
            raise ValueError("Parameter Name not found")
        self.bounds[parameter_name] = (min, max)
        self.using_bounds = True

    def set_constraint(self, param_max_cost):
        """basisv2.py:192-200: the current basis must not cost more than ``param_max_cost`` (C(x) >= 0 form for SciPy)."""
        self.constraint_func = {"type": "ineq", "fun": lambda x: param_max_cost - self.circuit_cost(x)}
        self.param_max_cost = float(param_max_cost)
        self.using_constraints = True

    def remove_constraint(self):
        self.constraint_func = None
        self.using_constraints = False

    def constraint_layout(self, k: int):
        """The constraint ``circuit_cost(x) <= param_max_cost`` of the span-k template as (weights[n_dev], cost_max) with
        sum_i weights[i] x_dev[i] <= cost_max, for the device (slam_v2_set_constraint).  The circuit cost is probed: it has to be
        affine in the gate parameters over the box (RiSwapGate: alpha; ConversionGainGate: (|gc| + |gg|) t / (pi / 2) with
        gc, gg bounded to one sign); otherwise NotImplementedError.  ValueError when no point of the box satisfies it."""
        if not self.using_constraints:
            raise ValueError("no constraint set")
        saved = self.cycles
        try:
            self.cycles = int(k)
            n_dev, idx, _, _, blo, bhi = self.device_layout(k)
            n_p, n_q = self._n_p(k), self.n_gate_params * k
            lo = np.array([blo[d] for d in idx[n_p:]])
            hi = np.array([bhi[d] for d in idx[n_p:]])
            plo = np.where(np.isfinite(lo), lo, np.where(np.isfinite(hi), hi - 2.0, -1.0))  # a finite probing box
            phi = np.where(np.isfinite(hi), hi, plo + 2.0)
            phi = np.where(phi > plo, phi, plo)
            zeros_p = np.zeros(n_p)
            cost = lambda qv: float(self.circuit_cost(np.concatenate([zeros_p, qv])))
            q0 = 0.5 * (plo + phi)
            c_q0 = cost(q0)
            w = np.zeros(n_q)
            for j in range(n_q):
                h = 0.25 * (phi[j] - plo[j])
                if h <= 0.0:
                    continue  # a parameter fixed by its bound: constant contribution, inside c_q0
                e = np.zeros(n_q)
                e[j] = h
                w[j] = (cost(q0 + e) - c_q0) / h
            rng = np.random.default_rng(2024)
            for _ in range(8):
                qv = rng.uniform(plo, phi)
                if abs(cost(qv) - (c_q0 + w @ (qv - q0))) > 1e-9 * (1.0 + abs(c_q0)):
                    raise NotImplementedError(
                        "set_constraint on the HIP path needs a circuit cost that is affine in the gate parameters over their bounds "
                        "(ConversionGainGate: bound gc and gg to one sign with add_bound)")
        finally:
            self.cycles = saved
        w_dev = np.zeros(n_dev)
        w_dev[idx[n_p:]] = w
        cost_max = self.param_max_cost - (c_q0 - w @ q0)
        cheapest = float(np.sum(np.where(w > 0, w * lo, np.where(w < 0, w * hi, 0.0))))  # -inf when an unbounded side is cheap
        if cheapest > cost_max + 1e-12:
            raise ValueError(f"set_constraint({self.param_max_cost}): the cheapest circuit inside the bounds costs {cheapest + (c_q0 - w @ q0):.6g}")
        if not np.any(w != 0.0):
            raise ValueError("set_constraint: the circuit cost does not depend on any free parameter")
        return w_dev, float(cost_max)

    def parameter_guess(self, t=0):
        """basisv2.py:150-172: uniform in the parameter's bound, (-4 pi, 4 pi) by default; also (re)builds bounds_list."""
        parent = super().parameter_guess(t)
        if parent is not None:
            return parent
        random_list, self.bounds_list = [], []
        for name in self.parameter_names():
            cbound = self.bounds.get(name, DEFAULT_BOUND)
            self.bounds_list.append(cbound)
            lo, hi = (DEFAULT_BOUND if cbound is None else cbound)
            lo = DEFAULT_BOUND[0] if lo is None else lo
            hi = DEFAULT_BOUND[1] if hi is None else hi
            random_list.append(np.random.uniform(lo, hi, 1)[0])
        if not self.using_bounds:
            self.bounds_list = None  # so the optimizer can use BFGS (basisv2.py:170-171)
        return random_list

    # ---- user vector <-> device vector -------------------------------------------------------------
    def device_layout(self, k: int):
        """For span k: (n_dev, index of every user parameter in the device vector, init_lo, init_hi, bound_lo, bound_hi)
        in device order.  Device vector: 6 (k + 1) U-gate angles (theta, phi, lambda per qubit and layer), then
        ``_dev_qn`` parameters per gate.  vz_only: rz(lambda) = U(0, 0, lambda) up to a global phase, so theta and phi are
        fixed at 0; a dummy gate parameter is fixed at 0."""
        n_p_dev = 6 * (k + 1)
        n_dev = n_p_dev + self._dev_qn * k
        names = self.parameter_names(k)
        idx = []
        first_layer = 1 if self.no_exterior_1q else 0  # no_exterior_1q: device layers 0 and k stay fixed at U(0, 0, 0) = 1
        for i in range(self._n_p(k)):
            # rz of (layer j, qubit b) = P{2j+b} -> lambda slot 6j + 3b + 2
            idx.append(6 * first_layer + (3 * i + 2 if self.vz_only else i))
        for j in range(k):
            for m in range(self.n_gate_params):
                idx.append(n_p_dev + self._dev_qn * j + m)
        idx = np.asarray(idx, dtype=np.int64)
        init_lo = np.zeros(n_dev)
        init_hi = np.zeros(n_dev)
        blo = np.zeros(n_dev)  # slots no user parameter maps to stay fixed at 0
        bhi = np.zeros(n_dev)
        for name, d in zip(names, idx):
            lo, hi = self.bounds.get(name, DEFAULT_BOUND) or DEFAULT_BOUND
            lo_i = DEFAULT_BOUND[0] if lo is None else float(lo)
            hi_i = DEFAULT_BOUND[1] if hi is None else float(hi)
            init_lo[d], init_hi[d] = lo_i, hi_i
            if self.using_bounds:
                # basisv2.py:160-169: once any bound is set, bounds_list holds every parameter's bound -- the explicit one
                # or the default (-4 pi, 4 pi); a None side is open
                blo[d] = -np.inf if lo is None else float(lo)
                bhi[d] = np.inf if hi is None else float(hi)
            else:
                blo[d], bhi[d] = -np.inf, np.inf  # bounds_list = None: plain BFGS (basisv2.py:170-171)
        return n_dev, idx, init_lo, init_hi, blo, bhi

    def to_device_vector(self, Xk, k=None) -> np.ndarray:
        k = self.cycles if k is None else k
        n_dev, idx, *_ = self.device_layout(k)
        X = np.atleast_2d(np.asarray(Xk, dtype=np.float64))
        if X.shape[1] != len(idx):
            raise ValueError(f"expected {len(idx)} parameters for {k} cycles, got {X.shape[1]}")
        out = np.zeros((X.shape[0], n_dev))
        out[:, idx] = X
        return out

    def from_device_vector(self, Xdev, k=None) -> np.ndarray:
        k = self.cycles if k is None else k
        _, idx, *_ = self.device_layout(k)
        return np.asarray(Xdev, dtype=np.float64)[..., idx]

    # ---- numerics -----------------------------------------------------------------------------------
    def _ctx(self):
        ctx = runtime.get_context(self.device)
        ctx.v2_set_gates(self._gate_maps)
        if ctx.n_targets == 0:
            ctx.set_targets(np.eye(4, dtype=np.complex128)[None])
        return ctx

    def eval(self, Xk):
        """basisv2.py:145-147: the template unitary, evaluated by the HIP library."""
        if self.cycles <= 0:
            raise ValueError("build() the template first")
        _, _, W = self._ctx().v2_eval(self.gate_sequence(), self.to_device_vector(Xk), want_grad=False, want_unitary=True)
        return W[0]

    def gates_of(self, Xk) -> list:
        """The 2Q gate objects of the bound circuit, in circuit order."""
        k = self.cycles
        q = np.asarray(Xk, dtype=np.float64)[self._n_p(k):]
        return [self.base_gates[g](*q[self.n_gate_params * j : self.n_gate_params * (j + 1)]) for j, g in enumerate(self.gate_sequence(k))]

    def circuit_cost(self, Xk):
        """basisv2.py:94-128: sum of the 2Q gates' costs."""
        cost = 0.0
        for g in self.gates_of(Xk):
            if hasattr(g, "cost"):
                cost += float(g.cost())
        return cost

    def circuit_fidelity(self, Xk):
        """basisv2.py:130-143: product of the RiSwap gates' ``cost()``."""
        fidelity = 1.0
        for g in self.gates_of(Xk):
            if type(g).__name__ == "RiSwapGate":
                fidelity *= float(g.cost())
        return fidelity

    def reconstruct(self, ret):
        self.build(ret.cycles)
        print("Cost:", self.circuit_cost(Xk=ret.Xk))
        return self.gates_of(ret.Xk)
