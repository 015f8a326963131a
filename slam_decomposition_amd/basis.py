"""``CircuitTemplate`` (reference: src/slam/basis.py:52-169) without qiskit.

The template is the alternating circuit  [U(q0) U(q1)] (G [U(q0) U(q1)])^k  on two qubits;
its unitary is W(x) = K_k G_k ... G_1 K_0 with K_j = U3(x[6j+3:6j+6]) (x) U3(x[6j:6j+3])
(qiskit little-endian).  ``eval`` and everything the optimizer does with the template run in the
HIP library; this class only carries the structure (which 2Q gate at which position).
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

from . import runtime, span_rules
from .basis_abc import VariationalTemplate
from .gates import RiSwapGate, gate_matrix


class CircuitTemplate(VariationalTemplate):
    def __init__(
        self,
        n_qubits=2,
        base_gates=None,
        edge_params=None,
        no_exterior_1q=False,
        use_polytopes=False,
        maximum_span_guess=5,
        preseed=False,
        device=0,
    ):
        # reference defaults are the mutable literals [RiSwapGate(1/2)], [[(0, 1)]] (basis.py:55-56)
        if base_gates is None:
            base_gates = [RiSwapGate(1 / 2)]
        if edge_params is None:
            edge_params = [[(0, 1)]]
        if n_qubits != 2:
            raise NotImplementedError("the HIP template optimizer handles 2-qubit templates only")
        for el in edge_params:
            for e in el:
                if tuple(e) != (0, 1):
                    raise NotImplementedError("only edge (0, 1) is implemented on the HIP path")
        self.filename = None
        self.n_qubits = n_qubits
        self.no_exterior_1q = no_exterior_1q
        self.base_gates = list(base_gates)
        self.gate_matrices = np.stack([gate_matrix(g) for g in self.base_gates])
        self.edge_params = edge_params
        self.device = device
        # compliant with the reference's optimizer (basis.py:77-80)
        self.using_bounds = False
        self.bounds_list = None
        self.using_constraints = False
        self.constraint_func = None
        # basis.py:82-86: the brute-force range exists only without polytopes
        self.spanning_range = None if use_polytopes else range(1, maximum_span_guess + 1)
        self.maximum_span_guess = maximum_span_guess
        self.coverage = None
        self._span_exact = False
        if use_polytopes:
            # the reference needs monodromy's precomputed coverage sets; here: span_rules.py (closed forms for the CX, iSWAP,
            # sqrt(iSWAP), B classes) and coverage.py (the monodromy inequalities themselves, any gate sequence) -- exact
            from .weyl import c1c2c3

            self._gate_coords_all = [c1c2c3(m) for m in self.gate_matrices]
            self._gate_coords = self._gate_coords_all[0]
            self._span_sequence = False
            if len(self.base_gates) == 1:
                try:
                    span_rules.family_of(self._gate_coords)
                    self._span_exact = True
                except NotImplementedError:
                    pass
            if not self._span_exact:
                # a sequence of different gates, or a gate outside the single-gate classes (span_rules.sequence_minimal_span)
                seq = [self._gate_coords_all[i % len(self.base_gates)] for i in range(int(maximum_span_guess))]
                if span_rules.sequence_is_exact(seq, int(maximum_span_guess)):
                    self._span_exact = self._span_sequence = True
        super().__init__(preseed=preseed, use_polytopes=use_polytopes)
        self._reset()
        self.trotter = False

    # ---- structure ---------------------------------------------------------------------------
    def _reset(self):
        self.cycles = 0

    def build(self, n_repetitions):
        """basis.py:124-134.  Deviation: the reference's
        ``cycle(base_gates)`` is never reset, so with several base gates its gate order depends on
        the call history; here every build restarts the cycle: [g0, g1, g0, ...][:k]."""
        self._reset()
        if n_repetitions <= 0:
            raise ValueError()
        self.cycles = int(n_repetitions)

    def gate_sequence(self, k=None) -> List[int]:
        """Indices into ``base_gates`` of the k two-qubit gates, in circuit order."""
        k = self.cycles if k is None else k
        return [i % len(self.base_gates) for i in range(k)]

    @property
    def n_params(self) -> int:
        # basis.py:154,165: without the exterior 1Q gates the template is G_k K_{k-1} ... K_1 G_1 -- 6 (k - 1) parameters
        return 6 * (self.cycles - 1) if self.no_exterior_1q else 6 * (self.cycles + 1)

    def param_slice(self, cycles: int) -> slice:
        """Where the template's parameters sit in a device row of 6 (k + 1) values (layer-major): all of it, or -- with
        ``no_exterior_1q`` -- the interior layers 1 .. k - 1 (the device pins layers 0 and k at zero: SLAM_FLAG_NO_EXTERIOR)."""
        return slice(6, 6 * cycles) if self.no_exterior_1q else slice(0, 6 * (cycles + 1))

    def device_vector(self, Xk, cycles=None) -> np.ndarray:
        """``Xk`` ([n_params] or [M, n_params]) in the device's 6 (k + 1) layout (zeros in the pinned exterior layers)."""
        k = self.cycles if cycles is None else int(cycles)
        Xk = np.atleast_2d(np.asarray(Xk, dtype=np.float64))
        if not self.no_exterior_1q:
            return Xk
        full = np.zeros((Xk.shape[0], 6 * (k + 1)))
        full[:, 6 : 6 * k] = Xk
        return full

    def get_spanning_range(self, target_u):
        """basis.py:95-100: the brute-force range, or -- with polytopes -- only the template size the target
        needs (``range(k, k + 1)``, polytope_wrap.py:39-94; here from the analytic rules of span_rules.py)."""
        if not self.use_polytopes:
            return self.spanning_range
        from .weyl import c1c2c3

        k = int(self.minimal_spans(np.array([c1c2c3(target_u)]))[0])
        if self._span_exact:
            return range(k, k + 1)
        return range(k, self.maximum_span_guess + 1)  # lower bound: brute force from there

    @property
    def span_rules_exact(self) -> bool:
        """True: ``minimal_spans`` is the template size each target needs; False (templates longer than
        ``span_rules.MAX_EXACT_SPAN`` only): a lower bound -- the span loop runs from it to ``maximum_span_guess``."""
        return self._span_exact

    def minimal_spans(self, target_coords) -> np.ndarray:
        """Batched form of the polytope lookup: template size per target from its Weyl coordinates (exact or a lower bound,
        see ``span_rules_exact``).  A target that the whole template cannot reach raises, as the reference's lookup does
        (polytope_wrap.py:91-93)."""
        if not self.use_polytopes:
            raise ValueError("minimal_spans needs use_polytopes=True")
        kmax = int(self.maximum_span_guess)
        seq = [self._gate_coords_all[i] for i in self.gate_sequence(kmax)]
        if self._span_exact and self._span_sequence:
            k = span_rules.sequence_minimal_span(target_coords, seq, kmax)
            if np.any(k > kmax):
                raise ValueError("Monodromy did not find a polytope containing U")  # polytope_wrap.py:91-93
            return k
        if self._span_exact:
            return span_rules.minimal_span(target_coords, self._gate_coords)
        lb = span_rules.span_lower_bound(target_coords, seq, kmax)
        if np.any(lb > kmax):
            raise ValueError("Monodromy did not find a polytope containing U")  # polytope_wrap.py:91-93
        return lb

    # ---- numerics -----------------------------------------------------------------------------
    def eval(self, Xk):
        """4x4 unitary of the bound template (basis.py:102-104), computed by libslamhip."""
        if self.cycles <= 0:
            raise ValueError("build() the template first")
        Xk = np.asarray(Xk, dtype=np.float64).reshape(1, -1)
        if Xk.shape[1] != self.n_params:
            raise ValueError(f"expected {self.n_params} parameters, got {Xk.shape[1]}")
        ctx = runtime.get_context(self.device)
        ctx.set_gates(self.gate_matrices)
        if ctx.n_targets == 0:
            ctx.set_targets(np.eye(4, dtype=np.complex128)[None])
        w, _ = ctx.eval_unitary(self.gate_sequence(), self.device_vector(Xk))
        return w[0]

    def parameter_guess(self, t=0):
        """basis.py:106-111: uniform in [0, 2pi) from NumPy's global generator."""
        parent = super().parameter_guess(t)
        if parent is not None:
            return parent
        return np.random.random(self.n_params) * 2 * np.pi

    # ---- qiskit-free replacements for assign_Xk / .circuit -------------------------------------
    def to_gate_list(self, Xk) -> list:
        """The bound circuit as a list of instructions, in time order:
        ("u", qubit, (theta, phi, lam)) and ("gate", gate_object, (0, 1)).  Replaces
        ``assign_Xk`` (basis.py:113-116), which returns a qiskit ``QuantumCircuit``."""
        Xk = list(np.asarray(Xk, dtype=np.float64))
        if len(Xk) != self.n_params:
            raise ValueError(f"expected {self.n_params} parameters, got {len(Xk)}")
        out = []
        seq = self.gate_sequence()
        if self.no_exterior_1q:
            for j in range(1, self.cycles + 1):
                out.append(("gate", self.base_gates[seq[j - 1]], (0, 1)))
                if j < self.cycles:
                    out.append(("u", 0, tuple(Xk[6 * (j - 1) : 6 * (j - 1) + 3])))
                    out.append(("u", 1, tuple(Xk[6 * (j - 1) + 3 : 6 * j])))
            return out
        for j in range(self.cycles + 1):
            if j > 0:
                out.append(("gate", self.base_gates[seq[j - 1]], (0, 1)))
            out.append(("u", 0, tuple(Xk[6 * j : 6 * j + 3])))
            out.append(("u", 1, tuple(Xk[6 * j + 3 : 6 * j + 6])))
        return out

    @staticmethod
    def qiskit_parameter_order(n: int) -> List[int]:
        """qiskit returns ``circuit.parameters`` sorted by name, so the reference zips ``Xk`` with
        P0, P1, P10, P11, ..., P2, ... (basis.py:113-116).  ``order[j]`` is the index (P-number) of
        the j-th parameter in that order."""
        return sorted(range(n), key=lambda i: f"P{i}")

    @classmethod
    def from_qiskit_order(cls, Xk_sorted: Sequence[float]) -> np.ndarray:
        """Reference-order vector (zipped with name-sorted parameters) -> index order P0..P{n-1}."""
        n = len(Xk_sorted)
        out = np.empty(n)
        for j, i in enumerate(cls.qiskit_parameter_order(n)):
            out[i] = Xk_sorted[j]
        return out

    @classmethod
    def to_qiskit_order(cls, Xk: Sequence[float]) -> np.ndarray:
        n = len(Xk)
        return np.array([Xk[i] for i in cls.qiskit_parameter_order(n)])


class CircuitCoverage:
    """One entry of a coverage set: the circuits made of a given multiset of basis gates with free local gates in between -- what
    a ``monodromy.coverage.CircuitPolytope`` is to the reference (``operations`` = gate keys, ``cost``; polytope_wrap.py:78-90,
    basis.py:336-359) -- with the polytope replaced by ``span_rules.multiset_coverage`` (14 half-spaces from ``coverage.region``)."""

    def __init__(self, operations, cost, gate_indices, gate_coords):
        self.operations = list(operations)
        self.cost = cost
        self.gate_indices = [int(i) for i in gate_indices]
        self.gate_coords = np.asarray(gate_coords, dtype=np.float64).reshape(-1, 3)

    def __len__(self):
        return len(self.operations)

    def __repr__(self):
        return f"CircuitCoverage(operations={self.operations}, cost={self.cost})"

    def inside(self, target_coords, slack: float = 8e-8):
        """``(mask, exact)`` for Weyl coordinates [N, 3] (units of pi): see ``span_rules.multiset_coverage``."""
        return span_rules.multiset_coverage(np.asarray(target_coords, dtype=np.float64).reshape(-1, 3), self.gate_coords, slack)

    @property
    def exact(self) -> bool:
        return bool(self.inside(np.array([[0.3, 0.2, 0.1]]))[1])

    def has_element(self, target_coords) -> bool:
        """``CircuitPolytope.has_element`` for one target."""
        return bool(self.inside(target_coords)[0][0])


class MixedOrderBasisCircuitTemplate(CircuitTemplate):
    """Templates over a SET of basis gates in which every target gets the cheapest circuit that reaches it (reference:
    src/slam/basis.py:213-359 with ``monodromy_range_from_target``, src/slam/utils/polytopes/polytope_wrap.py:39-94: the coverage set
    sorted by cost, the first entry containing the target is bound with ``set_polytope`` and ``build`` lays out ITS gates).

    The reference takes the coverage set from monodromy (``gate_set_to_coverage`` or a pickled file); neither exists here.  The
    coverage set is every multiset of the basis gates up to ``maximum_span_guess`` gates, sorted by cost (sum of ``gate.cost()``,
    polytope_wrap.py:175-176), each with its exact region (``coverage.py``: the monodromy inequalities from first principles).
    ``TemplateOptimizer`` runs the entries in cost order over the targets they contain; a target the optimiser misses in its entry
    (too few restarts) stays open for the costlier entries that contain it too.

    Kept from the reference: the constructor checks and their errors, ``gc < gg`` ordering and unit duration of the gates,
    ``gate_hash`` / ``coverage`` / ``scores`` / ``set_polytope`` / ``unit_cost`` / ``build(n, scaled_gate)``.  The reference's coverage sets
    grow until they fill the chamber (26 applications of a pi/32 gate in scripts/haar_improvements.ipynb); here they stop at
    ``maximum_span_guess`` gates (default 5 = the longest template the optimizer kernels take), beyond which the lookup raises.  Deviation:
    ``get_spanning_range`` returns ``range(k, k + 1)`` with k the number of gates of the bound entry; the reference returns the
    entry's INDEX in the sorted list (polytope_wrap.py:94), which equals k for one basis gate and trips ``build``'s
    ``assert n_repetitions == len(gate_list)`` (basis.py:358) for several."""

    mixed_order = True

    def __init__(self, base_gates, chatty_build=True, cost_1q=0, bare_cost=True, coverage_saved_memory=True,
                 use_smush_polytope=False, maximum_span_guess=5, device=0, **kwargs):
        import itertools

        from .gates import ConversionGainGate
        from .weyl import c1c2c3

        self.homogenous = len(base_gates) == 1
        if cost_1q != 0 or bare_cost is False:
            raise ValueError("just don't do this lol")  # basis.py:235-240
        if not all(isinstance(gate, ConversionGainGate) for gate in base_gates):
            raise ValueError("all base gates must be ConversionGainGate")  # basis.py:242-243
        if use_smush_polytope:
            raise ValueError("Smush Polytope not in memory, need to compute using parallel_drive_volume.py")  # basis.py:291-294
        # gc < gg so that both orderings share a coverage set, durations normalised to 1 (basis.py:245-260)
        new_base_gates = []
        for gate in base_gates:
            if not gate.params[2] < gate.params[3]:
                p = list(gate.params)
                p[2], p[3] = p[3], p[2]
                gate = ConversionGainGate(*p)
            else:
                gate = ConversionGainGate(*gate.params)
            gate.normalize_duration(1)
            new_base_gates.append(gate)
        super().__init__(n_qubits=2, base_gates=new_base_gates, edge_params=[[(0, 1)]], no_exterior_1q=False, use_polytopes=True,
                         maximum_span_guess=maximum_span_guess, preseed=False, device=device)
        self.gate_hash = {}
        for gate in self.base_gates:
            if str(gate) in self.gate_hash:
                raise ValueError("need unique gate strings for hashing to work")  # polytope_wrap.py:148-149
            self.gate_hash[str(gate)] = gate
        keys = list(self.gate_hash)
        coords = [c1c2c3(m) for m in self.gate_matrices]
        costs = [float(g.cost()) for g in self.base_gates]
        entries = []
        for k in range(1, int(maximum_span_guess) + 1):
            for combo in itertools.combinations_with_replacement(range(len(keys)), k):
                entries.append(CircuitCoverage([keys[i] for i in combo], sum(costs[i] for i in combo), combo, [coords[i] for i in combo]))
        entries.sort(key=lambda e: (round(e.cost, 12), len(e), e.gate_indices))
        self.coverage = entries
        self.scores = None
        self.circuit_polytope = None
        self.cost = None
        self._span_exact = all(e.exact for e in entries)

    # ---- the bound entry ------------------------------------------------------------------------
    def set_polytope(self, circuit_polytope):
        self.circuit_polytope = circuit_polytope
        self.cost = circuit_polytope.cost

    def unit_cost(self, n_):
        return self.cost

    def _reset(self):
        self.circuit_polytope = None
        super()._reset()

    def build(self, n_repetitions, scaled_gate=None):
        """basis.py:336-359: the template of the bound coverage entry (its gates, in its order)."""
        assert self.circuit_polytope is not None
        entry = self.circuit_polytope
        if scaled_gate is not None:
            if not self.homogenous:
                raise ValueError("Can't use this hacky substitute method for mixed basis sets")
            self.base_gates = [scaled_gate]
            self.gate_matrices = np.stack([gate_matrix(scaled_gate)])
            seq = [0] * int(n_repetitions)
        else:
            seq = list(entry.gate_indices)
        assert n_repetitions == len(seq)
        if n_repetitions <= 0:
            raise ValueError()
        self.cycles = int(n_repetitions)
        self._sequence = seq
        self.circuit_polytope = entry

    def gate_sequence(self, k=None) -> List[int]:
        if self.circuit_polytope is None:
            raise ValueError("set_polytope() a coverage entry first (get_spanning_range does)")
        seq = getattr(self, "_sequence", None) or list(self.circuit_polytope.gate_indices)
        if k is not None and k != len(seq):
            raise ValueError(f"the bound coverage entry has {len(seq)} gates, not {k}")
        return list(seq)

    # ---- lookup ---------------------------------------------------------------------------------
    def candidate_entries(self, target_coords):
        """Per coverage entry, in cost order: ``(entry, mask[N], exact)`` -- the targets the entry can contain."""
        c = np.asarray(target_coords, dtype=np.float64).reshape(-1, 3)
        return [(e,) + tuple(e.inside(c)) for e in self.coverage]

    def minimal_spans(self, target_coords) -> np.ndarray:
        """Number of gates of the cheapest entry that can contain each target (exact where every entry is, see
        ``span_rules_exact``); 0 for local targets."""
        c = np.asarray(target_coords, dtype=np.float64).reshape(-1, 3)
        k = np.full(len(c), -1, dtype=np.int64)
        for e, mask, _ in self.candidate_entries(c):
            k = np.where((k < 0) & mask, len(e), k)
        k = np.where(np.max(np.abs(span_rules._fold(c)), axis=1) < span_rules._TOL, 0, k)
        if np.any(k < 0):
            raise ValueError("Monodromy did not find a polytope containing U")  # polytope_wrap.py:91-93
        return k

    def get_spanning_range(self, target_u):
        """``monodromy_range_from_target`` (polytope_wrap.py:39-94): binds the cheapest coverage entry that contains the target and
        returns the one-element range of its size."""
        from .weyl import c1c2c3

        coords = np.array([c1c2c3(target_u)])
        if np.max(np.abs(coords)) < span_rules._TOL:
            return range(0, 1)  # polytope_wrap.py:53-54
        for e, mask, _ in self.candidate_entries(coords):
            if mask[0]:
                self.set_polytope(e)
                self._sequence = None
                return range(len(e), len(e) + 1)
        raise ValueError("Monodromy did not find a polytope containing U")  # polytope_wrap.py:91-93
