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
        if no_exterior_1q:
            raise NotImplementedError("no_exterior_1q=True is not implemented on the HIP path")
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
            # the reference needs monodromy's precomputed coverage sets; here: analytic rules (span_rules.py) -- exact for
            # one basis gate of a known class, otherwise a sound LOWER bound from which the span loop starts
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
                # a sequence of different gates (or a gate outside the single-gate classes) whose one-, two- and three-gate
                # coverage is known exactly (span_rules.sequence_minimal_span: e.g. [iSWAP, B])
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
        return 6 * (self.cycles + 1)

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
        """True: ``minimal_spans`` is the template size each target needs (one basis gate of a class with closed-form coverage
        regions); False: a lower bound (mixed sequences, other gates) -- the span loop runs from it to ``maximum_span_guess``."""
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
        w, _ = ctx.eval_unitary(self.gate_sequence(), Xk)
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
