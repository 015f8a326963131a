"""Two-qubit basis gates: the 4x4 matrices the template optimizer consumes.

The reference wraps these in qiskit ``Gate`` subclasses
(src/slam/utils/gates/custom_gates.py); qiskit is not a dependency here, so a
gate is a small object exposing ``__array__`` / ``to_matrix()`` / ``params`` /
``name`` / ``num_qubits`` -- the members the hot path touches.  Any object with
``__array__`` or ``to_matrix()`` returning a 4x4 unitary (including a real
qiskit ``Gate``) is accepted by ``CircuitTemplate``.
"""
from __future__ import annotations

import math
from typing import Sequence

import numpy as np

_SX = np.array([[0, 1], [1, 0]], dtype=np.complex128)
_SY = np.array([[0, -1j], [1j, 0]], dtype=np.complex128)
_SZ = np.array([[1, 0], [0, -1]], dtype=np.complex128)


class Gate2Q:
    """Minimal stand-in for ``qiskit.circuit.Gate`` (2 qubits)."""

    num_qubits = 2

    def __init__(self, name: str, params: Sequence[float], label: str | None = None):
        self.name = name
        self.params = list(params)
        self.label = label or name

    def __array__(self, dtype=None, copy=None):
        m = self._matrix()
        return m if dtype is None else m.astype(dtype)

    def to_matrix(self) -> np.ndarray:
        return np.asarray(self.__array__(), dtype=np.complex128)

    def _matrix(self) -> np.ndarray:  # pragma: no cover - abstract
        raise NotImplementedError

    def __repr__(self):
        return f"{type(self).__name__}({', '.join(repr(p) for p in self.params)})"

    def __str__(self):
        return repr(self)


class UnitaryGate(Gate2Q):
    """A fixed 4x4 unitary given as an array."""

    def __init__(self, matrix, name: str = "unitary"):
        m = np.asarray(matrix, dtype=np.complex128)
        if m.shape != (4, 4):
            raise ValueError("UnitaryGate needs a 4x4 matrix")
        super().__init__(name, [])
        self._m = m

    def _matrix(self):
        return self._m


class CXGate(Gate2Q):
    """qiskit ``CXGate`` (control qubit 0, target qubit 1; little-endian matrix), star-imported by
    the reference at src/slam/basis.py:10."""

    def __init__(self):
        super().__init__("cx", [])

    def _matrix(self):
        return np.array([[1, 0, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0], [0, 1, 0, 0]], dtype=np.complex128)


class CZGate(Gate2Q):
    def __init__(self):
        super().__init__("cz", [])

    def _matrix(self):
        return np.diag([1, 1, 1, -1]).astype(np.complex128)


class SwapGate(Gate2Q):
    def __init__(self):
        super().__init__("swap", [])

    def _matrix(self):
        return np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.complex128)


class RiSwapGate(Gate2Q):
    """``RiSwapGate(alpha)`` = iSWAP**alpha (src/slam/utils/gates/custom_gates.py:534-606)."""

    def __init__(self, alpha: float):
        super().__init__("riswap", [alpha], label="riswap")
        self.duration = self.cost()

    def cost(self) -> float:
        return float(self.params[0])  # custom_gates.py:564-568

    def _matrix(self):
        a = float(self.params[0]) / 2.0  # custom_gates.py:582-595
        c = math.cos(math.pi * a)
        isin = 1j * math.sin(math.pi * a)
        return np.array([[1, 0, 0, 0], [0, c, isin, 0], [0, isin, c, 0], [0, 0, 0, 1]], dtype=np.complex128)


class iSwapGate(RiSwapGate):
    def __init__(self):
        super().__init__(1.0)
        self.name = "iswap"


def canonical_matrix(c1: float, c2: float, c3: float) -> np.ndarray:
    """``weylchamber.canonical_gate(c1, c2, c3)`` = exp(i pi/2 (c1 XX + c2 YY + c3 ZZ)); the three
    terms commute, so the exponential factorises."""
    out = np.eye(4, dtype=np.complex128)
    for c, s in ((c1, _SX), (c2, _SY), (c3, _SZ)):
        ang = math.pi / 2.0 * c
        out = out @ (math.cos(ang) * np.eye(4) + 1j * math.sin(ang) * np.kron(s, s))
    return out


class CanonicalGate(Gate2Q):
    """``CanonicalGate(alpha, beta, gamma)`` (custom_gates.py:384-392): arguments in radians,
    rescaled by 2/pi before ``weylchamber.canonical_gate``."""

    def __init__(self, alpha, beta, gamma, name="can"):
        super().__init__(name, [alpha, beta, gamma])
        a, b, g = (2 * x / np.pi for x in (alpha, beta, gamma))
        self.data = canonical_matrix(a, b, g)

    def _matrix(self):
        return self.data


class BerkeleyGate(CanonicalGate):
    """custom_gates.py:395-400."""

    def __init__(self):
        super().__init__(np.pi / 4, np.pi / 8, 0, name="B")

    def __str__(self):
        return "B"


def conversion_gain_matrix(phi_c: float, phi_g: float, gc: float, gg: float, t: float = 1.0) -> np.ndarray:
    """Closed form of exp(-i t H), H = gc (e^{i phi_c} A B^+ + h.c.) + gg (e^{i phi_g} A B + h.c.)
    (src/slam/hamiltonian.py:84-111): two decoupled 2x2 rotations on {|01>,|10>} and {|00>,|11>}."""
    U = np.zeros((4, 4), dtype=np.complex128)
    cc, sc = math.cos(gc * t), math.sin(gc * t)
    cg, sg = math.cos(gg * t), math.sin(gg * t)
    U[1, 1] = U[2, 2] = cc
    U[2, 1] = -1j * np.exp(1j * phi_c) * sc
    U[1, 2] = -1j * np.exp(-1j * phi_c) * sc
    U[0, 0] = U[3, 3] = cg
    U[3, 0] = -1j * np.exp(1j * phi_g) * sg
    U[0, 3] = -1j * np.exp(-1j * phi_g) * sg
    return U


class ConversionGainGate(Gate2Q):
    """``ConversionGainGate(p1, p2, g1, g2, t_el)`` (custom_gates.py:163-212); effective argument
    meaning phi_c = p1, phi_g = p2, gc = g1, gg = g2."""

    def __init__(self, p1, p2, g1, g2, t_el=1):
        super().__init__("2QGate", [p1, p2, g1, g2, t_el])
        self.duration = self.cost()
        self.name = str(self)

    def cost(self):
        norm = np.pi / 2  # custom_gates.py:207-211
        return (sum(abs(np.array(self.params[2:4]))) * self.params[-1]) / norm

    def __str__(self):
        g1, g2, t = self.params[2], self.params[3], self.params[4]
        return f"2QGate({g1:.8f}, {g2:.8f}, {t:.8f})"

    def normalize_duration(self, new_duration):
        """custom_gates.py:192-205: rescale the drive strengths so that the gate lasts ``new_duration`` (same unitary, same cost)."""
        t = self.params[-1]
        self.params[2] = self.params[2] * t / new_duration
        self.params[3] = self.params[3] * t / new_duration
        self.params[-1] = new_duration
        self.duration = self.cost()
        self.name = str(self)

    def _matrix(self):
        p1, p2, g1, g2, t = (float(v) for v in self.params)
        return conversion_gain_matrix(p1, p2, g1, g2, t)


def gate_matrix(gate) -> np.ndarray:
    """4x4 complex128 matrix of a gate object / array."""
    if hasattr(gate, "to_matrix"):
        m = gate.to_matrix()
    else:
        m = np.asarray(gate)
    m = np.asarray(m, dtype=np.complex128)
    if m.shape != (4, 4):
        raise ValueError(f"2Q basis gate must be a 4x4 matrix, got shape {m.shape}")
    return m
