"""Cost functions (reference: src/slam/cost_function.py:117-145).

``BasicCost`` and ``SquareCost`` (the monotone map 0.8 (2 L - L^2) of BasicCost L, cost_function.py:169-173) are the
objectives the HIP optimizer implements (``slam_set_cost``); anything else makes ``TemplateOptimizer`` raise the reference's
"Unrecognized Cost Function".  The classes keep the reference's interface; ``unitary_fidelity`` on two single matrices is the reference's own
one-line NumPy expression (used for spot checks and logging, never inside the optimizer loop --
there the loss is fused into the HIP kernel).
"""
from __future__ import annotations

from abc import ABC

import numpy as np


class UnitaryCostFunction(ABC):
    def __init__(self):
        self.normalization = 1  # src/slam/cost_function.py:123

    def unitary_fidelity(self, current_u, target_u):
        raise NotImplementedError


class BasicCost(UnitaryCostFunction):
    """1 - |Tr(T^dagger U)| / d  (src/slam/cost_function.py:140-145)."""

    def unitary_fidelity(self, current_u, target_u):
        h = np.asarray(target_u).conj().T
        cur = np.asarray(current_u)
        return 1 - np.abs(np.trace(h @ cur)) / cur.shape[0]


class SquareCost(UnitaryCostFunction):
    """1 - (|Tr(T^dagger U)|^2 + d) / (d (d + 1))  (src/slam/cost_function.py:169-173): the objective most
    of the reference's notebooks use.  On the HIP path it is the same fused kernel with a different
    scalar map of |Tr| (and of the gradient scale)."""

    def unitary_fidelity(self, current_u, target_u):
        h = np.asarray(target_u).conj().T
        d = np.asarray(target_u).shape[0]
        return 1 - (np.abs(np.trace(h @ np.asarray(current_u))) ** 2 + d) / (d * (d + 1))
