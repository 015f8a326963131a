"""``VariationalTemplate`` / ``DataDictEntry`` (reference: src/slam/basis_abc.py)."""
from __future__ import annotations

from abc import ABC
from dataclasses import dataclass

import numpy as np

from .weyl import c1c2c3


class VariationalTemplate(ABC):
    """The parts of src/slam/basis_abc.py:18-90 the hot path uses.  The pickle "preseed" cache and
    its KD-tree (basis_abc.py:27-29,60-77) are only active with ``use_polytopes`` (needs the
    un-vendored ``monodromy`` package) and are out of scope."""

    def __init__(self, preseed: bool, use_polytopes: bool):
        self.data_dict = {}
        self.coordinate_tree = None
        self.use_polytopes = use_polytopes
        if not self.use_polytopes and self.spanning_range is None:
            raise NotImplementedError
        self.preseeded = preseed and self.use_polytopes  # basis_abc.py:41-43
        if self.preseeded:
            # the reference would now load its pickle cache, seed restarts from the KD-tree neighbour and save every
            # result (basis_abc.py:27-29,60-77; optimizer.py:107-118,126-149).  None of that exists here: refuse
            # rather than silently run unseeded.  (preseed=True without polytopes is a no-op in the reference too.)
            raise NotImplementedError("preseed=True with use_polytopes=True (pickle cache + KD-tree neighbour seeding) "
                                      "is not implemented on the HIP path")
        self.seed = None

    def eval(self, Xk):
        raise NotImplementedError

    def parameter_guess(self, temperature=0):
        return None  # basis_abc.py:50-58 without a preseed

    def assign_seed(self, Xk):
        self.seed = Xk

    def target_invariant(self, target_U):
        target_U = np.asarray(target_U)
        if not (4, 4) == target_U.shape:  # basis_abc.py:80-84
            return (-1, -1, -1, -1)
        return c1c2c3(target_U)


@dataclass
class DataDictEntry:
    """src/slam/basis_abc.py:93-98."""

    success_label: int
    loss_result: float
    Xk: list
    cycles: int
