"""Target distributions (reference: src/slam/sampler.py:20-71)."""
from __future__ import annotations

import random
from abc import ABC
from sys import maxsize

import numpy as np
from scipy.stats import unitary_group

from .gates import gate_matrix


def random_unitary(dims: int, seed=None) -> np.ndarray:
    """qiskit ``random_unitary(dims, seed).data`` recipe: SciPy's Haar ``unitary_group`` driven by
    ``np.random.default_rng(seed)``."""
    return unitary_group.rvs(dims, random_state=np.random.default_rng(seed))


class SampleFunction(ABC):
    def __init__(self, n_qubits=2, n_samples=1):
        self.n_qubits = n_qubits
        self.n_samples = n_samples

    def __iter__(self):
        for _ in range(self.n_samples):
            yield self._get_unitary()

    def _get_unitary(self):
        raise NotImplementedError


class GateSample(SampleFunction):
    """src/slam/sampler.py:33-39."""

    def __init__(self, gate, n_samples=1):
        self.gate = gate
        super().__init__(getattr(gate, "num_qubits", 2), n_samples)

    def _get_unitary(self):
        return gate_matrix(self.gate)


class HaarSample(SampleFunction):
    """src/slam/sampler.py:62-71.  Faithful to the reference: Python's ``random`` is re-seeded
    with ``self.seed`` on *every* draw, so an integer seed yields ``n_samples`` identical
    unitaries and ``seed=None`` yields fresh OS entropy each time.
    Use :class:`HaarBatch` for distinct reproducible targets."""

    def __init__(self, seed=None, n_samples=1, n_qubits=2):
        self.seed = seed
        super().__init__(n_samples=n_samples, n_qubits=n_qubits)

    def _get_unitary(self):
        random.seed(self.seed)
        return random_unitary(dims=2**self.n_qubits, seed=random.randint(0, maxsize))


class HaarBatch(SampleFunction):
    """``n_samples`` distinct Haar targets, T_i = random_unitary(4, seed0 + start + i): the
    synthetic benchmark set of SURVEY.md §8(d) (not in the reference)."""

    def __init__(self, seed0: int = 20260000, n_samples: int = 1, start: int = 0, n_qubits=2):
        self.seed0 = int(seed0)
        self.start = int(start)
        super().__init__(n_samples=n_samples, n_qubits=n_qubits)

    def __iter__(self):
        for i in range(self.n_samples):
            yield random_unitary(2**self.n_qubits, seed=self.seed0 + self.start + i)

    def as_array(self) -> np.ndarray:
        return np.stack(list(self))


class DeviceHaarBatch(SampleFunction):
    """``n_samples`` distinct Haar targets generated on the GPU (``slam_sample_haar``): same recipe as
    :class:`HaarSample` (Ginibre -> QR with positive diagonal) but driven by Philox4x32-10 keyed on
    ``(seed, start + i)``, so no host RNG and no upload.  ``TemplateOptimizer`` recognises this sampler
    and leaves the targets on the device; iterating it (like any sampler) copies them back once."""

    def __init__(self, seed: int = 0, n_samples: int = 1, start: int = 0, device: int = 0, n_qubits=2):
        if n_qubits != 2:
            raise NotImplementedError("device sampler: 2 qubits only")
        self.seed = int(seed)
        self.start = int(start)
        self.device = device
        self._cache = None
        super().__init__(n_samples=n_samples, n_qubits=n_qubits)

    def fill(self, ctx, first: int = 0, count=None) -> None:
        """Make this batch -- or its window [first, first + count), one device's shard -- the resident targets of
        ``ctx`` (generated in place)."""
        count = self.n_samples - first if count is None else count
        ctx.sample_haar(self.seed, count, self.start + first)

    def as_array(self) -> np.ndarray:
        if self._cache is None:
            from . import runtime

            ctx = runtime.get_context(self.device)
            self.fill(ctx)
            self._cache = ctx.get_targets(0, self.n_samples)
        return self._cache

    def __iter__(self):
        return iter(self.as_array())
