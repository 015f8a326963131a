"""``VariationalTemplate`` / ``DataDictEntry`` (reference: src/slam/basis_abc.py)."""
from __future__ import annotations

from abc import ABC
from collections.abc import MutableSequence, Sequence
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


class LazyList(MutableSequence):
    """A list whose big numeric stretches stay NumPy arrays until somebody looks at their elements: ``training_loss`` and
    ``best_cycle_list`` of a 327 680-target batch are 0.65 M Python objects if built eagerly -- 15 ms of ``tolist()``, a fifth of what
    the GPU needs for the batch.  Behaves like the reference's plain lists (optimizer.py:38-40,307-311): ``append`` / ``extend`` /
    indexing / slicing / iteration / ``len`` / ``==`` against lists / ``+``; elements come out as Python ``float`` / ``int`` (``tolist``
    semantics); ``np.asarray(x)`` takes the arrays directly when all chunks are numeric."""

    def __init__(self, items=()):
        self._chunks = []  # each a Python list or a 1-D ndarray
        self._starts = [0]
        if len(items):
            self.extend(items)

    # -- building ---------------------------------------------------------------------------------------------------
    def _push(self, chunk):
        if len(chunk):
            self._chunks.append(chunk)
            self._starts.append(self._starts[-1] + len(chunk))

    def extend_array(self, a):
        """Append the elements of a 1-D array without converting them (the array is kept, not copied: do not modify it)."""
        self._push(np.asarray(a).reshape(-1))

    def append(self, v):
        if self._chunks and isinstance(self._chunks[-1], list):
            self._chunks[-1].append(v)
            self._starts[-1] += 1
        else:
            self._push([v])

    def extend(self, vs):
        if isinstance(vs, np.ndarray) and vs.ndim == 1:
            self.extend_array(vs)
        elif isinstance(vs, LazyList):
            for c in vs._chunks:
                self._push(c if isinstance(c, np.ndarray) else list(c))
        else:
            vs = list(vs)
            if self._chunks and isinstance(self._chunks[-1], list):
                self._chunks[-1].extend(vs)
                self._starts[-1] += len(vs)
            else:
                self._push(vs)

    def _as_list(self) -> list:
        """Collapse into ONE Python list chunk (needed for in-place edits); returns it."""
        out = []
        for c in self._chunks:
            out.extend(c.tolist() if isinstance(c, np.ndarray) else c)
        self._chunks = [out] if out else []
        self._starts = [0, len(out)] if out else [0]
        return out

    def _set_list(self, lst):
        self._chunks = [lst] if lst else []
        self._starts = [0, len(lst)] if lst else [0]

    def insert(self, i, v):
        lst = self._as_list()
        lst.insert(i, v)
        self._set_list(lst)

    def __setitem__(self, i, v):
        lst = self._as_list()
        lst[i] = v
        self._set_list(lst)

    def __delitem__(self, i):
        lst = self._as_list()
        del lst[i]
        self._set_list(lst)

    # -- reading ----------------------------------------------------------------------------------------------------
    def __len__(self):
        return self._starts[-1]

    def __iter__(self):
        for c in self._chunks:
            yield from (c.tolist() if isinstance(c, np.ndarray) else c)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        n = len(self)
        j = int(i)
        if j < 0:
            j += n
        if not 0 <= j < n:
            raise IndexError("list index out of range")
        import bisect

        b = bisect.bisect_right(self._starts, j) - 1
        v = self._chunks[b][j - self._starts[b]]
        return v.item() if isinstance(v, np.generic) else v

    def tolist(self) -> list:
        return list(self)

    def __array__(self, dtype=None, copy=None):
        if self._chunks and all(isinstance(c, np.ndarray) for c in self._chunks):
            a = self._chunks[0] if len(self._chunks) == 1 else np.concatenate(self._chunks)
        else:
            a = np.array(list(self))
        return a.astype(dtype) if dtype is not None else a

    def __eq__(self, other):
        if isinstance(other, (list, tuple, LazyList)):
            return len(other) == len(self) and all(a == b for a, b in zip(self, other))
        return NotImplemented

    def __add__(self, other):
        return list(self) + list(other)

    def __radd__(self, other):
        return list(other) + list(self)

    def __repr__(self):
        return repr(list(self)) if len(self) <= 64 else f"LazyList(n={len(self)})"


class RowBlocks:
    """The rows of several equally wide 2-D blocks seen as one [n, width] array without copying them together: row ``i`` of the
    whole is a row of the block that holds it.  What ``TargetDataList`` needs of an array -- ``len``, ``ndim``, ``shape``, ``[i]`` --
    plus ``as_array()`` for readers that want the real thing."""

    ndim = 2

    def __init__(self, blocks):
        self._blocks = [np.asarray(b) for b in blocks]
        self._starts = np.cumsum([0] + [len(b) for b in self._blocks])
        widths = {b.shape[1] for b in self._blocks}
        if len(widths) > 1:
            raise ValueError("RowBlocks: blocks of different widths")

    def __len__(self):
        return int(self._starts[-1])

    @property
    def shape(self):
        return (len(self), self._blocks[0].shape[1] if self._blocks else 0)

    def __getitem__(self, i):
        if isinstance(i, tuple):
            return self[i[0]][i[1:] if len(i) > 2 else i[1]]
        j = int(i)
        if j < 0:
            j += len(self)
        if not 0 <= j < len(self):
            raise IndexError("row index out of range")
        b = int(np.searchsorted(self._starts, j, side="right")) - 1
        return self._blocks[b][j - int(self._starts[b])]

    def as_array(self) -> np.ndarray:
        return np.concatenate(self._blocks) if self._blocks else np.zeros((0, 0))


class TargetDataList(Sequence):
    """``target_data`` of a big batch: the list of ``DataDictEntry`` that ``approximate_from_distribution`` returns
    (optimizer.py:180-186), materialised on access.  Holds the batch's result arrays -- labels, losses, the padded
    parameter block, cycles -- and builds an entry (whose ``Xk`` is the first ``6 (cycles + 1)`` parameters of the target's
    row) when it is indexed or iterated over; building 65 536 dataclass objects up front cost more than the span loop on
    the GPU.  List semantics for readers: ``len``, integer / slice indexing, iteration, ``==`` against any sequence of
    entries, ``list(...)`` for a real list.  Entries are cached, so ``data[i] is data[i]``."""

    def __init__(self, labels, losses, x_rows, cycles, width_of):
        self._labels = labels
        self._losses = losses
        self._x = x_rows          # ndarray [n, nmax] (padded rows) or a list of per-target vectors
        self._cycles = cycles
        self._width_of = width_of  # cycles -> number of parameters of the entry's Xk (None: rows are final already)
        self._cache = {}

    def __len__(self):
        return len(self._losses)

    def _entry(self, i: int) -> "DataDictEntry":
        e = self._cache.get(i)
        if e is None:
            c = int(self._cycles[i])
            row = self._x[i]
            if self._width_of is not None:
                w = self._width_of(c)
                row = row[w] if isinstance(w, slice) else row[:w]
            e = DataDictEntry(int(self._labels[i]), float(self._losses[i]), row, c)
            self._cache[i] = e
        return e

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._entry(j) for j in range(*i.indices(len(self)))]
        n = len(self)
        j = int(i)
        if j < 0:
            j += n
        if not 0 <= j < n:
            raise IndexError("target_data index out of range")
        return self._entry(j)

    def __eq__(self, other):
        try:
            if len(other) != len(self):
                return False
        except TypeError:
            return NotImplemented
        return all(_entries_equal(a, b) for a, b in zip(self, other))

    def __repr__(self):
        return f"TargetDataList(n={len(self)})"


def _entries_equal(a, b) -> bool:
    if not isinstance(a, DataDictEntry) or not isinstance(b, DataDictEntry):
        return a == b
    return (a.success_label == b.success_label and a.loss_result == b.loss_result and a.cycles == b.cycles
            and np.array_equal(np.asarray(a.Xk), np.asarray(b.Xk)))
