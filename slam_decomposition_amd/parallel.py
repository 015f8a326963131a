"""Multi-GPU sharding of the (target x restart) batch and the final merge.

The path shards by target: every rank owns a contiguous block of targets with all
their restarts, so the per-target argmin over restarts is local and no collective sits on the
data path.  The only exchange is at the very end: a min-all-reduce of the best-loss vector
(every rank contributes +inf outside its shard), plus -- when the caller wants the winning
parameters everywhere -- a masked sum-all-reduce of best_x / best_cycles (exactly one rank holds
each target, the others contribute zeros).

``comm`` is anything with ``allreduce_min(np.ndarray)`` and ``allreduce_sum(np.ndarray)`` that
reduce in place across ranks; :class:`TorchDistComm` wraps ``torch.distributed`` (backend "nccl" is
RCCL over xGMI on MI355X, "gloo" on CPUs for tests).  torch is imported only there.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def shard_range(n_targets: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [first, first + count) of targets owned by ``rank`` (sizes differ by <= 1)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank / world size")
    base, rem = divmod(int(n_targets), world)
    first = rank * base + min(rank, rem)
    count = base + (1 if rank < rem else 0)
    return first, count


class LocalComm:
    """world_size = 1: reductions are the identity."""

    rank = 0
    world = 1

    def allreduce_min(self, a: np.ndarray) -> None:
        pass

    def allreduce_sum(self, a: np.ndarray) -> None:
        pass


class TorchDistComm:
    """``torch.distributed`` process group as the communicator (one process per GPU)."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist

        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._torch = torch
        self._dist = dist
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()
        self.device = device  # None = CPU tensors (gloo); "cuda" for nccl/RCCL

    def _reduce(self, a: np.ndarray, op) -> None:
        t = self._torch.from_numpy(np.ascontiguousarray(a))
        if self.device is not None:
            t = t.to(self.device)
        self._dist.all_reduce(t, op=op)
        a[...] = t.cpu().numpy()

    def allreduce_min(self, a: np.ndarray) -> None:
        self._reduce(a, self._dist.ReduceOp.MIN)

    def allreduce_sum(self, a: np.ndarray) -> None:
        self._reduce(a, self._dist.ReduceOp.SUM)


def merge_results(comm, n_targets: int, first: int, best_loss, best_x=None, best_cycles=None):
    """Merge per-rank shard results into whole-job arrays on every rank.

    best_loss [count], best_x [count, nmax], best_cycles [count] are this rank's shard (targets
    first .. first + count - 1).  Returns (loss[n_targets], x[n_targets, nmax] | None,
    cycles[n_targets] | None), identical on all ranks and bit-identical to a single-rank run."""
    count = len(best_loss)
    loss = np.full(n_targets, np.inf, dtype=np.float64)
    loss[first : first + count] = best_loss
    comm.allreduce_min(loss)
    x = cyc = None
    if best_x is not None:
        x = np.zeros((n_targets, best_x.shape[1]), dtype=np.float64)
        x[first : first + count] = best_x
        comm.allreduce_sum(x)
    if best_cycles is not None:
        cyc = np.zeros(n_targets, dtype=np.int64)
        cyc[first : first + count] = best_cycles
        comm.allreduce_sum(cyc)
        cyc = cyc.astype(np.int32)
    return loss, x, cyc
