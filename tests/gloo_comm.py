"""Test infrastructure: a ``torch.distributed`` (gloo, CPU) communicator with the interface
``slam_decomposition_amd.parallel.merge_results`` expects, so that the N > 1 host logic can be exercised
without GPUs.  The product's communicator is ``parallel.RcclComm`` (RCCL through libslamhip's C ABI)."""
import numpy as np


class GlooComm:
    def __init__(self):
        import torch
        import torch.distributed as dist

        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._torch = torch
        self._dist = dist
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()

    def _reduce(self, a: np.ndarray, op) -> None:
        t = self._torch.from_numpy(np.ascontiguousarray(a))
        self._dist.all_reduce(t, op=op)
        a[...] = t.numpy()

    def allreduce_min(self, a: np.ndarray) -> None:
        self._reduce(a, self._dist.ReduceOp.MIN)

    def allreduce_sum(self, a: np.ndarray) -> None:
        self._reduce(a, self._dist.ReduceOp.SUM)

    def allreduce_max(self, a: np.ndarray) -> None:
        self._reduce(a, self._dist.ReduceOp.MAX)

    def barrier(self) -> None:
        self._dist.barrier()
