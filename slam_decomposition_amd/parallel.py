"""Multi-GPU sharding of the (target x restart) batch and the final merge.

The path shards by target: every rank owns a contiguous block of targets with all
their restarts, so the per-target argmin over restarts is local and no collective sits on the
data path.  The only exchange is at the very end: a min-all-reduce of the best-loss vector
(every rank contributes +inf outside its shard) -- the running minimum of
``TemplateOptimizer._run`` (src/slam/optimizer.py:281-284) taken over the ranks -- plus, when the
caller wants the winning parameters everywhere, a masked sum-all-reduce of best_x / best_cycles
(exactly one rank holds each target, the others contribute zeros).

One process per GPU.  The collective is RCCL over xGMI reached through libslamhip's C ABI
(``slam_comm_*``, :class:`RcclComm`) -- no torch, no MPI.  ``comm`` below is anything with
``rank``, ``world`` and in-place ``allreduce_min / allreduce_sum / allreduce_max(np.ndarray)``;
tests plug a gloo communicator in (tests/gloo_comm.py), :class:`LocalComm` is world size 1.

Rendezvous: rank 0 creates the 128-byte ``ncclUniqueId`` and the other ranks of the node read it
from a file (``exchange_unique_id``); the file's name comes from ``SLAM_COMM_FILE`` or, under
``torch.distributed.run`` / ``bench.py``'s own launcher, from the launcher's pid and port.
"""
from __future__ import annotations

import os
import struct
import tempfile
import time
from typing import Callable, Optional, Tuple

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

    def allreduce_max(self, a: np.ndarray) -> None:
        pass

    def barrier(self) -> None:
        pass

    def close(self) -> None:
        pass


# ---------------------------------------------------------------------------------------------
# rendezvous of the ranks of one node through a file
# ---------------------------------------------------------------------------------------------
def rendezvous_path(env=os.environ) -> str:
    """File through which rank 0 hands the ncclUniqueId to the other ranks of this job.

    ``SLAM_COMM_FILE`` if set (bench.py's own launcher sets it); otherwise derived from what all ranks
    of one ``torch.distributed.run`` job share and no other job does: the launcher's pid (the workers'
    parent), MASTER_PORT and the run id."""
    path = env.get("SLAM_COMM_FILE")
    if path:
        return path
    key = f"{os.getppid()}_{env.get('MASTER_PORT', '0')}_{env.get('TORCHELASTIC_RUN_ID', 'none')}"
    key = "".join(ch if ch.isalnum() or ch in "_-" else "_" for ch in key)
    return os.path.join(tempfile.gettempdir(), f"slam_comm_{os.getuid()}_{key}.id")


_ID_MAGIC = b"SLAMID01"
_JOB_START = time.time()  # wall clock at import: no id file of THIS job can be older


def _generation_path(path: str, generation: int) -> str:
    return path if generation == 0 else f"{path}.g{generation}"


def exchange_unique_id(rank: int, world: int, path: str, make_id: Callable[[], bytes], timeout: float = 300.0,
                       nbytes: int = 128, generation: int = 0, not_before: Optional[float] = None) -> bytes:
    """Rank 0 writes ``make_id()`` to the generation's file atomically (temp file + rename); the others wait for it.

    Generation-safe: every communicator a process builds on one ``path`` has its own sequence number (``generation``,
    counted per path by :class:`RcclComm` -- all ranks of a job build their communicators in the same order), which is in
    the file's name and in its header (magic, generation, world size, rank 0's write time).  Rank 0 removes a
    pre-existing file of that name before it creates the id; readers reject a file with the wrong header or one written
    before ``not_before`` (default: 10 min before this process imported the module -- a leftover of a crashed job with a
    repeated ``SLAM_COMM_FILE`` / torchrun key is older than any rank of this job)."""
    if world == 1:
        return make_id()
    gpath = _generation_path(path, generation)
    if not_before is None:
        not_before = _JOB_START - 600.0
    if rank == 0:
        try:
            os.remove(gpath)  # a stale file of an earlier job: nobody may pick it up while the new id is being made
        except OSError:
            pass
        uid = make_id()
        if len(uid) != nbytes:
            raise ValueError(f"unique id must be {nbytes} bytes")
        tmp = f"{gpath}.tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(_ID_MAGIC + struct.pack("<qqd", int(generation), int(world), time.time()) + uid)
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, gpath)
        return uid
    t0 = time.monotonic()
    hdr = len(_ID_MAGIC) + 24
    while True:
        try:
            with open(gpath, "rb") as f:
                blob = f.read()
            if len(blob) == hdr + nbytes and blob[: len(_ID_MAGIC)] == _ID_MAGIC:
                gen, w, stamp = struct.unpack("<qqd", blob[len(_ID_MAGIC) : hdr])
                if gen == generation and w == world and stamp >= not_before:
                    return blob[hdr:]
        except FileNotFoundError:
            pass
        if time.monotonic() - t0 > timeout:
            raise TimeoutError(f"rank {rank}: no communicator id (generation {generation}) at {gpath} after {timeout:.0f} s (is rank 0 running?)")
        time.sleep(0.02)


class RcclComm:
    """RCCL communicator of this rank through libslamhip (``_ffi.Comm``): ``backend "nccl"`` without torch."""

    _generations: dict = {}  # rendezvous path -> communicators this process has built on it

    def __init__(self, device: int, rank: int, world: int, path: Optional[str] = None, timeout: float = 300.0):
        from . import _ffi

        self.rank, self.world, self.device = int(rank), int(world), int(device)
        self.path = path or rendezvous_path()
        # a second communicator on the same path gets its own file: a rank that is already through the first
        # ncclCommInitRank can never read the first communicator's id again
        gen = RcclComm._generations.get(self.path, 0)
        RcclComm._generations[self.path] = gen + 1
        uid = exchange_unique_id(self.rank, self.world, self.path, _ffi.Comm.unique_id, timeout, generation=gen)
        self.raw = _ffi.Comm(device, rank, world, uid)  # collective: returns once every rank has joined
        # what RCCL itself says about the communicator; SLAM_ERR_STATE if it is not the job we asked for
        self.rccl_rank, self.rccl_world = self.raw.rccl_rank_world()
        if self.rank == 0 and self.world > 1:
            try:
                os.remove(_generation_path(self.path, gen))  # every rank holds the id by now
            except OSError:
                pass

    @classmethod
    def from_env(cls, device: Optional[int] = None, env=os.environ) -> "RcclComm":
        """RANK / WORLD_SIZE / LOCAL_RANK as set by ``torch.distributed.run`` or bench.py's launcher."""
        rank = int(env.get("RANK", "0"))
        world = int(env.get("WORLD_SIZE", "1"))
        dev = int(env.get("LOCAL_RANK", "0")) if device is None else device
        return cls(dev, rank, world, rendezvous_path(env))

    @staticmethod
    def _flat(a: np.ndarray) -> np.ndarray:
        # in place means in place: reshape(-1) of a non-contiguous array is a COPY, and reducing that would leave the
        # caller's array untouched without a word
        if not isinstance(a, np.ndarray) or a.dtype != np.float64 or not a.flags.c_contiguous:
            raise ValueError("all-reduce buffers must be C-contiguous float64 arrays (reduced in place)")
        return a.reshape(-1)

    def allreduce_min(self, a: np.ndarray) -> None:
        self.raw.allreduce_min(self._flat(a))

    def allreduce_sum(self, a: np.ndarray) -> None:
        self.raw.allreduce_sum(self._flat(a))

    def allreduce_max(self, a: np.ndarray) -> None:
        self.raw.allreduce_max(self._flat(a))

    def barrier(self) -> None:
        self.raw.barrier()

    def close(self) -> None:
        self.raw.close()


class FileComm:
    """Rehearsal communicator: the ranks of one node reduce through files in a shared directory.

    For the cases RCCL cannot serve -- several ranks sharing ONE GPU (a one-GPU box rehearsing the N > 1 path of
    bench.py: RCCL refuses duplicate devices) or no GPU at all (CPU tests).  Same interface as :class:`RcclComm`."""

    def __init__(self, rank: int, world: int, directory: str, timeout: float = 300.0):
        self.rank, self.world, self.dir, self.timeout = int(rank), int(world), directory, float(timeout)
        self._seq = 0
        os.makedirs(directory, exist_ok=True)

    def _name(self, seq: int, rank: int) -> str:
        return os.path.join(self.dir, f"ar_{seq}_{rank}.npy")

    def _allreduce(self, a: np.ndarray, op) -> None:
        seq, self._seq = self._seq, self._seq + 1
        tmp = self._name(seq, self.rank) + f".tmp{os.getpid()}.npy"
        np.save(tmp, np.ascontiguousarray(a))
        os.replace(tmp, self._name(seq, self.rank))
        acc = None
        for r in range(self.world):
            t0 = time.monotonic()
            while True:
                try:
                    part = np.load(self._name(seq, r))
                    break
                except (FileNotFoundError, ValueError, EOFError):
                    if time.monotonic() - t0 > self.timeout:
                        raise TimeoutError(f"rank {self.rank}: rank {r} never arrived at all-reduce {seq}")
                    time.sleep(0.005)
            acc = part if acc is None else op(acc, part)
        a[...] = acc.reshape(a.shape)
        if seq >= 2:  # every rank has passed all-reduce seq - 1, hence finished reading seq - 2
            try:
                os.remove(self._name(seq - 2, self.rank))
            except OSError:
                pass

    def allreduce_min(self, a: np.ndarray) -> None:
        self._allreduce(a, np.minimum)

    def allreduce_sum(self, a: np.ndarray) -> None:
        self._allreduce(a, np.add)

    def allreduce_max(self, a: np.ndarray) -> None:
        self._allreduce(a, np.maximum)

    def barrier(self) -> None:
        self._allreduce(np.zeros(1), np.add)

    def close(self) -> None:
        pass


def merge_results(comm, n_targets: int, first: int, best_loss, best_x=None, best_cycles=None):
    """Merge per-rank shard results into whole-job arrays on every rank.

    best_loss [count], best_x [count, nmax], best_cycles [count] are this rank's shard (targets
    first .. first + count - 1).  Returns (loss[n_targets], x[n_targets, nmax] | None,
    cycles[n_targets] | None), identical on all ranks.  The merge itself adds nothing: every entry comes
    from exactly one rank.  Whether a sharded job equals the single-rank job bit for bit is therefore a
    property of the per-rank results: it does with ``SLAM_FLAG_ORDERED`` (``TemplateOptimizer``'s default,
    ``deterministic=True``) and seeds keyed on the global target index (``OptParams.target_base``); with the
    plain early-exit flag the winning restart of a target depends on timing."""
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
        c = np.zeros(n_targets, dtype=np.float64)  # one dtype on the wire: cycles are small integers, exact in f64
        c[first : first + count] = best_cycles
        comm.allreduce_sum(c)
        cyc = c.astype(np.int32)
    return loss, x, cyc
