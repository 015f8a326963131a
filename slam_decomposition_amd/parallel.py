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


def step_groups(step_ids, group: int):
    """Consecutive steps handed to the library as one call: ``step_ids`` cut into runs of ``group``."""
    step_ids = list(step_ids)
    group = max(1, int(group))
    return [step_ids[i : i + group] for i in range(0, len(step_ids), group)]


def merge_slices(step_ids, first_step: int, n_per_step: int, rank: int, world: int, n_streams: int, group: int = 1):
    """Where each step's resident best-loss window goes in the JOB's merged vector (the one ``ncclAllReduce(min)`` runs over).

    A rank's timed region is ``step_ids`` (consecutive step numbers starting at ``first_step``), ``n_per_step`` targets each; the
    job vector is rank-major: rank r owns ``[r n_loc, (r + 1) n_loc)`` with ``n_loc = len(step_ids) n_per_step``, and inside it the
    steps in order.  Steps are dealt to ``n_streams`` contexts in groups of ``group`` (group g goes to stream g mod n_streams, the
    order bench.py's workers use); step s sits at targets ``[s n_per_step, (s + 1) n_per_step)`` of its context's resident array.
    Yields ``(stream, local_first, count, global_first)`` -- the arguments of ``slam_comm_merge_add``.  The slices of all ranks
    tile ``[0, world n_loc)`` exactly once (tests/test_host_logic.py)."""
    step_ids = list(step_ids)
    if not (0 <= rank < world) or n_streams < 1 or n_per_step < 1:
        raise ValueError("bad rank / world / streams / step size")
    if step_ids != list(range(first_step, first_step + len(step_ids))):
        raise ValueError("the timed region must be consecutive steps starting at first_step")
    n_loc = len(step_ids) * n_per_step
    groups = step_groups(step_ids, group)
    for w in range(n_streams):
        for g in groups[w::n_streams]:
            for s in g:
                yield w, s * n_per_step, n_per_step, rank * n_loc + (s - first_step) * n_per_step


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
    parent), MASTER_PORT, the run id and the elastic restart count (a restarted worker group is a new attempt)."""
    path = env.get("SLAM_COMM_FILE")
    if path:
        return path
    key = (f"{os.getppid()}_{env.get('MASTER_PORT', '0')}_{env.get('TORCHELASTIC_RUN_ID', 'none')}"
           f"_{env.get('TORCHELASTIC_RESTART_COUNT', '0')}")
    key = "".join(ch if ch.isalnum() or ch in "_-" else "_" for ch in key)
    return os.path.join(tempfile.gettempdir(), f"slam_comm_{os.getuid()}_{key}.id")


_ID_MAGIC = b"SLAMID02"


def _generation_path(path: str, generation: int) -> str:
    return path if generation == 0 else f"{path}.g{generation}"


def _write_atomic(path: str, blob: bytes) -> None:
    tmp = f"{path}.tmp{os.getpid()}"
    with open(tmp, "wb") as f:
        f.write(blob)
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp, path)


def _read_words(path: str, n: int) -> Optional[tuple]:
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    return struct.unpack(f"<{n}Q", blob) if len(blob) == 8 * n else None


def exchange_unique_id(rank: int, world: int, path: str, make_id: Callable[[], bytes], timeout: float = 300.0,
                       nbytes: int = 128, generation: int = 0) -> bytes:
    """Rank 0 hands ``make_id()`` to the other ranks through files, with a handshake that does not depend on any clock.

    Every communicator a process builds on one ``path`` has its own sequence number (``generation``, counted per path by
    :class:`RcclComm` -- all ranks of a job build their communicators in the same order), which is in the files' names and
    in the id file's header.  Protocol (files next to ``path``; every write is temp file + rename):

      1. every rank draws a random 64-bit NONCE; a rank r > 0 publishes its in ``<path>.ready.<r>``;
      2. rank 0 waits until it has a nonce of every other rank, creates the id and writes the id file: magic, generation,
         world size, its own nonce, the world - 1 nonces it saw, the id;
      3. rank r accepts an id file only if it carries ITS nonce, and then answers with ``<path>.ack.<r>``: its nonce and rank 0's;
      4. rank 0 waits for every rank's ack (both nonces must match).  While it waits it keeps re-reading the ready files: if one
         shows another nonce than the id file carries (it had picked up the leftover of a crashed attempt before the live rank
         overwrote it), it rewrites the id file with the nonces it sees now (same id).  With all acks in, it removes the files.

    A leftover of an earlier attempt with the same key -- id, ready or ack files of a job that died at start-up, a torchrun
    elastic restart within seconds -- can therefore never be joined or mistaken for an answer: it does not carry the nonces the
    live ranks drew for this attempt (the previous version trusted a 10-minute window of wall-clock freshness)."""
    if world == 1:
        return make_id()
    gpath = _generation_path(path, generation)
    hdr = len(_ID_MAGIC) + 24 + 8 * (world - 1)
    t0 = time.monotonic()

    def expired(what: str):
        if time.monotonic() - t0 > timeout:
            raise TimeoutError(f"rank {rank}: {what} (generation {generation}, {gpath}) after {timeout:.0f} s")

    nonce = int.from_bytes(os.urandom(8), "little") | 1
    if rank != 0:
        _write_atomic(f"{gpath}.ready.{rank}", struct.pack("<Q", nonce))
        while True:
            try:
                with open(gpath, "rb") as f:
                    blob = f.read()
            except OSError:
                blob = b""
            if len(blob) == hdr + nbytes and blob[: len(_ID_MAGIC)] == _ID_MAGIC:
                gen, w, n0 = struct.unpack("<qqQ", blob[len(_ID_MAGIC) : len(_ID_MAGIC) + 24])
                mine = struct.unpack("<Q", blob[len(_ID_MAGIC) + 24 + 8 * (rank - 1) : len(_ID_MAGIC) + 32 + 8 * (rank - 1)])[0]
                if gen == generation and w == world and mine == nonce:
                    _write_atomic(f"{gpath}.ack.{rank}", struct.pack("<QQ", nonce, n0))
                    return blob[hdr:]
            expired("no communicator id carrying this rank's nonce (is rank 0 running?)")
            time.sleep(0.01)

    # rank 0
    try:
        os.remove(gpath)  # a leftover id file carries nobody's live nonce; removed all the same
    except OSError:
        pass
    uid = None
    written: Optional[list] = None
    while True:
        seen = [(_read_words(f"{gpath}.ready.{r}", 1) or (None,))[0] for r in range(1, world)]
        if all(v is not None for v in seen):
            if uid is None:
                uid = make_id()
                if len(uid) != nbytes:
                    raise ValueError(f"unique id must be {nbytes} bytes")
            if seen != written:
                _write_atomic(gpath, _ID_MAGIC + struct.pack("<qqQ", int(generation), int(world), nonce) + struct.pack(f"<{world - 1}Q", *seen) + uid)
                written = list(seen)
            acks = [_read_words(f"{gpath}.ack.{r}", 2) for r in range(1, world)]
            if acks == [(v, nonce) for v in written]:
                break
        expired("not every rank published its nonce and acknowledged the id")
        time.sleep(0.01)
    for name in [gpath] + [f"{gpath}.{kind}.{r}" for kind in ("ready", "ack") for r in range(1, world)]:
        try:
            os.remove(name)
        except OSError:
            pass
    return uid


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
