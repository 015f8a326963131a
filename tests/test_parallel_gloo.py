"""CPU, world_size 2 (gloo): sharding + final merge equal the single-rank result bit for bit.

The per-rank "work" here is the oracle (there is no GPU in this test); what is under test is the
multi-GPU host logic of slam_decomposition_amd/parallel.py used by bench.py; the communicator is pluggable
(tests/gloo_comm.py here, parallel.RcclComm -- RCCL through the C ABI -- on GPUs)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent(
    """
    import os, sys
    sys.path.insert(0, os.environ["SLAM_ROOT"])
    sys.path.insert(0, os.path.join(os.environ["SLAM_ROOT"], "tests"))
    import numpy as np
    import torch.distributed as dist
    from oracle import slam_oracle as o
    from gloo_comm import GlooComm
    from slam_decomposition_amd.parallel import merge_results, shard_range

    dist.init_process_group("gloo")
    comm = GlooComm()
    N = 7
    first, count = shard_range(N, comm.rank, comm.world)
    g = o.berkeley_matrix()
    loss = np.empty(count); xs = np.zeros((count, 18)); cyc = np.empty(count, dtype=np.int32)
    for j in range(count):
        t = first + j
        bl, bx, bk, _ = o.run_reference(o.haar_unitary(900 + t), [g], range(1, 3), 2, 1e-8,
                                         x0_fn=lambda k, r, t=t: o.x0_philox(4, t, r, k), analytic_jac=True)
        loss[j] = bl; xs[j, : len(bx)] = bx; cyc[j] = bk
    L, X, C = merge_results(comm, N, first, loss, xs, cyc)
    np.savez(os.environ["SLAM_OUT"] + f".rank{comm.rank}.npz", L=L, X=X, C=C)
    dist.destroy_process_group()
    """
)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_merge_equals_single_rank(tmp_path):
    from oracle import slam_oracle as o

    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = str(tmp_path / "res")
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, SLAM_ROOT=ROOT, SLAM_OUT=out, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    r0 = np.load(out + ".rank0.npz")
    r1 = np.load(out + ".rank1.npz")
    for key in ("L", "X", "C"):
        assert np.array_equal(r0[key], r1[key])
    # single-rank reference
    g = o.berkeley_matrix()
    for t in range(7):
        bl, bx, bk, _ = o.run_reference(o.haar_unitary(900 + t), [g], range(1, 3), 2, 1e-8,
                                         x0_fn=lambda k, r, t=t: o.x0_philox(4, t, r, k), analytic_jac=True)
        assert r0["L"][t] == bl and r0["C"][t] == bk
        assert np.array_equal(r0["X"][t, : len(bx)], bx)
    assert np.all(r0["L"] < 1e-8) and np.all(r0["C"] == 2)


def test_file_rendezvous_hands_the_id_to_every_rank(tmp_path):
    """exchange_unique_id: rank 0 publishes atomically, the others (started earlier or later) read 128 bytes."""
    import threading

    from slam_decomposition_amd.parallel import exchange_unique_id, rendezvous_path

    path = str(tmp_path / "id")
    uid = bytes(range(128))
    got = {}

    def reader(r):
        got[r] = exchange_unique_id(r, 3, path, lambda: b"", timeout=20)

    ts = [threading.Thread(target=reader, args=(r,)) for r in (1, 2)]
    ts[0].start()
    import time

    time.sleep(0.1)
    assert exchange_unique_id(0, 3, path, lambda: uid) == uid
    ts[1].start()
    for t in ts:
        t.join()
    assert got == {1: uid, 2: uid}
    # world 1 needs no file; ranks of one launcher derive the same path, another launcher's port a different one
    assert exchange_unique_id(0, 1, str(tmp_path / "none"), lambda: uid) == uid
    a = rendezvous_path({"MASTER_PORT": "29500"})
    assert a == rendezvous_path({"MASTER_PORT": "29500"}) and a != rendezvous_path({"MASTER_PORT": "29501"})
    assert rendezvous_path({"SLAM_COMM_FILE": "/x/y"}) == "/x/y"
    import pytest

    with pytest.raises(TimeoutError):
        exchange_unique_id(1, 2, str(tmp_path / "never"), lambda: uid, timeout=0.2)


def test_file_rendezvous_rejects_stale_and_foreign_ids(tmp_path):
    """a 128-byte file left at the rendezvous path by a crashed job (fixed SLAM_COMM_FILE, repeated torchrun
    key) or by an EARLIER communicator of the same job must never be taken for this communicator's id -- the ranks
    would sit in ncclCommInitRank with different ids for ever."""
    import os
    import struct
    import threading
    import time

    import pytest

    from slam_decomposition_amd import parallel
    from slam_decomposition_amd.parallel import exchange_unique_id

    path = str(tmp_path / "id")
    old, new = bytes([7]) * 128, bytes(range(128))
    # (a) a bare 128-byte leftover (round 2's format) and a well-formed file from long ago: both ignored
    open(path, "wb").write(old)
    with pytest.raises(TimeoutError):
        exchange_unique_id(1, 2, path, lambda: b"", timeout=0.2)
    open(path, "wb").write(parallel._ID_MAGIC + struct.pack("<qqd", 0, 2, time.time() - 86400.0) + old)
    with pytest.raises(TimeoutError):
        exchange_unique_id(1, 2, path, lambda: b"", timeout=0.2)
    # (b) a reader that started before rank 0 gets the NEW id: rank 0 removes the leftover before it makes its own
    got = {}
    t = threading.Thread(target=lambda: got.setdefault(1, exchange_unique_id(1, 2, path, lambda: b"", timeout=20)))
    t.start()
    time.sleep(0.1)
    assert exchange_unique_id(0, 2, path, lambda: new) == new
    t.join()
    assert got[1] == new
    # (c) the second communicator on the same path has its own file: generation 0's id is never read for generation 1,
    # neither is a file written for another world size
    with pytest.raises(TimeoutError):
        exchange_unique_id(1, 2, path, lambda: b"", timeout=0.2, generation=1)
    with pytest.raises(TimeoutError):
        exchange_unique_id(1, 3, path, lambda: b"", timeout=0.2)
    newer = bytes([9]) * 128
    assert exchange_unique_id(0, 2, path, lambda: newer, generation=1) == newer
    assert exchange_unique_id(1, 2, path, lambda: b"", timeout=5, generation=1) == newer
    assert exchange_unique_id(1, 2, path, lambda: b"", timeout=5) == new and os.path.exists(path + ".g1")


FILE_WORKER = textwrap.dedent(
    """
    import os, sys
    sys.path.insert(0, os.environ["SLAM_ROOT"])
    import numpy as np
    from slam_decomposition_amd.parallel import FileComm, merge_results, shard_range

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    comm = FileComm(rank, world, os.environ["SLAM_COMM_DIR"], timeout=60)
    N = 11
    first, count = shard_range(N, rank, world)
    rng = np.random.default_rng(5)                       # the same stream on every rank: the "single-rank result"
    loss_all, x_all, cyc_all = rng.random(N), rng.random((N, 6)), rng.integers(1, 4, N)
    comm.barrier()
    for _ in range(3):                                   # several rounds: sequence numbers, clean-up of old files
        L, X, C = merge_results(comm, N, first, loss_all[first:first + count], x_all[first:first + count], cyc_all[first:first + count])
        assert np.array_equal(L, loss_all) and np.array_equal(X, x_all) and np.array_equal(C, cyc_all)
    t = np.array([float(rank)])
    comm.allreduce_max(t)
    assert t[0] == world - 1
    print("ok", rank)
    """
)


def test_file_communicator_three_ranks(tmp_path):
    """parallel.FileComm (bench.py's rehearsal communicator for ranks that share one GPU, and its fallback when RCCL cannot
    be initialised): three processes, merge_results equals the unsharded arrays on every rank."""
    script = tmp_path / "fworker.py"
    script.write_text(FILE_WORKER)
    procs = []
    for rank in range(3):
        env = dict(os.environ, SLAM_ROOT=ROOT, RANK=str(rank), WORLD_SIZE="3", SLAM_COMM_DIR=str(tmp_path / "fc"), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
