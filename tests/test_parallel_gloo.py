"""CPU, world_size 2 (gloo): sharding + final merge equal the single-rank result bit for bit.

The per-rank "work" here is the oracle (there is no GPU in this test); what is under test is the
multi-GPU host logic of slam_decomposition_amd/parallel.py used by bench.py."""
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
    import numpy as np
    import torch.distributed as dist
    from oracle import slam_oracle as o
    from slam_decomposition_amd.parallel import TorchDistComm, merge_results, shard_range

    dist.init_process_group("gloo")
    comm = TorchDistComm()
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
