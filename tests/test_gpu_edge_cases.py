"""GPU: edge cases of the C ABI and of the batching logic (ragged sizes, windows, error codes)."""
import numpy as np
import pytest

from oracle import slam_oracle as o
from slam_decomposition_amd import _ffi

pytestmark = pytest.mark.gpu

SQ = o.riswap_matrix(0.5)
SEQS = [[0], [0, 0], [0, 0, 0]]


def _prm(**kw):
    d = dict(restarts=5, seed=42, flags=_ffi.FLAG_EARLY_EXIT)
    d.update(kw)
    return _ffi.OptParams(**d)


@pytest.mark.parametrize("N,R", [(1, 1), (1, 5), (3, 7), (17, 5), (70, 3), (33, 32)])
def test_ragged_batches(hip_ctx, N, R):
    """Reference default TRAINING_RESTARTS = 5 (optimizer.py:19) and other sizes that do not line up
    with 16 quads per wave or with the work-queue chunk."""
    targets = o.haar_batch(N, seed0=9000 + N)
    hip_ctx.set_targets(targets)
    hip_ctx.set_gates(SQ[None])
    best_loss, best_x, best_cycles = hip_ctx.decompose(1, 3, SEQS, _prm(restarts=R), 1e-10)
    assert best_loss.shape == (N,) and np.all(np.isfinite(best_loss))
    for t in range(N):
        k = int(best_cycles[t])
        assert 1 <= k <= 3
        W = o.template_eval(best_x[t, : 6 * (k + 1)], [SQ] * k)
        assert abs(o.basic_cost(W, targets[t]) - best_loss[t]) < 1e-12
    if R >= 5:
        assert np.mean(best_loss < 1e-10) > 0.9


def test_window_equals_whole_batch_results(hip_ctx):
    """slam_decompose_range over two windows solves the same targets as one whole-batch call (the seeds
    are keyed on the absolute target index, so every restart starts from the same x0)."""
    N = 40
    targets = o.haar_batch(N, seed0=777)
    hip_ctx.set_targets(targets)
    hip_ctx.set_gates(SQ[None])
    prm = _prm(restarts=8, flags=0)  # no early exit: every restart runs to its own end -> deterministic
    whole = hip_ctx.decompose(1, 3, SEQS, prm, 1e-10)
    a = hip_ctx.decompose_range(0, 13, 1, 3, SEQS, prm, 1e-10)
    b = hip_ctx.decompose_range(13, 27, 1, 3, SEQS, prm, 1e-10)
    assert np.array_equal(np.concatenate([a[2], b[2]]), whole[2])
    assert np.array_equal(np.concatenate([a[0], b[0]]), whole[0])
    assert np.array_equal(np.concatenate([a[1], b[1]]), whole[1])
    # results of the untouched window stay resident and fetchable
    la, xa, ca = hip_ctx.fetch_results_range(3, 0, 13)
    assert np.array_equal(la, a[0]) and np.array_equal(ca, a[2])


def test_runs_without_early_exit_are_bitwise_reproducible(hip_ctx):
    targets = o.haar_batch(24, seed0=31)
    hip_ctx.set_targets(targets)
    hip_ctx.set_gates(o.cx_matrix()[None])
    prm = _prm(restarts=6, flags=0)
    r1 = hip_ctx.decompose(1, 3, SEQS, prm, 1e-10)
    r2 = hip_ctx.decompose(1, 3, SEQS, prm, 1e-10)
    for u, v in zip(r1, r2):
        assert np.array_equal(u, v)


def test_error_codes(hip_ctx):
    targets = o.haar_batch(2, seed0=1)
    hip_ctx.set_targets(targets)
    hip_ctx.set_gates(SQ[None])
    with pytest.raises(_ffi.SlamHipError) as e:
        hip_ctx.minimize_stage([0, 1], _prm())  # gate index outside the table
    assert e.value.code == -1 and "gate_seq" in str(e.value)
    with pytest.raises(_ffi.SlamHipError) as e:
        hip_ctx.minimize_stage([0], _prm(restarts=0))
    assert e.value.code == -1
    with pytest.raises(_ffi.SlamHipError) as e:
        hip_ctx.minimize_stage([0] * 17, _prm())  # span 17: unsupported (round 5: 6..16 run, one wavefront per item)
    assert e.value.code == -3
    with pytest.raises(_ffi.SlamHipError) as e:
        hip_ctx.minimize_stage([0], _prm(restarts=1), active=np.array([5], np.int32))
    assert e.value.code == -1
    x0 = np.zeros((2, 1, 12))
    x0[1, 0, 3] = np.inf
    with pytest.raises(_ffi.SlamHipError) as e:
        hip_ctx.minimize_stage([0], _prm(restarts=1), x0=x0)
    assert e.value.code == -1 and "x0" in str(e.value)
    with pytest.raises(_ffi.SlamHipError) as e:
        hip_ctx.eval_loss_grad([0], np.zeros((1, 12)), np.array([9], np.int32))
    assert e.value.code == -1
    with pytest.raises(_ffi.SlamHipError) as e:
        hip_ctx.decompose_range(1, 5, 1, 3, SEQS, _prm(), 1e-10)  # window past the end
    assert e.value.code == -1
    fresh = _ffi.Context(0)
    try:
        with pytest.raises(_ffi.SlamHipError) as e:
            fresh.minimize_stage([0], _prm())
        assert e.value.code == -5  # no targets / gates yet
    finally:
        fresh.close()


def test_non_unitary_and_dense_gates(hip_ctx):
    """A dense (Haar) 2Q basis gate exercises the GC_DENSE kernels end to end."""
    G = o.haar_unitary(424242)
    targets = o.haar_batch(6, seed0=5)
    hip_ctx.set_targets(targets)
    hip_ctx.set_gates(G[None])
    best_loss, best_x, best_cycles = hip_ctx.decompose(1, 3, SEQS, _prm(restarts=12), 1e-10)
    for t in range(6):
        k = int(best_cycles[t])
        W = o.template_eval(best_x[t, : 6 * (k + 1)], [G] * k)
        assert abs(o.basic_cost(W, targets[t]) - best_loss[t]) < 1e-12
    assert np.all(best_loss < 1e-10)  # a generic gate reaches Haar targets within 3 applications


def test_stats_accounting(hip_ctx):
    targets = o.haar_batch(8, seed0=3)
    hip_ctx.set_targets(targets)
    hip_ctx.set_gates(SQ[None])
    hip_ctx.reset_stats()
    out = hip_ctx.minimize_stage([0, 0], _prm(restarts=4, flags=0))
    st = hip_ctx.stats()
    assert st["evals"][2] == int(out["item_evals"].sum()) and st["items"][2] == 32
    assert st["kernel_launches"] == 1 and st["kernel_ms"] > 0 and st["kernel_ms_span"][2] == st["kernel_ms"]


def test_concurrent_contexts_give_the_single_context_results():
    """Six host threads, one context / stream each, the same batch at the same time (the bench's mode of
    operation): without early exit every thread must return bit for bit what a lone context returns."""
    import threading

    targets = o.haar_batch(200, seed0=77)
    prm = _prm(restarts=8, flags=0, seed=9)

    def run(ctx, items_per_quad=0):
        ctx.set_targets(targets)
        ctx.set_gates(SQ[None])
        p = _ffi.OptParams(restarts=8, flags=0, seed=9, items_per_quad=items_per_quad)
        return ctx.decompose(1, 3, SEQS, p, 1e-10)

    lone = run(_ffi.Context(0))
    out = [None] * 6
    ctxs = [_ffi.Context(0) for _ in range(6)]

    def work(i):
        for _ in range(3):
            out[i] = run(ctxs[i], items_per_quad=i % 4)  # launch shaping does not change results either

    th = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    [t.start() for t in th]
    [t.join() for t in th]
    for res in out:
        for u, v in zip(lone, res):
            assert np.array_equal(u, v)
    for c in ctxs:
        c.close()


def test_decompose_list_and_weyl_error_paths(hip_ctx):
    targets = o.haar_batch(12, seed0=5)
    hip_ctx.set_targets(targets)
    hip_ctx.set_gates(SQ[None])
    prm = _prm(restarts=8)
    # an explicit list in arbitrary order == the same targets through the window call
    whole = hip_ctx.decompose(1, 3, SEQS, prm, 1e-10)
    hip_ctx.decompose_list([7, 2, 9], 1, 3, SEQS, prm, 1e-10)
    bl, bx, bc = hip_ctx.fetch_results_range(3, 0, 12)
    for t in (7, 2, 9):
        assert bc[t] == whole[2][t] and bl[t] < 1e-10
    with pytest.raises(_ffi.SlamHipError, match="outside"):
        hip_ctx.decompose_list([0, 12], 1, 3, SEQS, prm, 1e-10)
    with pytest.raises(_ffi.SlamHipError, match="empty target list"):
        hip_ctx.decompose_list([], 1, 3, SEQS, prm, 1e-10)
    with pytest.raises(_ffi.SlamHipError, match="k_layout"):
        hip_ctx.decompose_list([1], 2, 2, [[0, 0]], prm, 1e-10, k_layout=1)
    # non-finite input does not hang the Jacobi sweeps; the answer is non-finite or arbitrary, not an error
    bad = targets[:2].copy()
    bad[0, 0, 0] = np.nan
    c = hip_ctx.c1c2c3(bad)
    assert c.shape == (2, 3) and np.all(np.isfinite(c[1]))
    with pytest.raises(ValueError):
        hip_ctx.c1c2c3(np.zeros((3, 4, 3), dtype=complex))
