"""GPU parity: fused loss + analytic gradient (HIP, through the C ABI) vs the NumPy oracle.

Bar: same targets, same x -> loss and gradient within 1e-12 per
evaluation (fp64 both sides; differences come from FMA contraction and sincos ulps).
"""
import numpy as np
import pytest

from oracle import slam_oracle as o

pytestmark = pytest.mark.gpu

TOL = 1e-12

GATES = {
    "cx": o.cx_matrix(),
    "sqiswap": o.riswap_matrix(0.5),
    "iswap": o.riswap_matrix(1.0),
    "b": o.berkeley_matrix(),
    "cg": o.conversion_gain_matrix(0.3, -0.7, 0.9, 0.4, 1.0),
}


def _oracle_batch(x, tof, targets, gate_mats):
    loss = np.empty(len(x))
    grad = np.empty_like(x)
    for m in range(len(x)):
        loss[m], grad[m] = o.loss_and_grad(x[m], gate_mats, targets[tof[m]])
    return loss, grad


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("gname", ["cx", "sqiswap", "b", "cg"])
def test_eval_matches_oracle(hip_ctx, k, gname):
    rng = np.random.default_rng(100 * k + len(gname))
    targets = o.haar_batch(5, seed0=777)
    hip_ctx.set_targets(targets)
    hip_ctx.set_gates(np.stack([GATES[gname]]))
    M = 37  # ragged: not a multiple of 16 quads
    n = o.n_params(k)
    x = rng.uniform(-2 * np.pi, 4 * np.pi, size=(M, n))
    tof = rng.integers(0, 5, size=M).astype(np.int32)
    loss, grad = hip_ctx.eval_loss_grad([0] * k, x, tof)
    ref_loss, ref_grad = _oracle_batch(x, tof, targets, [GATES[gname]] * k)
    assert np.max(np.abs(loss - ref_loss)) < TOL
    assert np.max(np.abs(grad - ref_grad)) < TOL


def test_eval_mixed_gate_sequence(hip_ctx):
    rng = np.random.default_rng(5)
    targets = o.haar_batch(3, seed0=4242)
    hip_ctx.set_targets(targets)
    table = np.stack([GATES["iswap"], GATES["b"], GATES["cg"]])
    hip_ctx.set_gates(table)
    seq = [0, 1, 2]
    x = rng.uniform(0, 2 * np.pi, size=(16, 24))
    tof = (np.arange(16) % 3).astype(np.int32)
    loss, grad = hip_ctx.eval_loss_grad(seq, x, tof)
    ref_loss, ref_grad = _oracle_batch(x, tof, targets, [table[i] for i in seq])
    assert np.max(np.abs(loss - ref_loss)) < TOL
    assert np.max(np.abs(grad - ref_grad)) < TOL


def test_eval_kat1_known_answer(hip_ctx):
    """KAT-1 (reference notebook scripts/decomp_trajectory.ipynb:140-162): the recorded 24
    bound parameters give BasicCost 2.2193e-09 against SWAP on the HIP path too."""
    import json, os

    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat1.json")))
    swap = np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=complex)
    hip_ctx.set_targets(swap[None])
    hip_ctx.set_gates(o.riswap_matrix(0.5)[None])
    loss, _ = hip_ctx.eval_loss_grad([0, 0, 0], np.array([kat["params"]]), np.zeros(1, np.int32))
    # SquareCost = 1 - (|t|^2 + 4)/20 with |t| = 4 (1 - BasicCost)
    t = 4 * (1 - loss[0])
    square = 1 - (t * t + 4) / 20
    assert abs(square - kat["square_cost_vs_swap"]) < 5e-15 + 1e-6 * kat["square_cost_vs_swap"]


def test_eval_large_angles_and_empty(hip_ctx):
    targets = o.haar_batch(2, seed0=99)
    hip_ctx.set_targets(targets)
    hip_ctx.set_gates(GATES["sqiswap"][None])
    # empty batch is a no-op
    loss, grad = hip_ctx.eval_loss_grad([0, 0], np.zeros((0, 18)), np.zeros(0, np.int32))
    assert loss.shape == (0,) and grad.shape == (0, 18)
    # large |x| (periodicity / range reduction)
    rng = np.random.default_rng(0)
    x = rng.uniform(-1e4, 1e4, size=(16, 18))
    x[0, :] = 3.0e7  # still the table-driven path (|x| < 2e8)
    x[1, :] = 5.0e8  # beyond it: out-of-line libm path
    tof = np.zeros(16, np.int32)
    loss, grad = hip_ctx.eval_loss_grad([0, 0], x, tof)
    ref_loss, ref_grad = _oracle_batch(x, tof, targets, [GATES["sqiswap"]] * 2)
    assert np.max(np.abs(loss - ref_loss)) < 1e-10
    assert np.max(np.abs(grad - ref_grad)) < 1e-10


def test_square_cost_kat1_and_gradient(hip_ctx):
    """SquareCost on the HIP path: KAT-1's recorded value (decomp_trajectory.ipynb:87,
    3.550885807612758e-09 against SWAP) and the gradient against the oracle."""
    import json, os

    from slam_decomposition_amd import _ffi

    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat1.json")))
    swap = np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=complex)
    sq = o.riswap_matrix(0.5)
    try:
        hip_ctx.set_cost(_ffi.COST_SQUARE)
        hip_ctx.set_targets(swap[None])
        hip_ctx.set_gates(sq[None])
        loss, _ = hip_ctx.eval_loss_grad([0, 0, 0], np.array([kat["params"]]), np.zeros(1, np.int32))
        # 3.55e-9 = 0.8 (2L - L^2) with L = 1 - |t|/4: one ulp of |t|/4 is 3.6e-16 here, and the parameters
        # are printed with 15-16 digits
        assert abs(loss[0] - kat["square_cost_vs_swap"]) < 3e-15
        rng = np.random.default_rng(8)
        targets = o.haar_batch(3, seed0=66)
        hip_ctx.set_targets(targets)
        x = rng.uniform(0, 2 * np.pi, size=(9, 18))
        tof = (np.arange(9) % 3).astype(np.int32)
        loss, grad = hip_ctx.eval_loss_grad([0, 0], x, tof)
        for m in range(9):
            v, g = o.square_loss_and_grad(x[m], [sq, sq], targets[tof[m]])
            W = o.template_eval(x[m], [sq, sq])
            assert abs(o.square_cost(W, targets[tof[m]]) - v) < 1e-15
            assert abs(loss[m] - v) < 1e-12 and np.max(np.abs(grad[m] - g)) < 1e-12
        with pytest.raises(_ffi.SlamHipError, match="Unrecognized Cost Function"):
            hip_ctx.set_cost(7)
    finally:
        hip_ctx.set_cost(_ffi.COST_BASIC)
