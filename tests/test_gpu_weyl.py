"""GPU: batched Weyl-chamber coordinates on the device against the oracle's
restatement of weylchamber.c1c2c3 (LAPACK eigenvalues) and the NumPy port of the device algorithm."""
import json
import os

import numpy as np
import pytest
from scipy.stats import unitary_group

from oracle import slam_oracle as o
from slam_decomposition_amd.weyl import c1c2c3_batch

pytestmark = pytest.mark.gpu

SWAP = np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=complex)


def _canon(c):
    """On the c3 = 0 face (c1, c2, 0) and (1 - c1, c2, 0) are the same class, and which one weylchamber returns
    depends on the sign of a rounding error: compare modulo that."""
    c = np.array(c, dtype=float)
    if abs(c[2]) < 5e-9:
        c[0] = min(c[0], 1.0 - c[0])
    return c


def test_haar_targets_match_reference_algorithm(hip_ctx):
    U = o.haar_batch(600, seed0=123)
    dev = hip_ctx.c1c2c3(U)
    for i in range(len(U)):
        ref = o.c1c2c3(U[i])
        assert tuple(dev[i]) == tuple(float(v) for v in ref), (i, dev[i], ref)
        assert tuple(dev[i]) == o.c1c2c3_jacobi_port(U[i])
    # unrounded: agreement with LAPACK to 1e-13
    raw = hip_ctx.c1c2c3(U[:100], ndigits=-1)
    ref_raw = np.array([o.c1c2c3(u, ndigits=15) for u in U[:100]])
    assert np.max(np.abs(raw - ref_raw)) < 1e-13
    # the host batch implementation used so far gives the same numbers
    assert np.array_equal(dev, c1c2c3_batch(U))


def test_known_gates_and_their_local_equivalents(hip_ctx):
    named = {
        "cx": (o.cx_matrix(), (0.5, 0.0, 0.0)),
        "swap": (SWAP, (0.5, 0.5, 0.5)),
        "iswap": (o.riswap_matrix(1.0), (0.5, 0.5, 0.0)),
        "sqiswap": (o.riswap_matrix(0.5), (0.25, 0.25, 0.0)),
        "b": (o.berkeley_matrix(), (0.5, 0.25, 0.0)),
        "identity": (np.eye(4, dtype=complex), (0.0, 0.0, 0.0)),
        "cg": (o.conversion_gain_matrix(0, 0, 3 * np.pi / 8, np.pi / 8, 1), (0.5, 0.25, 0.0)),
    }
    got = hip_ctx.c1c2c3(np.stack([g for g, _ in named.values()]))
    for (name, (_, want)), c in zip(named.items(), got):
        assert np.allclose(_canon(c), _canon(want), atol=1e-8), (name, c, want)
    # degenerate spectra in general position: random local gates and a global phase around each
    rng = np.random.default_rng(0)
    Us, want = [], []
    for name, (G, c) in named.items():
        for _ in range(40):
            L = np.kron(unitary_group.rvs(2, random_state=rng), unitary_group.rvs(2, random_state=rng))
            R = np.kron(unitary_group.rvs(2, random_state=rng), unitary_group.rvs(2, random_state=rng))
            Us.append(L @ G @ R * np.exp(1j * rng.uniform(0, 2 * np.pi)))
            want.append(c)
    got = hip_ctx.c1c2c3(np.stack(Us))
    for u, c, w in zip(Us, got, want):
        assert np.allclose(_canon(c), _canon(w), atol=2e-8), (c, w)
        assert np.allclose(_canon(c), _canon(o.c1c2c3_jacobi_port(u)), atol=2e-8)


def test_kat1_coordinates_and_eval_c1c2c3(hip_ctx):
    """KAT-1 (scripts/decomp_trajectory.ipynb:90,213): the recorded 24 parameters give
    c1c2c3 = (0.49999821, 0.49997201, 0.49996939); here the template unitary never leaves the device."""
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat1.json")))
    hip_ctx.set_targets(SWAP[None])
    hip_ctx.set_gates(o.riswap_matrix(0.5)[None])
    c = hip_ctx.eval_c1c2c3([0, 0, 0], np.array([kat["params"]]))
    assert tuple(c[0]) == tuple(kat["c1c2c3_full"])
    # a batch of random parameter vectors against eval_unitary + oracle
    rng = np.random.default_rng(4)
    x = rng.uniform(0, 2 * np.pi, size=(33, 18))
    c = hip_ctx.eval_c1c2c3([0, 0], x)
    W, _ = hip_ctx.eval_unitary([0, 0], x)
    for m in range(33):
        assert tuple(c[m]) == tuple(float(v) for v in o.c1c2c3(W[m]))


def test_resident_targets_and_empty(hip_ctx):
    hip_ctx.sample_haar(5, 300)
    T = hip_ctx.get_targets()
    c = hip_ctx.targets_c1c2c3()
    assert np.array_equal(c, hip_ctx.c1c2c3(T))
    assert np.array_equal(hip_ctx.targets_c1c2c3(100, 50), c[100:150])
    assert hip_ctx.c1c2c3(np.zeros((0, 4, 4), dtype=complex)).shape == (0, 3)
    with pytest.raises(Exception):
        hip_ctx.targets_c1c2c3(290, 20)
