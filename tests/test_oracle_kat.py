"""CPU: the oracle against the reference's recorded known-answer data.

The reference has no tests of its own (src/tests/main_test.py is a placeholder); these notebook
outputs are the only pinned values for the hot path.
"""
import json
import os

import numpy as np
import pytest

from oracle import slam_oracle as o

HERE = os.path.dirname(__file__)
KAT1 = json.load(open(os.path.join(HERE, "golden", "kat1.json")))
SWAP = np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=complex)
SQ = o.riswap_matrix(0.5)


def test_kat1_forward_chain_loss_and_coordinates():
    """scripts/decomp_trajectory.ipynb:140-162 (params) -> :87 SquareCost, :90/:213 c1c2c3 of the
    full circuit, :234-235 c1c2c3 of circuit[:8] and circuit[3:].  Pins qiskit's U matrix, the
    little-endian Kronecker order, the RiSwap matrix and the c1c2c3 algorithm + rounding."""
    x = KAT1["params"]
    W = o.template_eval(x, [SQ] * 3)
    # params are printed with 15 significant digits -> the loss agrees to ~1e-7 relative
    assert abs(o.square_cost(W, SWAP) - KAT1["square_cost_vs_swap"]) < 1e-15
    assert o.c1c2c3(W) == tuple(KAT1["c1c2c3_full"])
    assert o.c1c2c3(o.template_eval(x[:18], [SQ] * 2)) == tuple(KAT1["c1c2c3_first8"])
    assert o.c1c2c3(o.template_eval(x[6:], [SQ] * 2)) == tuple(KAT1["c1c2c3_from3"])
    assert np.allclose(W.conj().T @ W, np.eye(4), atol=1e-14)


def test_kat5_named_gate_coordinates():
    """KAT-5: SWAP (decomp_trajectory.ipynb:84), CX-class (local_smush_test.ipynb:179), iSWAP family,
    B gate, FSim(pi/2, pi/6) (fsim_continuous.ipynb:104), CX.SWAP.CX (pm_runner.ipynb:154)."""
    assert o.c1c2c3(SWAP) == (0.5, 0.5, 0.5)
    assert o.c1c2c3(o.cx_matrix()) == (0.5, 0.0, 0.0)
    assert o.c1c2c3(o.riswap_matrix(1.0)) == (0.5, 0.5, 0.0)
    assert o.c1c2c3(SQ) == (0.25, 0.25, 0.0)
    assert o.c1c2c3(o.berkeley_matrix()) == (0.5, 0.25, 0.0)
    th, ph = np.pi / 2, np.pi / 6
    fsim = np.array(
        [[1, 0, 0, 0], [0, np.cos(th), -1j * np.sin(th), 0], [0, -1j * np.sin(th), np.cos(th), 0], [0, 0, 0, np.exp(-1j * ph)]]
    )
    assert o.c1c2c3(fsim) == (0.5, 0.5, 0.08333333)
    # three alternating CNOTs are a SWAP (the pm_runner circuit is only recorded as an image; its
    # printed coordinates are (0.5, 0.5, 0.5))
    cx01 = o.cx_matrix()
    cx10 = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], dtype=complex)
    assert o.c1c2c3(cx01 @ cx10 @ cx01) == (0.5, 0.5, 0.5)


def test_kat3_readme_target_span_rule():
    """README.md:55-62: Haar target (0.58941013, 0.22184674, 0.11209285) is solved at k = 2 with
    sqrt(iSWAP); mirrored into c1 <= 1/2 it satisfies |z| <= x - y
    (utils/transpiler_pass/weyl_decompose.py:348)."""
    c1, c2, c3 = 0.58941013, 0.22184674, 0.11209285
    x, y, z = 1 - c1, c2, -c3
    assert abs(z) <= x - y
    # the canonical gate with these coordinates really is reachable with two sqrt(iSWAP)
    T = o.canonical_matrix(c1, c2, c3)
    best = min(
        o.run_reference(T, [SQ], range(2, 3), 6, 1e-8, x0_fn=lambda k, r: o.x0_philox(5, 0, r, k), analytic_jac=True)[0]
        for _ in range(1)
    )
    assert best < 1e-8


def test_conversion_gain_closed_form_equals_expm():
    rng = np.random.default_rng(0)
    for _ in range(20):
        p = rng.uniform(-3, 3, 5)
        assert np.max(np.abs(o.conversion_gain_matrix(*p) - o.conversion_gain_matrix_expm(*p))) < 5e-15
    # named gates of utils/gates/parallel_drive_volume.py:91-96 as (gc, gg, t)
    assert o.c1c2c3(o.conversion_gain_matrix(0, 0, np.pi / 2, 0, 1)) == (0.5, 0.5, 0.0)
    assert o.c1c2c3(o.conversion_gain_matrix(0, 0, np.pi / 2, 0, 0.5)) == (0.25, 0.25, 0.0)
    assert o.c1c2c3(o.conversion_gain_matrix(0, 0, np.pi / 4, np.pi / 4, 1)) == (0.5, 0.0, 0.0)
    assert o.c1c2c3(o.conversion_gain_matrix(0, 0, 3 * np.pi / 8, np.pi / 8, 1)) == (0.5, 0.25, 0.0)


@pytest.mark.parametrize("k", [1, 2, 3, 5])
def test_analytic_gradient_matches_central_differences(k):
    """analytic dloss/dx vs central differences of the CPU loss."""
    rng = np.random.default_rng(k)
    gates = [o.cx_matrix(), SQ, o.berkeley_matrix(), o.conversion_gain_matrix(0.2, 0.4, 0.7, 0.3), o.riswap_matrix(1.0)]
    for trial in range(3):
        T = o.haar_unitary(100 + trial)
        gs = [gates[(trial + j) % len(gates)] for j in range(k)]
        x = rng.uniform(-7, 7, o.n_params(k))
        val, grad = o.loss_and_grad(x, gs, T)
        assert abs(val - o.loss(x, gs, T)) < 1e-15
        fd = o.fd_grad(x, gs, T, h=1e-6)
        assert np.max(np.abs(grad - fd)) < 2e-9


def test_haar_sampler_semantics():
    """src/slam/sampler.py:62-71: an integer seed re-seeds every draw -> identical unitaries."""
    a = o.haar_sample_reference(7, 3)
    assert np.array_equal(a[0], a[1]) and np.array_equal(a[1], a[2])
    assert np.allclose(a[0].conj().T @ a[0], np.eye(4), atol=1e-14)
    b = o.haar_sample_reference(8, 1)
    assert not np.allclose(a[0], b[0])


def test_philox_known_answer():
    """Philox4x32-10 known-answer vectors from the Random123 distribution (kat_vectors)."""
    z = np.zeros((1, 4), dtype=np.uint32)
    assert [hex(v) for v in o.philox4x32(z, (0, 0))[0]] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    f = np.full((1, 4), 0xFFFFFFFF, dtype=np.uint32)
    assert [hex(v) for v in o.philox4x32(f, (0xFFFFFFFF, 0xFFFFFFFF))[0]] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    pi = np.array([[0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344]], dtype=np.uint32)
    assert [hex(v) for v in o.philox4x32(pi, (0xA4093822, 0x299F31D0))[0]] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]
    x = o.x0_philox(3, 5, 1, 2)
    assert x.shape == (18,) and np.all((x >= 0) & (x < 2 * np.pi))


def test_config0_reference_path_cnot_span2():
    """BASELINE.json configs[0] on the reference CPU path (plumbing): CNOT, maximum_span_guess=2,
    1 Haar target, 1 restart.  Unreachable with <= 2 CNOTs: ValueError (optimizer.py:89-93) unless
    override_fail, then a non-zero converged loss and a DataDictEntry with success_label 0."""
    T = o.haar_unitary(o.BENCH_TARGET_SEED0)
    x0_fn = lambda k, r: o.x0_philox(1, 0, r, k)  # noqa: E731
    with pytest.raises(ValueError, match="Failed to converge within error threshold"):
        o.approximate_target_U(T, [o.cx_matrix()], maximum_span_guess=2, training_restarts=1, x0_fn=x0_fn)
    td = o.approximate_target_U(T, [o.cx_matrix()], maximum_span_guess=2, training_restarts=1, x0_fn=x0_fn, override_fail=True)
    assert td.success_label == 0 and 1e-6 < td.loss_result < 0.2 and td.cycles in (1, 2)
    assert len(td.Xk) == 6 * (td.cycles + 1)


def test_bfgs_port_agrees_with_scipy_bfgs():
    """The CPU port of the kernel's iteration ends in the same minima as SciPy's BFGS."""
    import scipy.optimize as opt

    from oracle.bfgs_port import minimize_port

    g = o.cx_matrix()
    agree = 0
    for t in range(4):
        T = o.haar_unitary(50 + t)
        for k in (2, 3):
            x0 = o.x0_philox(11, t, 0, k)
            f, x, it, st, nev = minimize_port(x0, [g] * k, T)
            res = opt.minimize(lambda xx: o.loss_and_grad(xx, [g] * k, T), x0, jac=True, method="BFGS", options={"gtol": 1e-9})
            assert st in (0, 4)
            assert abs(o.loss(x, [g] * k, T) - f) < 1e-14
            if k == 3:
                assert f < 1e-12 and res.fun < 1e-10
            agree += abs(f - res.fun) < 1e-6
    assert agree >= 6


def test_philox_haar_port_is_unitary_and_haar_like():
    """The NumPy restatement of the device sampler: unitary to rounding, Haar second moments."""
    U = np.stack([o.haar_philox_port(11, i) for i in range(3000)])
    err = np.abs(np.einsum("nji,njk->nik", U.conj(), U) - np.eye(4)).max()
    assert err < 5e-15
    p = np.abs(U) ** 2
    assert abs(p.mean() - 0.25) < 1e-12  # rows sum to one exactly
    assert abs((p**2).mean() - 0.1) < 4e-3
    assert not np.allclose(o.haar_philox_port(11, 0), o.haar_philox_port(12, 0))


def test_kat1_through_the_v2_restatement():
    """The same recorded run is a CircuitTemplateV2 one (decomp_trajectory.ipynb:84-90: base_gates=[RiSwapGate], every Q
    bounded to [0.5, 0.5]): the V2 restatement (oracle/v2_oracle.py) with Q = 0.5 gives the recorded loss, and its
    analytic gradient -- including d/d alpha -- is the derivative of that loss."""
    from oracle import v2_oracle as v

    x = np.concatenate([KAT1["params"], [0.5, 0.5, 0.5]])
    fns = [lambda a: o.riswap_matrix(a)] * 3
    W = v.template_eval(x, fns, 1, 3)
    assert abs(o.square_cost(W, SWAP) - KAT1["square_cost_vs_swap"]) < 1e-15
    assert o.c1c2c3(W) == tuple(KAT1["c1c2c3_full"])
    gmap = ([0, -1, -1, -1], [-0.5 * np.pi, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0])  # RiSwap(alpha) = CG(a = -pi alpha / 2)
    f, g = v.loss_and_grad(x, [gmap] * 3, 1, 3, SWAP, square=True)
    assert abs(f - KAT1["square_cost_vs_swap"]) < 1e-15
    assert np.max(np.abs(g - v.fd_grad(x, fns, 1, 3, SWAP, square=True))) < 1e-9


def test_kat1_riswap_sweep_pins_the_gate_parameter_path():
    """decomp_trajectory.ipynb cell 12 (tests/golden/kat1_riswap_sweep.json, tools/make_kat_sweep.py): 25 recorded
    ``c1c2c3`` triples of the KAT-1 circuit with the LAST RiSwapGate at alpha = t, t = linspace(0, 0.5, 25), mirrored on the
    x axis as cell 10 does.  The only recorded data with RiSwapGate(alpha != 1/2): it pins the gate-parameter forward
    path of the V2 restatement (custom_gates.py:582-595 through basisv2.py:262-287) digit for digit."""
    from oracle import v2_oracle as v

    sweep = json.load(open(os.path.join(HERE, "golden", "kat1_riswap_sweep.json")))
    fns = [lambda a: o.riswap_matrix(a)] * 3
    ts = np.linspace(0, 0.5, 25)
    assert len(sweep["c1c2c3"]) == len(ts)
    for t, want in zip(ts, sweep["c1c2c3"]):
        x = np.concatenate([KAT1["params"], [0.5, 0.5, t]])
        c = list(o.c1c2c3(v.template_eval(x, fns, 1, 3)))
        if c[0] > 0.5:
            c[0] = -1 * c[0] + 1  # "eliminating x-axis symmetry" (cell 10)
        assert [round(float(v_), 8) for v_ in c] == want, (t, c, want)  # exact to the 8 recorded digits
    # the sweep's ends are the two triples KAT-1 already holds
    assert sweep["c1c2c3"][0] == KAT1["c1c2c3_first8"] and sweep["c1c2c3"][-1] == KAT1["c1c2c3_full"]
