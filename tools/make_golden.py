"""Generate tests/golden/hotpath_golden.npz from the CPU oracle (after it passed KAT-1).

The reference itself cannot run here (qiskit / weylchamber / qutip / monodromy are not installed and
there is no network), so the golden vectors come from the NumPy/SciPy restatement in oracle/, which is
pinned by the reference's recorded notebook outputs (tests/test_oracle_kat.py).  Run from the repo root:
    python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import slam_oracle as o  # noqa: E402


def main():
    rng = np.random.default_rng(20261003)
    targets = np.stack([o.haar_unitary(s) for s in range(8)])  # unitary_group.rvs(4, default_rng(s)), s = 0..7
    coords = np.array([o.c1c2c3(t) for t in targets])
    gates = {"cx": o.cx_matrix(), "sqiswap": o.riswap_matrix(0.5), "iswap": o.riswap_matrix(1.0), "b": o.berkeley_matrix()}
    out = {"targets": targets, "target_c1c2c3": coords, "gate_names": np.array(list(gates))}
    for gi, (name, g) in enumerate(gates.items()):
        out[f"gate_{name}"] = g
        for k in (1, 2, 3):
            x = rng.uniform(0, 2 * np.pi, o.n_params(k))
            T = targets[(gi + k) % 8]
            W = o.template_eval(x, [g] * k)
            val, grad = o.loss_and_grad(x, [g] * k, T)
            out[f"x_{name}_{k}"] = x
            out[f"tidx_{name}_{k}"] = np.int64((gi + k) % 8)
            out[f"W_{name}_{k}"] = W
            out[f"loss_{name}_{k}"] = np.float64(val)
            out[f"grad_{name}_{k}"] = grad
            out[f"fdgrad_{name}_{k}"] = o.fd_grad(x, [g] * k, T)
    # converged (loss, k) per target from the reference loop (SciPy BFGS, analytic jac, 6 restarts)
    conv = {}
    for name in ("cx", "sqiswap", "b"):
        res = []
        for t in range(8):
            bl, bx, bk, _ = o.run_reference(
                targets[t], [gates[name]], range(1, 4), 6, 1e-8, x0_fn=lambda k, r, t=t: o.x0_philox(77, t, r, k), analytic_jac=True
            )
            res.append((bl, bk))
        conv[name] = np.array(res)
        out[f"converged_{name}"] = conv[name]
    # Philox seeds
    out["x0_seed77_t3_r2_k3"] = o.x0_philox(77, 3, 2, 3)
    np.savez(os.path.join(ROOT, "tests", "golden", "hotpath_golden.npz"), **out)
    print({k: v[:, 1].astype(int).tolist() for k, v in conv.items()})


if __name__ == "__main__":
    main()
