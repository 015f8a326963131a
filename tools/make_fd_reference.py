"""Dev tool (CPU): golden fixture of the REFERENCE-FAITHFUL optimizer path for the GPU parity test --
``oracle.run_reference(..., analytic_jac=False)``: scipy.optimize.minimize(method="BFGS", maxiter=2500) with SciPy's own 2-point finite
differences, sequential restarts with early break (src/slam/optimizer.py:233-303, :270-278) -- on 64 Haar targets each for the
CNOT and sqrt(iSWAP) bases, spans 1..3, 8 restarts, success level 1e-8 (BASELINE.json's metric).  Targets and start points are the
counter-based ones the device generates (haar_philox_port / x0_philox), so the GPU test runs the SAME problems.

usage: tools/make_fd_reference.py      -> tests/golden/fd_reference.npz   (~1 min on 8 cores)
"""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N, R, TARGET_SEED, OPT_SEED, LEVEL = 64, 8, 424242, 20261003, 1e-8


def one(args):
    from oracle import slam_oracle as o

    basis, idx = args
    gates = [o.cx_matrix()] if basis == "cx" else [o.riswap_matrix(0.5)]
    target = o.haar_philox_port(TARGET_SEED, idx)
    best, xk, k, stats = o.run_reference(target, gates, range(1, 4), R, LEVEL, x0_fn=lambda kk, r: o.x0_philox(OPT_SEED, idx, r, kk), analytic_jac=False)
    found = o.c1c2c3_raw(o.template_eval(xk, o.gate_sequence(gates, k)))
    return best, k, found, o.c1c2c3_raw(target), stats["nfev"]


def main():
    out = {"n": N, "restarts": R, "target_seed": TARGET_SEED, "opt_seed": OPT_SEED, "level": LEVEL}
    with mp.get_context("spawn").Pool(min(8, os.cpu_count() or 1)) as pool:
        for basis in ("cx", "sqiswap"):
            res = pool.map(one, [(basis, i) for i in range(N)], chunksize=1)
            out[f"{basis}_loss"] = np.array([r[0] for r in res])
            out[f"{basis}_cycles"] = np.array([r[1] for r in res], dtype=np.int32)
            out[f"{basis}_found_coords"] = np.array([r[2] for r in res])
            out[f"{basis}_target_coords"] = np.array([r[3] for r in res])
            out[f"{basis}_nfev"] = np.array([r[4] for r in res], dtype=np.int64)
            print(basis, "solved", int((out[f"{basis}_loss"] < LEVEL).sum()), "of", N, "cycles", np.bincount(out[f"{basis}_cycles"]), "mean nfev", out[f"{basis}_nfev"].mean())
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "fd_reference.npz"), **out)


if __name__ == "__main__":
    main()
