"""Dev tool (CPU): round-5 golden fixtures of the REFERENCE-FAITHFUL optimizer path -- the oracle driven like
src/slam/optimizer.py:233-303: scipy.optimize.minimize(method="BFGS", maxiter=2500) with SciPy's own finite differences (:270-278),
sequential restarts with early break -- for the bases of BASELINE configs[3] and configs[4] and one CircuitTemplateV2 case:

  * ``iswap+b``  the mixed sequence [iSWAP, B, iSWAP][:k] (configs[3]), 16 restarts
  * ``b``        the Berkeley gate alone, 16 restarts
  * ``sweep0 / sweep24 / sweep64 / sweep100``  four ConversionGain(0, 0, gc, gg, 1) bases of the configs[4] sweep (bench.sweep_gate):
                 one that reaches no Haar target in three applications, one that reaches some, two that reach all; 16 restarts
  * ``v2``       CircuitTemplateV2(base_gates=[RiSwapGate]) -- a free alpha per gate instance -- SquareCost, spans 1..2, 4 restarts,
                 explicit start points (kept in the fixture): v2_oracle.run_reference, BFGS + finite differences

64 counter-based Haar targets (haar_philox_port / x0_philox: the ones the device generates), spans 1..3, level 1e-8.
usage: tools/make_fd_reference_r5.py [workers]   -> tests/golden/fd_reference_r5.npz   (~10 min on 6 cores)
"""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N, R, TARGET_SEED, OPT_SEED, LEVEL = 64, 16, 424242, 20261003, 1e-8
SWEEP = (0, 24, 64, 100)
V2_N, V2_R, V2_SEED = 16, 4, 99


def gates_of(basis):
    from oracle import slam_oracle as o

    if basis == "iswap+b":
        return [o.riswap_matrix(1.0), o.berkeley_matrix()]
    if basis == "b":
        return [o.berkeley_matrix()]
    if basis.startswith("sweep"):
        from bench import sweep_gate

        return [sweep_gate(int(basis[5:]))]
    raise ValueError(basis)


def one(args):
    from oracle import slam_oracle as o

    basis, idx = args
    gates = gates_of(basis)
    target = o.haar_philox_port(TARGET_SEED, idx)
    best, xk, k, stats = o.run_reference(target, gates, range(1, 4), R, LEVEL, x0_fn=lambda kk, r: o.x0_philox(OPT_SEED, idx, r, kk), analytic_jac=False)
    found = o.c1c2c3_raw(o.template_eval(xk, o.gate_sequence(gates, k)))
    return best, k, found, o.c1c2c3_raw(target), stats["nfev"]


def v2_x0(idx, k, r):
    rng = np.random.default_rng([V2_SEED, idx, k, r])
    return np.concatenate([rng.uniform(0.0, 2 * np.pi, 6 * (k + 1)), rng.uniform(0.0, 1.0, k)])


def one_v2(idx):
    from oracle import slam_oracle as o
    from oracle import v2_oracle as v

    target = o.haar_philox_port(TARGET_SEED, idx)
    best, bx, bk = v.run_reference(target, lambda k: [o.riswap_matrix] * k, 1, range(1, 3), V2_R, LEVEL, lambda k: None,
                                   lambda k, r: v2_x0(idx, k, r), square=True)
    return best, bk


def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else min(6, os.cpu_count() or 1)
    out = {"n": N, "restarts": R, "target_seed": TARGET_SEED, "opt_seed": OPT_SEED, "level": LEVEL, "sweep": np.array(SWEEP),
           "v2_n": V2_N, "v2_restarts": V2_R}
    with mp.get_context("spawn").Pool(workers) as pool:
        for basis in ("iswap+b", "b") + tuple(f"sweep{b}" for b in SWEEP):
            res = pool.map(one, [(basis, i) for i in range(N)], chunksize=1)
            out[f"{basis}_loss"] = np.array([r[0] for r in res])
            out[f"{basis}_cycles"] = np.array([r[1] for r in res], dtype=np.int32)
            out[f"{basis}_found_coords"] = np.array([r[2] for r in res])
            out[f"{basis}_target_coords"] = np.array([r[3] for r in res])
            out[f"{basis}_nfev"] = np.array([r[4] for r in res], dtype=np.int64)
            print(basis, "solved", int((out[f"{basis}_loss"] < LEVEL).sum()), "of", N, "cycles", np.bincount(out[f"{basis}_cycles"]), "mean nfev",
                  out[f"{basis}_nfev"].mean(), flush=True)
        res = pool.map(one_v2, range(V2_N), chunksize=1)
        out["v2_loss"] = np.array([r[0] for r in res])
        out["v2_cycles"] = np.array([r[1] for r in res], dtype=np.int32)
        for k in (1, 2):
            out[f"v2_x0_k{k}"] = np.array([[v2_x0(i, k, r) for r in range(V2_R)] for i in range(V2_N)])
        print("v2 solved", int((out["v2_loss"] < LEVEL).sum()), "of", V2_N, "cycles", np.bincount(out["v2_cycles"]), flush=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "fd_reference_r5.npz"), **out)


if __name__ == "__main__":
    main()
