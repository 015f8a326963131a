"""Dev tool (CPU): the coverage region of a TWO-gate product g2 . L . g1 (L over all local unitaries SU(2) x SU(2)) in Weyl
coordinates -- what monodromy's polytope for the pair describes (src/slam/utils/polytopes/polytope_wrap.py:39-94) -- by sampling.

usage: tools/fit_two_gate_region.py <gate1> <gate2> [samples]     gates: iswap | b | cx | sqiswap | riswap:<alpha> | cg:<gc>:<gg>

Prints the range of the linear functionals x, y, |z|, x +- y +- |z| of the folded coordinates (x >= y >= |z|, x <= 1/2) over the
samples -- the supporting half-spaces of the region are read off these -- and, for a region hypothesis given as a Python
expression in x, y, z (``--region "(x >= 0.25) & (abs(z) <= 0.25)"``), checks that it is FILLED (no empty cell of a 0.02 grid
strictly inside) and that no sample lies outside.  span_rules.two_gate_region holds the regions obtained this way; the GPU test
tests/test_gpu_round4.py checks them against the brute-force span loop.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_decomposition_amd import gates as G  # noqa: E402
from slam_decomposition_amd.weyl import c1c2c3_batch  # noqa: E402


def gate(spec: str) -> np.ndarray:
    if spec == "iswap":
        return G.RiSwapGate(1.0).to_matrix()
    if spec == "sqiswap":
        return G.RiSwapGate(0.5).to_matrix()
    if spec == "b":
        return G.BerkeleyGate().to_matrix()
    if spec == "cx":
        return G.CXGate().to_matrix()
    if spec.startswith("riswap:"):
        return G.RiSwapGate(float(spec.split(":")[1])).to_matrix()
    if spec.startswith("cg:"):
        _, gc, gg = spec.split(":")
        return G.ConversionGainGate(0.0, 0.0, float(gc), float(gg), 1.0).to_matrix()
    raise SystemExit(f"unknown gate {spec}")


def su2(n, rng):
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    a, b = q[:, 0] + 1j * q[:, 1], q[:, 2] + 1j * q[:, 3]
    u = np.empty((n, 2, 2), complex)
    u[:, 0, 0], u[:, 0, 1], u[:, 1, 0], u[:, 1, 1] = a, -b.conj(), b, a.conj()
    return u


def sample_products(g1, g2, n, seed=0):
    out = []
    for s in range(0, n, 100000):
        m = min(100000, n - s)
        rng = np.random.default_rng(seed + s)
        loc = np.einsum("nij,nkl->nikjl", su2(m, rng), su2(m, rng)).reshape(m, 4, 4)
        out.append(c1c2c3_batch(g2[None] @ loc @ g1[None], ndigits=12))
    return np.concatenate(out)


def fold(c):
    f = c.copy()
    m = f[:, 0] > 0.5
    f[m, 0] = 1.0 - f[m, 0]
    f[m, 2] = -f[m, 2]
    return f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("gate1")
    ap.add_argument("gate2")
    ap.add_argument("samples", nargs="?", type=int, default=400000)
    ap.add_argument("--region", default=None)
    a = ap.parse_args()
    f = fold(sample_products(gate(a.gate1), gate(a.gate2), a.samples))
    x, y, z = f.T
    az = np.abs(z)
    for name, v in (("x", x), ("y", y), ("|z|", az), ("x+y", x + y), ("x-y", x - y), ("x+y+|z|", x + y + az), ("x+y-|z|", x + y - az),
                    ("x-y+|z|", x - y + az), ("x-y-|z|", x - y - az)):
        print(f"{name:9s} in [{v.min():.4f}, {v.max():.4f}]")
    if a.region:
        inside = eval(a.region, {"x": x, "y": y, "z": z, "abs": np.abs, "np": np})
        print("samples outside the hypothesis:", int((~inside).sum()))
        h = 0.02
        occ = set(map(tuple, np.floor(f / h).astype(int)))
        cells = holes = 0
        for i in range(int(0.5 / h) + 1):
            for j in range(int(0.5 / h) + 1):
                for k in range(-int(0.5 / h) - 1, int(0.5 / h) + 1):
                    cx, cy, cz = (i + 0.5) * h, (j + 0.5) * h, (k + 0.5) * h
                    corners = np.array([[cx + sx * h, cy + sy * h, cz + sz * h] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])
                    X, Y, Z = corners.T
                    ok = (X <= 0.5) & (Y <= X) & (np.abs(Z) <= Y) & eval(a.region, {"x": X, "y": Y, "z": Z, "abs": np.abs, "np": np})
                    if ok.all():  # the cell and its neighbourhood lie strictly inside chamber and hypothesis
                        cells += 1
                        holes += (i, j, k) not in occ
        print(f"grid cells strictly inside the hypothesis: {cells}, without a sample: {holes}")


if __name__ == "__main__":
    main()
