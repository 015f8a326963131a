"""bench: workloads, the flop accounting of SURVEY.md 8(d), synthetic inputs."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PEAK_FP64_VALU_TFLOPS = 78.6  # 256 CU x 4 SIMD x 16 fp64 FMA lanes/clk x 2 flop x 2.4 GHz (MI355X_MICROARCH.md)
SUCCESS_LOSS = 1e-8  # BASELINE.json metric: loss < 1e-8
TARGET_SEED0 = 20260000
OPT_SEED = 20261003


def f_eval(k: int) -> int:
    """Algorithmic flops of one fused loss+gradient evaluation: dense accounting."""
    return 3036 * k + 1247


def f_eval_v2(k: int) -> int:
    """Parametrised-gate templates (CircuitTemplateV2): F_eval(k) plus, per gate, the four raw-angle derivatives
    Re(u (dG/d angle) h) over the four columns -- 4 angles x 4 columns x (2x2 complex block times a 2-vector: 22 flop, real part
    of the 2-term complex dot: 8 flop) = 480 flop -- and the gate's two block entries from its trig values (8 flop): 488 k."""
    return f_eval(k) + 488 * k


def f_forward(k: int) -> int:
    """Forward chain + loss only (what a rejected line-search trial is worth)."""
    return 1080 * k + 251


WORKLOADS = {
    # name: (gate builder name, targets per step, restarts, description)
    "cfg2": ("cx", 1024, 16, "BASELINE configs[1]: CNOT basis span<=3, 1024 Haar targets x 16 restarts, fp64"),
    "cfg3": ("sqiswap", 65536, 32, "BASELINE configs[2]: sqrt(iSWAP) basis span<=3, 65536 Haar targets x 32 restarts, fp64"),
    "cfg4": ("iswap+b", 32768, 16, "BASELINE configs[3] per-GPU shard: iSWAP + B mixed basis, 32768 Haar targets x 16 restarts"),
    # one step = one basis gate of this GPU's 16 (of 128) against the 4096 shared targets
    "cfg5": ("cgsweep", 4096, 16, "BASELINE configs[4] per-GPU shard: 16 of 128 ConversionGain(0,0,gc,gg,1) bases x 4096 shared Haar targets x 16 restarts"),
}
SWEEP_BASES_PER_GPU = 16
SWEEP_CPU_BASIS = 64  # m = 9/32, p = 0: the basis the CPU baseline of cfg5 runs


PER_SPAN_WARM_STEPS = 3  # untimed steps before the single-stream per-span pass


def _batches_in_flight(items_per_stage: int, span_rules: bool) -> int:
    """Library calls kept in flight per GPU.  Measured on MI355X (sqrt(iSWAP) x 32 restarts, equal total work, tools/r4_strong_regime.sh;
    decompositions/s relative to 65 536 targets x 5 in flight): 8192 targets -- the per-GPU batch of `--scaling strong` on 8 GPUs --
    x 8 / 12 / 16 in flight 0.80 / 0.84 / 0.88, 16 384 x 8 / 12 0.93 / 0.95, 32 768 x 5 / 8 0.96 / 0.98."""
    if span_rules:
        return 8
    if items_per_stage <= (1 << 18):
        return 16
    if items_per_stage <= (1 << 19):
        return 12
    if items_per_stage <= (1 << 20):
        return 8
    return 5


def sweep_gate(b: int) -> np.ndarray:
    """Basis b of the 128-gate parametric-Hamiltonian sweep (SURVEY.md §8(d) cfg 5, shaped like build_gates(),
    utils/gates/bare_candidates.py:47-69): gc = p m pi, gg = (1 - p) m pi, 16 values of m in (0, 0.5] x 8 of p in [0, 1]."""
    from slam_decomposition_amd import gates as G

    m = 0.5 * (b // 8 + 1) / 16
    pfrac = (b % 8) / 7
    return G.ConversionGainGate(0.0, 0.0, pfrac * m * np.pi, (1 - pfrac) * m * np.pi, 1.0).to_matrix()


def gate_table(name: str) -> np.ndarray:
    from slam_decomposition_amd import gates as G

    if name == "cx":
        return np.stack([G.CXGate().to_matrix()])
    if name == "sqiswap":
        return np.stack([G.RiSwapGate(0.5).to_matrix()])
    if name == "iswap+b":
        return np.stack([G.RiSwapGate(1.0).to_matrix(), G.BerkeleyGate().to_matrix()])
    if name == "cgsweep":
        return np.stack([sweep_gate(0)])
    raise ValueError(name)


def make_targets(n: int, seed0: int) -> np.ndarray:
    """T_i = unitary_group.rvs(4, default_rng(seed0 + i))."""
    from slam_decomposition_amd.sampler import random_unitary

    return np.stack([random_unitary(4, seed=seed0 + i) for i in range(n)])
