"""Dev tool (GPU, run under rocprofv3 --pmc SQ_INSTS_VALU by tools/r5_v2_valu.sh): a few span loops of the fixed-gate path and of the
CircuitTemplateV2 path on the same targets; prints the evaluations per span of ALL launches of the process (the counters cover them all)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slam_decomposition_amd import _ffi
from slam_decomposition_amd.basisv2 import CircuitTemplateV2
from slam_decomposition_amd.gates import RiSwapGate

N, R = 32768, 16
ctx = _ffi.Context(0)
ctx.sample_haar(20260777, N)
prm = _ffi.OptParams(restarts=R, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=5, flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED | _ffi.FLAG_NO_OVERLAP)
out = {}
# fixed gates
ctx.set_gates(np.stack([RiSwapGate(0.5).to_matrix()]))
ctx.set_cost(_ffi.COST_BASIC)
ctx.reset_stats()
for _ in range(2):
    ctx.decompose_range(0, N, 1, 3, [[0], [0, 0], [0, 0, 0]], prm, 1e-10, fetch=False)
ctx.synchronize()
out["fixed"] = ctx.stats()["evals"][:4]
# V2, RiSwap class with a free alpha per gate
basis = CircuitTemplateV2(base_gates=[RiSwapGate], maximum_span_guess=3)
ctx.v2_set_gates(basis._gate_maps)
ctx.set_cost(_ffi.COST_SQUARE)
layouts = {}
for k in (1, 2, 3):
    basis.build(k)
    layouts[k] = basis.device_layout(k)
ctx.reset_stats()
for _ in range(2):
    ctx.v2_decompose_range(0, N, 1, 3, [[0] * k for k in (1, 2, 3)], [layouts[k][2:6] for k in (1, 2, 3)], prm, 1e-10)
out["v2"] = ctx.stats()["evals"][:4]
print(json.dumps(out))
