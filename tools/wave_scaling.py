"""Dev tool (GPU): steady-state evaluation rate with 1 and 2 wavefronts per SIMD.  Every item runs the
same number of iterations (maxiter small, tolerances off), so there is no straggler tail."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from slam_decomposition_amd import _ffi
from bench import gate_table, make_targets, f_eval

gname = sys.argv[1] if len(sys.argv) > 1 else "sqiswap"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
N = 2048 * 16 * 8
ctx = _ffi.Context(0)
table = gate_table(gname)
ctx.set_gates(table)
ctx.set_targets(make_targets(N, 20260000))
for k in (1, 2, 3):
    seq = [i % len(table) for i in range(k)]
    for ipq, label in ((8, "2 waves/SIMD"), (16, "1 wave/SIMD"), (32, "1 wave on half the SIMDs"))[: int(os.environ.get("WS_CASES", "3"))]:
        prm = _ffi.OptParams(restarts=1, maxiter=iters, gtol=0.0, gtol_far=0.0, stop_loss=-1.0, seed=7, flags=int(os.environ.get("WS_FLAGS", "0")), items_per_quad=ipq)
        best = None
        for rep in range(4):
            ctx.reset_stats()
            ctx.minimize_stage(seq, prm, want_items=False)
            st = ctx.stats()
            if rep and (best is None or st["kernel_ms"] < best["kernel_ms"]):
                best = st
        ev, wr, ms = best["evals"][k], best["wave_rounds"][k], best["kernel_ms"]
        waves = N // (16 * ipq)
        print(f"{gname} k={k} {label:26s}: {ms:7.2f} ms, {ev/ms/1e6:6.3f} G evals/s = {100*ev*f_eval(k)/(ms*1e-3)/78.6e12:5.1f} % of peak, "
              f"occupancy {ev/16/wr:.3f}, {ms*1e3/(wr/waves):.2f} us per wave-round")
