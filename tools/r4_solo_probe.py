"""Dev tool (GPU): why do the solo per-span kernel times of bench.py's single-stream pass differ from the rate the same kernels
sustain with several batches in flight?  Runs the cfg3 span loop on ONE context / stream, several repetitions per mode, and prints
every repetition's HIP-event kernel time per span:
  back2back   decompose(fetch=False) in a tight loop (host gap = one event wake-up + ~8 launches)
  fetch       decompose_range(..., fetch=True): 13 MB of results into pageable memory between two steps (bench.py's one_step)
  sleep5      back2back with a 5 ms host sleep between steps (an idle chip: does the clock ramp show in the next launch?)
usage: tools/r4_solo_probe.py [gate] [N] [R] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from slam_decomposition_amd import _ffi
from bench import gate_table, f_eval, TARGET_SEED0, OPT_SEED

gname = sys.argv[1] if len(sys.argv) > 1 else "sqiswap"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
R = int(sys.argv[3]) if len(sys.argv) > 3 else 32
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
steps = 4  # resident batches, visited round-robin
ctx = _ffi.Context(0)
table = gate_table(gname)
ctx.set_gates(table)
ctx.sample_haar(TARGET_SEED0, steps * N)
seqs = [[i % len(table) for i in range(k)] for k in (1, 2, 3)]
prm = _ffi.OptParams(restarts=R, seed=OPT_SEED, flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED)


def one(mode, s):
    if mode == "fetch":
        ctx.decompose_range(s * N, N, 1, 3, seqs, prm, 1e-10)
    else:
        ctx.decompose_range(s * N, N, 1, 3, seqs, prm, 1e-10, fetch=False)
        if mode == "sleep5":
            time.sleep(0.005)


for mode in ("back2back", "fetch", "sleep5", "back2back"):
    for s in range(2):
        one(mode, s)  # warm
    rows = []
    t0 = time.perf_counter()
    for rep in range(reps):
        ctx.reset_stats()
        one(mode, rep % steps)
        st = ctx.stats()
        rows.append(st)
    wall = (time.perf_counter() - t0) / reps * 1e3
    print(f"== {mode}: wall {wall:.2f} ms/step")
    for k in (1, 2, 3):
        ms = [r["kernel_ms_span"][k] for r in rows]
        ev = rows[0]["evals"][k]
        fr = [r["evals"][k] * f_eval(k) / (r["kernel_ms_span"][k] * 1e-3) / 78.6e12 for r in rows]
        print(f"   k={k}: ms {[round(m, 3) for m in ms]}  frac {[round(f, 3) for f in fr]}  evals(rep0) {ev}")
    print(f"   total_ms (ev_t0..ev_t1) {[round(r['total_ms'], 3) for r in rows]}")
