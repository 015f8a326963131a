"""Dev tool: wall-clock timeline of overlapped bench steps (per thread)."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_decomposition_amd import _ffi
from bench import gate_table, make_targets
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
per = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = 1024
ctxs = [_ffi.Context(0) for _ in range(S)]
table = gate_table("cx")
targets = make_targets(n * S * (per + 1), 20260000)
for c in ctxs:
    c.set_gates(table); c.set_targets(targets)
seqs = [[0], [0, 0], [0, 0, 0]]
prm = _ffi.OptParams(restarts=16, seed=1, flags=1)
def step(c, s):
    return c.decompose_range(s * n, n, 1, 3, seqs, prm, 1e-10)
for w, c in enumerate(ctxs):
    step(c, w)
log = []
t0 = time.perf_counter()
def worker(w):
    for j in range(per):
        a = time.perf_counter() - t0
        step(ctxs[w], S + j * S + w)
        b = time.perf_counter() - t0
        log.append((w, j, a * 1e3, b * 1e3, ctxs[w].stats()["total_ms"]))
th = [threading.Thread(target=worker, args=(w,)) for w in range(S)]
[t.start() for t in th]; [t.join() for t in th]
tot = (time.perf_counter() - t0) * 1e3
for r in sorted(log): print("thread %d step %d: start %.2f end %.2f (%.2f ms; device span %.2f ms)" % (r[0], r[1], r[2], r[3], r[3]-r[2], r[4]))
print("total %.2f ms for %d steps -> %.3f ms/step" % (tot, S * per, tot / (S * per)))
