"""Dev tool (GPU): the wavefront-per-item kernels (templates of 6..16 gates) -- steady-state time per evaluation (uniform items) and
the distribution of iterations / statuses of a real stage (where the stage's time goes: mean work or stragglers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slam_decomposition_amd import _ffi
from slam_decomposition_amd.gates import ConversionGainGate

ctx = _ffi.Context(0)
ctx.set_gates(np.stack([ConversionGainGate(0.0, 0.0, 0.0, np.pi / 16, 1.0).to_matrix()]))
for k in (6, 8, 12, 16):
    N = 2048 * 8
    ctx.sample_haar(20260555, N)
    prm = _ffi.OptParams(restarts=1, maxiter=30, gtol=0.0, gtol_far=0.0, stop_loss=-1.0, seed=7, flags=0)
    for rep in range(2):
        ctx.reset_stats()
        t0 = time.perf_counter()
        ctx.minimize_stage([0] * k, prm, want_items=False)
        dt = time.perf_counter() - t0
    st = ctx.stats()
    ev = st["evals"][k]
    print(f"steady k={k}: {1e3 * dt:.2f} ms, {ev} evals, {ev / dt / 1e6:.1f} M evals/s, {2048 * dt / ev * 1e6:.2f} us per evaluation per wavefront (2048 resident), "
          f"frac {ev * (3036 * k + 1247) / dt / 78.6e12:.4f}", flush=True)
k, N, R = 8, 4096, 8
ctx.sample_haar(20260555, N)
prm = _ffi.OptParams(restarts=R, maxiter=2500, gtol=1e-9, stop_loss=1e-13, seed=20261003, flags=_ffi.FLAG_EARLY_EXIT | _ffi.FLAG_ORDERED)
t0 = time.perf_counter()
out = ctx.minimize_stage([0] * k, prm)
dt = time.perf_counter() - t0
it, ev, stt = out["item_iters"].ravel(), out["item_evals"].ravel(), out["item_status"].ravel()
ran = ev > 0
print(f"real k={k}: {1e3 * dt:.1f} ms; items run {ran.sum()} of {ran.size}; evals mean {ev[ran].mean():.1f} median {np.median(ev[ran]):.0f} p99 {np.percentile(ev[ran], 99):.0f} max {ev.max()}; "
      f"status counts {np.bincount(stt, minlength=6)}; solved targets {(out['best_loss'] < 1e-8).mean():.4f}")
for s in range(6):
    m = ran & (stt == s)
    if m.any():
        print(f"   status {s}: n {m.sum()}, evals mean {ev[m].mean():.1f} max {ev[m].max()}, loss median {np.median(out['item_loss'].ravel()[m]):.3g}")
