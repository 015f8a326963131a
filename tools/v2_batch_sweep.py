import sys, json
sys.path.insert(0, '.')
import bench
for (steps, warmup, nt, ns) in [(12,4,4096,4),(48,8,4096,4),(12,2,32768,4),(6,1,65536,2),(4,1,65536,1),(4,1,262144,1)]:
    r = bench.run_v2(0, 0, steps=steps, warmup=warmup, n_targets=nt, n_streams=ns)
    print(steps, warmup, nt, ns, '%.3g dec/s' % r['value'], 'frac %.3f' % r['roofline_frac'], 'ms/step %.3f' % r['ms_per_step'], {k: round(v,3) for k,v in r['kernel_ms_per_step_alone_or_overlapped'].items()}, r['evals_per_span'], flush=True)

# the general kernels (conversion-gain gates with two free drive strengths per gate: GQ = 0 instantiations)
from slam_decomposition_amd.gates import ConversionGainGate
r = bench.run_v2(0, 0, base_gate=lambda gc, gg: ConversionGainGate(0.0, 0.0, gc, gg, 1.0), gate_desc="lambda gc, gg: ConversionGainGate(0, 0, gc, gg, 1)")
print('cg_gc_gg default grouping', '%.3g dec/s' % r['value'], 'frac %.3f' % r['roofline_frac'], 'ms/step %.3f' % r['ms_per_step'], {k: round(v,3) for k,v in r['kernel_ms_per_step_alone_or_overlapped'].items()}, r['evals_per_span'], r['best_cycles_hist'], flush=True)
