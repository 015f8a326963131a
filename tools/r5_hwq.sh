#!/bin/bash
# GPU box: does the number of hardware queues the HIP runtime maps streams onto (GPU_MAX_HW_QUEUES, default 4) matter?
# usage: tools/r5_hwq.sh "<values, 0 = unset>" "<workloads>"
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_hwq; mkdir -p $O
show() { python3 -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; print('$2', '%.4g /s  %.3f ms/step [%.3f..%.3f] frac %.3f' % (d['value'], d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], r['frac']))"; }
for q in ${1:-0 4 8 16 32}; do
  if [ "$q" = "0" ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  for w in ${2:-cfg3 cfg2 cfg5 cfg4}; do
    python3 bench.py --workload $w --no-cpu-baseline --no-secondary --per-span-steps 0 > $O/${w}_$q.json 2>> $O/err.txt; show $O/${w}_$q.json "hwq $q $w"
  done
  python3 bench.py --v2-only 2>> $O/err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hwq $q v2 %.4g /s frac %.3f' % (d['value'], d['roofline_frac']))"
done
