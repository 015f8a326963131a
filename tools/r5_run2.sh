#!/bin/bash
# GPU box: measured agreement counts of the SciPy parity test, the api_large shapes, parity_sample on the cfg4 / cfg5 shards.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_run2; mkdir -p $OUT
timeout -k 10 300 python3 -m pytest tests/test_gpu_minimize_parity.py -q -s -k "converged_loss_matches_scipy" > $OUT/agree.txt 2>&1 || { tail -20 $OUT/agree.txt; exit 1; }
grep AGREE $OUT/agree.txt
timeout -k 10 400 python3 tools/r5_api_large_probe.py > $OUT/api_large_probe.txt 2>&1 || { tail -20 $OUT/api_large_probe.txt; exit 1; }
cat $OUT/api_large_probe.txt
for w in cfg4 cfg5; do
  timeout -k 10 400 python3 bench.py --workload $w --no-secondary > $OUT/bench_$w.json 2> $OUT/bench_$w.err || { tail -20 $OUT/bench_$w.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$OUT/bench_$w.json').read().strip().splitlines()[-1])
print('$w', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), {k:v for k,v in d['parity_sample'].items() if k!='what'})"
done
