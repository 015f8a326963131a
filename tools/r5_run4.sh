#!/bin/bash
# GPU box: slam_decompose_predicted tests, the span-rules benches (cfg3 / cfg4 / cfg5), api-only line.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_run4; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_round5.py tests/test_gpu_api.py tests/test_gpu_round4.py -x -q -k "predicted or polytope or span or unsupported or coverage or v2_template" > $OUT/t.txt 2>&1; rc=$?
tail -5 $OUT/t.txt
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" $OUT/t.txt | tail -20; exit 1; }
for w in cfg3 cfg4 cfg5; do
  timeout -k 10 300 python3 bench.py --workload $w --span-rules --no-secondary --no-cpu-baseline > $OUT/bench_${w}_span_rules.json 2> $OUT/bench_${w}_sr.err || { tail -20 $OUT/bench_${w}_sr.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$OUT/bench_${w}_span_rules.json').read().strip().splitlines()[-1])
print('$w span-rules', '%.4g' % d['value'], round(d['ms_per_step'],3), round(d['roofline']['frac'],4), d['solved_fraction'])"
done
timeout -k 10 300 python3 bench.py --api-only > $OUT/api_only.json 2> $OUT/api_only.err || { tail -20 $OUT/api_only.err; exit 1; }
cat $OUT/api_only.json
