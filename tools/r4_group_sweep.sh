#!/bin/bash
# GPU box: headline workload (configs[2]) with several steps per library call x calls in flight
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_group; mkdir -p $OUT
C="--steps 20 --warmup 5 --no-cpu-baseline --no-secondary --per-span-steps 0 --repeats 2"
for cfg in "1 5" "2 3" "2 5" "4 2" "4 3" "5 2" "1 8"; do set -- $cfg
  python3 bench.py $C --group $1 --streams $2 > $OUT/g$1_s$2.json 2>> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
done
for f in $OUT/*.json; do python3 -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
print('$f'.split('/')[-1], '%.4g dec/s' % d['value'], '%.3f ms/step' % d['ms_per_step'], 'frac %.3f' % r['frac'])
"; done
