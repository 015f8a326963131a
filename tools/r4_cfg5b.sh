#!/bin/bash
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_cfg5b; mkdir -p $OUT
C="--workload cfg5 --steps 160 --warmup 16 --no-cpu-baseline --no-secondary --per-span-steps 0 --repeats 2"
for cfg in "16 2" "8 4" "8 6" "4 8" "16 4" "2 12"; do set -- $cfg
  python3 bench.py $C --group $1 --streams $2 > $OUT/g$1_s$2.json 2>> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
done
python3 bench.py $C --no-multi > $OUT/single.json 2>> $OUT/err.txt
for f in $OUT/*.json; do python3 -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
print('$f'.split('/')[-1], '%.4g dec/s' % d['value'], '%.3f ms/step' % d['ms_per_step'], 'frac %.3f' % r['frac'])
"; done
