#!/bin/bash
# GPU box: BASELINE configs[4] shard (16 bases x 4096 x 16) through slam_decompose_multi against one call per basis (round 3).
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_cfg5; mkdir -p $OUT
C="--workload cfg5 --steps 160 --warmup 16 --no-cpu-baseline --no-secondary"
for rep in 1 2; do
  python3 bench.py $C > $OUT/multi_$rep.json 2>> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
  python3 bench.py $C --no-multi > $OUT/single_$rep.json 2>> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
done
python3 bench.py $C --streams 1 > $OUT/multi_1inflight.json 2>> $OUT/err.txt
python3 bench.py $C --streams 3 > $OUT/multi_3inflight.json 2>> $OUT/err.txt
for f in $OUT/*.json; do python3 -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
print('$f'.split('/')[-1], '%.4g dec/s' % d['value'], '%.3f ms/step' % d['ms_per_step'], 'frac %.3f' % r['frac'], 'solved %.4f' % d['solved_fraction'], 'per_span', {k:(round(v['hip_event_ms'],3), round(v['frac'],3)) for k,v in r.get('per_span',{}).items() if k!='all'})
"; done
