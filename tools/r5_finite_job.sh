#!/bin/bash
# GPU box: the finite job (5 windows of 65 536) through the bench's own loop and through the API, on one box.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_finite; mkdir -p $O
for cfg in "5 4" "5 5" "10 4" "10 5" "20 5"; do
  set -- $cfg
  python3 bench.py --steps $1 --warmup 5 --streams $2 --no-cpu-baseline --no-secondary --per-span-steps 0 > $O/b_$1_$2.json 2>> $O/err.txt
  python3 -c "
import json; d=json.loads(open('$O/b_$1_$2.json').read().strip().splitlines()[-1]); print('bench steps $1 streams $2: %.3f ms/step [%.3f..%.3f] -> %.2f ms per 5 windows' % (d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], 5*d['ms_per_step']))"
done
python3 tools/r5_api_large_probe.py
python3 tools/r5_api_large_probe.py 655360
