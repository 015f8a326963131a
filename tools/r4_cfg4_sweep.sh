cd "$GRAFT_REPO_ROOT"
C="--workload cfg4 --no-cpu-baseline --no-secondary --per-span-steps 0 --repeats 2"
for cfg in "1 5" "1 8" "2 4" "2 6" "4 3"; do set -- $cfg
  python3 bench.py $C --group $1 --streams $2 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('group $1 streams $2:', '%.4g dec/s' % d['value'], '%.3f ms/step' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'])"
done
