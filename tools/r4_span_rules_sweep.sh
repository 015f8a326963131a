cd "$GRAFT_REPO_ROOT"
for wl in cfg3 cfg4; do for st in 5 8 12; do
  python3 bench.py --workload $wl --span-rules --no-cpu-baseline --no-secondary --repeats 2 --streams $st | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl span-rules streams $st:', '%.4g dec/s' % d['value'], '%.3f ms/step' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'])"
done; done
