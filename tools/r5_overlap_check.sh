cd $GRAFT_REPO_ROOT
show() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1: %.4g /s  %.4f ms/step [%.4f..%.4f] frac %.3f solved %.4f per_span %s' % (d['value'], d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], r['frac'], d['solved_fraction'], {k:(round(v['hip_event_ms'],3),round(v['frac'],3)) for k,v in (r.get('per_span') or {}).items() if k!='all'}))"; }
python3 bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | show "cfg2-20 auto"
python3 bench.py --workload cfg2 --steps 20 --warmup 5 --staged --no-cpu-baseline --no-secondary 2>/dev/null | show "cfg2-20 staged"
python3 bench.py --streams 1 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | show "cfg3 1 stream auto"
python3 bench.py --workload cfg4 --streams 1 --group 1 --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --per-span-steps 0 2>/dev/null | show "cfg4 1 stream auto (B: no overlap expected)"
python3 bench.py --workload cfg2 --streams 1 --group 1 --steps 40 --warmup 8 --no-cpu-baseline --no-secondary --per-span-steps 0 2>/dev/null | show "cfg2 one batch per call"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --per-span-steps 0 2>/dev/null | show "default"
