cd "$GRAFT_REPO_ROOT"
for a in "" "--group 5" "--group 4 --streams 5" "--group 2 --streams 8" "--group 10"; do
  python3 bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline $a | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 --steps 20 $a:', '%.4g dec/s' % d['value'], '%.3f ms/step' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'], d['config']['batches_in_flight_per_gpu'], d['config']['steps_per_library_call'])"
done
