cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python3 -m pytest tests/test_gpu_round4.py tests/test_gpu_api.py tests/test_gpu_round3.py -x -q 2>&1 | tail -8 || exit 1
for a in "--workload cfg5 --steps 160 --warmup 16 --span-rules" "--workload cfg4 --span-rules" "--span-rules"; do
  python3 bench.py $a --no-cpu-baseline --no-secondary | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a:', '%.4g dec/s' % d['value'], '%.3f ms/step' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'], 'solved', d['solved_fraction'])"
done
