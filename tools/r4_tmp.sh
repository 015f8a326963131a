cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python3 -m pytest tests/test_gpu_round4.py tests/test_gpu_api.py tests/test_gpu_round2.py -x -q 2>&1 | tail -3 || exit 1
python3 bench.py --api-only 2>/dev/null | tail -1 | cut -c1-700
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d['secondary']
print('default %.4g dec/s %.3f ms frac %.3f kern %.3f' % (d['value'], d['ms_per_step'], r['frac'], r['frac_kernel']))
print('cfg2', s['cfg2']['value'], s['cfg2']['roofline_frac'], 'one batch', s['cfg2']['one_batch_per_call'])
print('api', s['api']['value'], s['api']['wall_ms'], s['api']['approximate_target_U_ms']['median'])"
python3 bench.py --workload cfg5 --steps 160 --warmup 16 --no-multi --no-cpu-baseline --no-secondary | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 no-multi %.4g dec/s frac %.3f' % (d['value'], d['roofline']['frac']))"
