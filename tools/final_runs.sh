mkdir -p gpurun_out/r2_final
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2_final/bench_default.json 2> gpurun_out/r2_final/bench_default.err
python3 bench.py --workload cfg2 --no-cpu-baseline > gpurun_out/r2_final/bench_cfg2.json 2>> gpurun_out/r2_final/err.txt
python3 bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_final/bench_cfg2_20.json 2>> gpurun_out/r2_final/err.txt
python3 bench.py --workload cfg2 --streams 1 --steps 40 --warmup 8 --no-cpu-baseline > gpurun_out/r2_final/bench_cfg2_1stream.json 2>> gpurun_out/r2_final/err.txt
python3 bench.py --workload cfg4 --no-cpu-baseline > gpurun_out/r2_final/bench_cfg4.json 2>> gpurun_out/r2_final/err.txt
python3 bench.py --workload cfg5 --steps 320 --warmup 32 --no-cpu-baseline > gpurun_out/r2_final/bench_cfg5.json 2>> gpurun_out/r2_final/err.txt
python3 bench.py --span-rules --no-cpu-baseline --no-secondary > gpurun_out/r2_final/bench_cfg3_span_rules.json 2>> gpurun_out/r2_final/err.txt
python3 bench.py --fast-exit --no-cpu-baseline --no-secondary > gpurun_out/r2_final/bench_cfg3_fast_exit.json 2>> gpurun_out/r2_final/err.txt
SLAM_BENCH_COMM=file timeout -k 10 300 python3 bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r2_final/bench_2ranks_one_gpu_filecomm.json 2>> gpurun_out/r2_final/err.txt
SLAM_BENCH_FORCE_LAUNCH=1 timeout -k 10 300 python3 bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r2_final/bench_launcher_rccl_world1.json 2>> gpurun_out/r2_final/err.txt
python3 tools/wave_scaling.py sqiswap > gpurun_out/r2_final/wave_scaling.txt 2>&1
for f in gpurun_out/r2_final/*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
print('$f'.split('/')[-1], '%.4g dec/s' % d['value'], '%.3f ms/step' % d['ms_per_step'], 'frac %.3f' % r['frac'], 'acc %.3f' % r['frac_accepted'], 'solved', d['solved_fraction'])
"; done
tail -3 gpurun_out/r2_final/err.txt
