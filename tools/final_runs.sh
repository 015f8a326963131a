#!/bin/bash
# GPU box: the unprofiled bench lines of a round (profiles/<tag>_bench_*.json).  usage: tools/final_runs.sh [tag]   (default r5)
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r5}
D=gpurun_out/${TAG}_final; mkdir -p $D
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench_default.json 2> $D/bench_default.err
python3 bench.py --workload cfg2 --no-cpu-baseline > $D/bench_cfg2.json 2>> $D/err.txt
python3 bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline > $D/bench_cfg2_20.json 2>> $D/err.txt
python3 bench.py --workload cfg2 --streams 1 --group 1 --steps 40 --warmup 8 --no-cpu-baseline > $D/bench_cfg2_1stream.json 2>> $D/err.txt
python3 bench.py --workload cfg4 --no-cpu-baseline > $D/bench_cfg4.json 2>> $D/err.txt
python3 bench.py --workload cfg4 --span-rules --no-cpu-baseline > $D/bench_cfg4_span_rules.json 2>> $D/err.txt
python3 bench.py --workload cfg5 --steps 160 --warmup 16 --no-cpu-baseline > $D/bench_cfg5.json 2>> $D/err.txt
python3 bench.py --workload cfg5 --steps 160 --warmup 16 --no-multi --no-cpu-baseline > $D/bench_cfg5_no_multi.json 2>> $D/err.txt
python3 bench.py --workload cfg5 --steps 160 --warmup 16 --span-rules --no-cpu-baseline > $D/bench_cfg5_span_rules.json 2>> $D/err.txt
python3 bench.py --span-rules --no-cpu-baseline --no-secondary > $D/bench_cfg3_span_rules.json 2>> $D/err.txt
python3 bench.py --fast-exit --no-cpu-baseline --no-secondary > $D/bench_cfg3_fast_exit.json 2>> $D/err.txt
python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-secondary > $D/bench_cfg3_100steps.json 2>> $D/err.txt
SLAM_BENCH_COMM=file timeout -k 10 300 python3 bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > $D/bench_2ranks_one_gpu_filecomm.json 2>> $D/err.txt
SLAM_BENCH_COMM=file timeout -k 10 300 python3 bench.py --gpus 2 --scaling strong --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > $D/bench_2ranks_one_gpu_filecomm_strong.json 2>> $D/err.txt
SLAM_BENCH_FORCE_LAUNCH=1 timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $D/bench_launcher_rccl_world1.json 2>> $D/err.txt
python3 tools/kbench.py sqiswap 65536 32 4 > $D/kbench.txt 2>&1; python3 tools/kbench.py cx 65536 32 4 >> $D/kbench.txt 2>&1; python3 tools/kbench.py cx 1024 16 6 >> $D/kbench.txt 2>&1
python3 tools/r4_wave_probe.py cx 16 > $D/wave_probe.txt 2>&1; python3 tools/r4_wave_probe.py sqiswap 5 >> $D/wave_probe.txt 2>&1
python3 tools/r4_mq_probe.py 64 > $D/mq_probe.txt 2>&1
python3 bench.py --workload cfg4 > $D/bench_cfg4_with_parity.json 2>> $D/err.txt
python3 bench.py --workload cfg5 --steps 160 --warmup 16 > $D/bench_cfg5_with_parity.json 2>> $D/err.txt
python3 bench.py --long-only > $D/long.txt 2>> $D/err.txt
python3 tools/r5_long_probe.py > $D/long_probe.txt 2>&1
python3 tools/r5_api_large_probe.py > $D/api_large_probe.txt 2>&1
for f in $D/*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
print('$f'.split('/')[-1], '%.4g dec/s' % d['value'], '%.3f ms/step' % d['ms_per_step'], '[%.3f .. %.3f]' % (d['ms_per_step_min'], d['ms_per_step_max']), 'frac %.3f' % r['frac'], 'acc %.3f' % r['frac_accepted'], 'kern', r.get('frac_kernel'), 'solved', d['solved_fraction'])
"; done
tail -3 $D/err.txt; cat $D/kbench.txt $D/wave_probe.txt $D/mq_probe.txt $D/long.txt $D/long_probe.txt $D/api_large_probe.txt
