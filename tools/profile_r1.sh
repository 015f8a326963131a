#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace summaries and PMC passes for bench.py.
# usage: tools/profile_r1.sh <tag> <workload> <steps> <warmup>
set -o pipefail
TAG=$1; WL=${2:-cfg2}; STEPS=${3:-10}; WARM=${4:-2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload $WL --steps $STEPS --warmup $WARM --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/trace.err || exit 1
cp $OUT/trace/*/*_kernel_stats.csv $OUT/kernel_stats.csv
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc1 -- python3 bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_pmc1.json 2> $OUT/pmc1.err || exit 1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- python3 bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_pmc2.json 2> $OUT/pmc2.err || exit 1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $OUT/pmc3 -- python3 bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_pmc3.json 2> $OUT/pmc3.err || exit 1
python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt
cat $OUT/kernel_stats.csv | head -6; cat $OUT/pmc_summary.txt
