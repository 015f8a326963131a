#!/bin/bash
# GPU box: kernel trace of the default bench (several batches in flight) + concurrency summary
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_cfg2; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 bench.py --no-cpu-baseline --steps ${STEPS:-100} --warmup 10 --streams ${STREAMS:-16} --items-per-quad ${IPQ:-0} > $OUT/bench.json 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
cat $OUT/bench.json | tail -1 | cut -c1-200
F=$(ls $OUT/t/*/*_kernel_trace.csv | head -1)
python3 tools/trace_concurrency.py $F | tee $OUT/summary.txt
rm -rf $OUT/t
