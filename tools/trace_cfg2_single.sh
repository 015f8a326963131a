cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_cfg2s; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 bench.py --workload cfg2 --streams 1 --steps 40 --warmup 8 --no-cpu-baseline --no-secondary --per-span-steps 0 > $OUT/bench.json 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
cp $OUT/t/*/*_kernel_stats.csv $OUT/kernel_stats.csv
cp $OUT/t/*/*_kernel_trace.csv $OUT/kernel_trace.csv
rm -rf $OUT/t
cat $OUT/kernel_stats.csv | cut -c1-160
