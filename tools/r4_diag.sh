#!/bin/bash
# GPU box (round 4): where does the gap between the solo per-span kernel fractions (0.38 / 0.43 / 0.14) and the
# wall-clock fraction of the driver's command (0.455, five batches in flight) come from?
#   1. tools/r4_solo_probe.py: solo kernel times per repetition, back to back / with the result fetch / with idle gaps
#   2. clocks + power (rocm-smi) under ONE batch in flight and under FIVE
#   3. rocprofv3 kernel trace of the driver's command: union of kernel intervals, per-kernel durations under concurrency
# usage: tools/r4_diag.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r4_diag; rm -rf $OUT; mkdir -p $OUT
python3 tools/r4_solo_probe.py sqiswap 65536 32 6 > $OUT/solo_probe.txt 2>&1 || { tail -5 $OUT/solo_probe.txt; exit 1; }
cat $OUT/solo_probe.txt
COMMON="--no-cpu-baseline --no-secondary"
probe() {  # $1 = tag, rest = bench args: clocks / power sampled while the bench runs
  python3 bench.py $COMMON --steps 150 --warmup 3 --repeats 1 --per-span-steps 0 "${@:2}" > $OUT/pp_$1.json 2> $OUT/pp_$1.err &
  BP=$!
  sleep 6
  for i in 1 2 3 4 5; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -3 | tr '\n' ' '
    echo
    sleep 0.5
  done > $OUT/smi_$1.txt
  wait $BP
  echo "== $1"; cat $OUT/smi_$1.txt
  python3 -c "
import json; d=json.loads(open('$OUT/pp_$1.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$1', '%.4g dec/s' % d['value'], '%.3f ms/step' % d['ms_per_step'], 'frac %.3f' % r['frac'], 'kernel_ms/step', {k: round(v/d['steps'],3) for k,v in (r.get('kernel_ms_span') or {}).items()})"
}
probe s1 --streams 1
probe s2 --streams 2
probe s5 --streams 5
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_headline -- python3 bench.py --steps 20 --warmup 5 $COMMON --per-span-steps 0 > $OUT/headline_bench_under_trace.json 2> $OUT/trace_headline.err || { tail -5 $OUT/trace_headline.err; exit 1; }
KT=$(ls $OUT/trace_headline/*/*_kernel_trace.csv | head -1)
python3 tools/trace_concurrency.py $KT > $OUT/headline_concurrency.txt
cat $OUT/headline_concurrency.txt
cp $OUT/trace_headline/*/*_kernel_stats.csv $OUT/headline_kernel_stats.csv
rm -rf $OUT/trace_headline
