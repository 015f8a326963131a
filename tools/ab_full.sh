#!/bin/bash
# GPU box: the in-tree build against a previous one (tools/ab_build.sh base <rev> -> lib/ab/base.so), everything a kernel change has to
# show in one call: bit-equality / parity tests, the 65 536 x 32 kernels alone (kbench), dynamic instructions per round (PMC), the driver's
# command, secondary.v2.  (Round 5: the start-point rings -- profiles/r5_ring_ab.txt.)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ab_full; mkdir -p $OUT
timeout -k 10 1000 python3 -m pytest tests/test_gpu_minimize_parity.py tests/test_gpu_round2.py tests/test_gpu_round4.py tests/test_gpu_edge_cases.py tests/test_gpu_api.py tests/test_gpu_v2.py -x -q > $OUT/pytest.txt 2>&1; rc=$?
tail -4 $OUT/pytest.txt
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" $OUT/pytest.txt | tail -20; exit 1; }
bash tools/ab_kbench1.sh "base cur" 6 | tee $OUT/kbench.txt || exit 1
for v in base cur; do
  if [ "$v" = cur ]; then unset SLAM_HIP_LIB; else export SLAM_HIP_LIB=$PWD/slam_decomposition_amd/lib/ab/$v.so; fi
  echo "== PMC $v"; bash tools/valu_per_round.sh sqiswap "1" | tee -a $OUT/vpr.txt || exit 1
done
unset SLAM_HIP_LIB
bash tools/ab_bench.sh "base cur" --steps 20 --warmup 5 --no-secondary --per-span-steps 3 | tee $OUT/bench_ab.txt
for rep in 1 2; do for v in base cur; do
  if [ "$v" = cur ]; then unset SLAM_HIP_LIB; else export SLAM_HIP_LIB=$PWD/slam_decomposition_amd/lib/ab/$v.so; fi
  python3 bench.py --v2-only > /tmp/v2.json 2>/tmp/v2.err || { tail -3 /tmp/v2.err; exit 1; }
  python3 -c "import json; d=json.loads(open('/tmp/v2.json').read().strip().splitlines()[-1]); print('v2 $v rep $rep', round(d['value']), round(d['ms_per_step'],4), round(d['roofline_frac'],4))" | tee -a $OUT/v2_ab.txt
done; done
