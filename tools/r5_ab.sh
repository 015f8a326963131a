#!/bin/bash
# GPU box (round 5): correctness of a library variant on the optimizer parity tests, then interleaved A/B of variants on the
# 65 536 x 32 sqrt(iSWAP) kernels (kbench) and dynamic instructions per round.
# usage: tools/r5_ab.sh "<variants; cur = in-tree>" [variant to test] [variants for the PMC pass]
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_ab; mkdir -p $OUT
if [ -n "$2" ]; then
  export SLAM_HIP_LIB=$PWD/slam_decomposition_amd/lib/ab/$2.so
  timeout -k 10 600 python3 -m pytest tests/test_gpu_minimize_parity.py tests/test_gpu_eval_parity.py tests/test_gpu_round2.py -x -q > $OUT/pytest_$2.txt 2>&1 || { tail -30 $OUT/pytest_$2.txt; exit 1; }
  tail -2 $OUT/pytest_$2.txt
  unset SLAM_HIP_LIB
fi
bash tools/ab_kbench1.sh "$1" 6 | tee $OUT/kbench.txt || exit 1
for v in ${3:-}; do
  if [ "$v" = cur ]; then unset SLAM_HIP_LIB; else export SLAM_HIP_LIB=$PWD/slam_decomposition_amd/lib/ab/$v.so; fi
  echo "== PMC $v"; bash tools/valu_per_round.sh sqiswap "1" | tee -a $OUT/vpr.txt || exit 1
done
