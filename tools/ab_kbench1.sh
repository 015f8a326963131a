#!/bin/bash
# GPU box: A/B of lib variants on ONE kbench workload (sqrt(iSWAP) 65536 x 32).  usage: tools/ab_kbench1.sh "<variants>" [reps]
for rep in 1 2 3; do
for v in $1; do
  if [ "$v" = cur ]; then unset SLAM_HIP_LIB; else export SLAM_HIP_LIB=$PWD/slam_decomposition_amd/lib/ab/$v.so; fi
  echo "== $v (rep $rep) $(python tools/kbench.py sqiswap 65536 32 ${2:-6} | cut -c22-150)" || exit 1
done; done
