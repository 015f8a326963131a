#!/bin/bash
# GPU box: A/B of lib variants on kbench workloads.  usage: tools/ab_kbench.sh "<variant names; 'cur' = in-tree build>" [reps]
for rep in 1 2 3; do
for v in $1; do
  if [ "$v" = cur ]; then unset SLAM_HIP_LIB; else export SLAM_HIP_LIB=$PWD/slam_decomposition_amd/lib/ab/$v.so; fi
  echo "== $v (rep $rep)"
  python tools/kbench.py sqiswap 65536 32 ${2:-6} | cut -c1-120 || exit 1
  python tools/kbench.py cx 65536 16 ${2:-6} | cut -c1-120 || exit 1
done; done
