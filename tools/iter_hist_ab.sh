#!/bin/bash
# GPU box: per-item evaluation-count distribution (tools/iter_hist.py) for library variants.  usage: tools/iter_hist_ab.sh "<variants>" <gate> <N> <R>
for v in $1; do
  if [ "$v" = cur ]; then unset SLAM_HIP_LIB; else export SLAM_HIP_LIB=$PWD/slam_decomposition_amd/lib/ab/$v.so; fi
  echo "== $v"; python tools/iter_hist.py $2 $3 $4 || exit 1
done
