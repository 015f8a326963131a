#!/bin/bash
# usage: [AB_FLAGS="-DX=1"] tools/ab_build.sh <name> [git-rev]   -- build libslamhip variant into slam_decomposition_amd/lib/ab/<name>.so
set -e
NAME=$1; REV=$2
mkdir -p slam_decomposition_amd/lib/ab
if [ -n "$REV" ]; then
  rm -rf /tmp/ab_src && mkdir -p /tmp/ab_src && git archive $REV slam_decomposition_amd/csrc include | tar -x -C /tmp/ab_src
  SRC=/tmp/ab_src
else
  SRC=.
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $AB_FLAGS -o slam_decomposition_amd/lib/ab/$NAME.so $SRC/slam_decomposition_amd/csrc/slam_hip.hip $SRC/slam_decomposition_amd/csrc/slam_comm.hip -ldl
echo built $NAME
