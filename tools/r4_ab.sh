#!/bin/bash
# GPU box: interleaved A/B of library variants (slam_decomposition_amd/lib/ab/<name>.so; cur = the in-tree build): the 65 536 x 32
# sqrt(iSWAP) kernels alone (kbench) and the driver's command.  usage: tools/r4_ab.sh "<variants>"
cd "$GRAFT_REPO_ROOT"
bash tools/ab_kbench1.sh "$1" 6
bash tools/ab_bench.sh "$1" --steps 20 --warmup 5 --no-secondary --per-span-steps 3
