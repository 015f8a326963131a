#!/bin/bash
# GPU box: everything profiles/r2_* holds (run through gpurun; copy gpurun_out/r2_final + gpurun_out/prof_r2 afterwards)
cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash tools/final_runs.sh 2>&1 | tail -14
bash tools/profile_r2.sh r2 cfg3 > gpurun_out/profile_r2.log 2>&1; tail -3 gpurun_out/profile_r2.log
bash tools/valu_per_round.sh "sqiswap cx" "1 2 3" > gpurun_out/r2_final/valu_per_round.txt 2>&1
for k in 1 2 3; do bash tools/pmc_steady.sh sqiswap $k 8 > gpurun_out/r2_final/steady_k$k.txt 2>&1; done
build/ubench_valu > gpurun_out/r2_final/ubench_valu.txt 2>&1
python3 tools/kbench.py sqiswap 65536 32 4 > gpurun_out/r2_final/kbench.txt 2>&1; python3 tools/kbench.py cx 65536 32 4 >> gpurun_out/r2_final/kbench.txt 2>&1; python3 tools/kbench.py cx 1024 16 6 >> gpurun_out/r2_final/kbench.txt 2>&1
cat gpurun_out/r2_final/valu_per_round.txt gpurun_out/r2_final/kbench.txt
