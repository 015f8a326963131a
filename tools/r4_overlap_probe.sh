cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_round4.py -x -q -k "overlapped or wave" 2>&1 | tail -8 || exit 1
for a in "cx 1024 16" "cx 2048 16" "cx 4096 16" "cx 20480 16" "sqiswap 4096 32" "sqiswap 65536 32"; do timeout -k 10 300 python3 tools/r4_overlap_probe.py $a || exit 1; done
