cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_round4.py -x -q -k "randomized" 2>&1 | tail -12
