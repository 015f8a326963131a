cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_round4.py -x -q -k "overlapped" 2>&1 | grep -E "Error|error|assert|FAILED|passed" | head -20
