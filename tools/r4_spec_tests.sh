cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5 || exit 1
SLAM_SPECULATE=0 timeout -k 10 600 python3 -m pytest tests/test_gpu_round4.py tests/test_gpu_round2.py tests/test_gpu_api.py -x -q 2>&1 | tail -3
