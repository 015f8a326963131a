cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6 || exit 1
SLAM_OVERLAP=0 timeout -k 10 600 python3 -m pytest tests/test_gpu_round4.py tests/test_gpu_round2.py tests/test_gpu_api.py -x -q -k "not overlapped" 2>&1 | tail -3
