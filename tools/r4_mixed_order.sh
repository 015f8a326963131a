cd "$GRAFT_REPO_ROOT" && timeout -k 10 900 python3 -m pytest tests/test_gpu_round4.py -x -q -k "mixed_order or predictor" 2>&1 | tail -30
