cd "$GRAFT_REPO_ROOT"
python3 bench.py --api-only 2>&1 | tail -3 | cut -c1-1500
