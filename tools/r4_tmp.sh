cd "$GRAFT_REPO_ROOT"
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
SLAM_STAGED=1 timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -k "not wave and not overlapped and not speculative" 2>&1 | tail -3
