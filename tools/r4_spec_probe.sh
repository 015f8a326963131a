# speculative spans (default for <= 2 targets per CU) against the one-wavefront-per-target loop (SLAM_SPECULATE=0) and the per-span launches
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_round4.py tests/test_gpu_round2.py tests/test_gpu_api.py -x -q 2>&1 | tail -5 || exit 1
echo "== approximate_target_U, speculative spans"; timeout -k 10 300 python3 tools/r4_single_target_latency.py || exit 1
echo "== approximate_target_U, SLAM_SPECULATE=0"; SLAM_SPECULATE=0 timeout -k 10 300 python3 tools/r4_single_target_latency.py || exit 1
for g in cx sqiswap; do
  echo "== $g, speculative spans"; timeout -k 10 300 python3 tools/r4_wave_probe.py $g 16 || exit 1
  echo "== $g, SLAM_SPECULATE=0"; SLAM_SPECULATE=0 timeout -k 10 300 python3 tools/r4_wave_probe.py $g 16 || exit 1
done
