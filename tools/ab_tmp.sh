python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do
for v in prev cur; do
  if [ "$v" = cur ]; then unset SLAM_HIP_LIB; else export SLAM_HIP_LIB=$PWD/slam_decomposition_amd/lib/ab/$v.so; fi
  echo "== $v (rep $rep)"
  python tools/kbench.py sqiswap 65536 32 4 | cut -c1-260 || exit 1
  python tools/kbench.py cx 65536 32 4 | cut -c1-260 || exit 1
done; done
unset SLAM_HIP_LIB
bash tools/valu_per_round.sh "sqiswap" "1 2 3"
