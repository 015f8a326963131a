for rep in 1 2; do
for v in prev cur rb1 rb2; do
  if [ "$v" = cur ]; then unset SLAM_HIP_LIB; else export SLAM_HIP_LIB=$PWD/slam_decomposition_amd/lib/ab/$v.so; fi
  echo "== $v (rep $rep)"
  python tools/kbench.py sqiswap 65536 32 4 | cut -c1-200 || exit 1
  python tools/kbench.py cx 65536 32 4 | cut -c1-200 || exit 1
done; done
