cd "$GRAFT_REPO_ROOT"
for cfg in "128 16 4" "256 16 4" "256 16 8" "256 8 8" "512 16 8" "512 32 8" "512 8 16"; do
  set -- $cfg
  SLAM_V2_STEPS=$1 SLAM_V2_GROUP=$2 SLAM_V2_STREAMS=$3 python3 bench.py --v2-only 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('v2 steps $1 group $2 in flight $3:', '%.4g dec/s' % d['value'], 'frac %.3f' % d['roofline_frac'], 'ms/step %.3f' % d['ms_per_step'])"
done
