cd "$GRAFT_REPO_ROOT"
for cfg in "64 8 4" "128 16 4" "128 8 8" "128 32 2" "256 32 4"; do set -- $cfg
  SLAM_V2_STEPS=$1 SLAM_V2_GROUP=$2 SLAM_V2_STREAMS=$3 python3 bench.py --v2-only | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps $1 group $2 streams $3:', '%.4g dec/s' % d['value'], 'frac %.3f' % d['roofline_frac'], d['evals_per_span'])"
done
