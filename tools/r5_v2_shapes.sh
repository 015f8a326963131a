#!/bin/bash
# GPU box: secondary.v2 over steps x steps per call x calls in flight (SLAM_V2_STEPS / _GROUP / _STREAMS)
cd "$GRAFT_REPO_ROOT"
for cfg in "512 32 8" "1024 32 8" "1024 64 8" "1024 64 6" "1024 128 4" "1024 32 12" "2048 64 8" "512 32 8"; do
  set -- $cfg
  SLAM_V2_STEPS=$1 SLAM_V2_GROUP=$2 SLAM_V2_STREAMS=$3 python3 bench.py --v2-only 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps $1 per call $2 in flight $3: %.4g /s frac %.3f ms/step %.4f' % (d['value'], d['roofline_frac'], d['ms_per_step']))"
done
