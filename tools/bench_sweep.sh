#!/bin/bash
# GPU box: ONE parametrised sweep over bench.py arguments (replaces the round-2..4 one-off sweep scripts: items per quad, batches in
# flight, steps per call, workloads x --span-rules, V2 shapes, strong-scaling batch sizes -- their results are in HISTORY.md).
# usage: tools/bench_sweep.sh [-e "VAR=a VAR=b ..."] "<args of run 1>" "<args of run 2>" ...
#   e.g. tools/bench_sweep.sh "--workload cfg4 --streams 8" "--workload cfg4 --streams 12" "--workload cfg5 --group 8 --streams 4"
#        tools/bench_sweep.sh -e "SLAM_V2_STEPS=512 SLAM_V2_GROUP=32" "--v2-only"
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
ENVS=""
if [ "$1" = "-e" ]; then ENVS="$2"; shift 2; fi
for args in "$@"; do
  env $ENVS python3 bench.py --no-cpu-baseline --no-secondary $args > /tmp/bs.json 2> /tmp/bs.err || { echo "FAILED: $args"; tail -3 /tmp/bs.err; exit 1; }
  python3 - "$args" <<'PY'
import json, sys
d = json.loads(open("/tmp/bs.json").read().strip().splitlines()[-1])
if "roofline" in d:
    print(f"{sys.argv[1]:60s} value {d['value']:.4g}  ms/step {d['ms_per_step']:.3f}  frac {d['roofline']['frac']:.4f}  in flight {d['config']['batches_in_flight_per_gpu']} x {d['config']['steps_per_library_call']}")
else:
    print(sys.argv[1], json.dumps(d)[:400])
PY
done
