#!/bin/bash
# GPU box: the round's standing check -- GPU test suite (or a -k selection), then the driver's bench command; everything under gpurun_out/r5_check/.
# usage: tools/r5_check.sh [pytest -k expression | "none"] [bench args | "nobench"]
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_check; mkdir -p $OUT
if [ "$1" != "none" ]; then
  if [ -n "$1" ]; then
    timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "$1" > $OUT/pytest.txt 2>&1; rc=$?
  else
    timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; rc=$?
  fi
  tail -5 $OUT/pytest.txt
  if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" $OUT/pytest.txt | tail -30; exit 1; fi
fi
[ "$2" = "nobench" ] && exit 0
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; rc=$?
if [ $rc -ne 0 ]; then tail -20 $OUT/bench_default.err; exit 1; fi
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_check/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.4g ms/step %.3f [%.3f..%.3f] frac %.3f frac_kernel %s" % (d["value"], d["ms_per_step"], d["ms_per_step_min"], d["ms_per_step_max"], r["frac"], r.get("frac_kernel")))
print("per_span", {k: (round(v["hip_event_ms"], 3), round(v["frac"], 3)) for k, v in r["per_span"].items() if k != "all"}, r["per_span"]["all"])
print("parity", {k: v for k, v in (d.get("parity_sample") or {}).items() if k != "what"})
print("cpu", {k: d["cpu_baseline"][k] for k in ("value", "cores", "per_core")})
s = d.get("secondary", {})
print("cfg2", s.get("cfg2", {}).get("value"), s.get("cfg2", {}).get("roofline_frac"))
print("v2", s.get("v2", {}).get("value"), s.get("v2", {}).get("roofline_frac"))
print("api", {k: v for k, v in (s.get("api") or {}).items() if k in ("value", "wall_ms", "wall_ms_all")})
print("api_large", {k: v for k, v in (s.get("api_large") or {}).items() if k != "workload"})
print("medium", s.get("medium_call"))
PY
