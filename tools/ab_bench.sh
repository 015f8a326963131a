#!/bin/bash
# GPU box: A/B of library variants on bench.py.  usage: tools/ab_bench.sh "<variants>" <bench args...>
V=$1; shift
for rep in 1 2; do for v in $V; do
  if [ "$v" = cur ]; then unset SLAM_HIP_LIB; else export SLAM_HIP_LIB=$PWD/slam_decomposition_amd/lib/ab/$v.so; fi
  python bench.py --no-cpu-baseline "$@" > /tmp/b.json 2>/tmp/b.err || { tail -3 /tmp/b.err; exit 1; }
  python -c "import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print('$v rep $rep', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"
done; done
