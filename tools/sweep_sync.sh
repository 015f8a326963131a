#!/bin/bash
# dev tool (GPU): host wait mode x streams x launch shaping on the default bench
for m in ${MODES:-spin block yield}; do for s in ${STREAMS:-16 32}; do for ipq in ${IPQS:-0 4}; do
  SLAM_HOST_WAIT=$m python bench.py --no-cpu-baseline --items-per-quad $ipq --streams $s --steps ${STEPS:-200} --warmup ${WARM:-20} > /tmp/b.json 2>/tmp/b.err || { tail -3 /tmp/b.err; exit 1; }
  python -c "import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print('mode=$m streams=$s ipq=$ipq', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"
  true
done; done; done
