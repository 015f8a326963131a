#!/bin/bash
# dev tool (GPU): bench cfg2 over launch shaping x batches in flight
for ipq in ${IPQS:-0 2 4 8}; do for s in ${STREAMS:-16 32}; do
  python bench.py --no-cpu-baseline --workload ${WL:-cfg2} --items-per-quad $ipq --streams $s --steps ${STEPS:-200} --warmup ${WARM:-20} > /tmp/b.json || exit 1
  python -c "import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print('ipq=$ipq streams=$s', d['value'], d['ms_per_step'], round(d['roofline']['frac'],4))"
done; done
