#!/bin/bash
# dev tool (GPU): bench cfg2 over HW queues x streams x launch shaping
for hq in ${HQS:-16 32}; do for s in ${STREAMS:-16 32}; do for ipq in ${IPQS:-0 4}; do
  GPU_MAX_HW_QUEUES=$hq python bench.py --no-cpu-baseline --items-per-quad $ipq --streams $s --steps ${STEPS:-200} --warmup ${WARM:-20} > /tmp/b.json || exit 1
  python -c "import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print('hwq=$hq streams=$s ipq=$ipq', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"
done; done; done
