#!/bin/bash
# GPU box: default bench over batches in flight (and steps).  usage: tools/sweep_steps.sh "<streams list>" [steps]
for rep in 1 2; do for s in ${1:-5 8 10 12 16 20}; do
  python bench.py --gpus 1 --warmup 5 --steps ${2:-20} --no-cpu-baseline --no-secondary --streams $s > /tmp/b.json 2>/tmp/b.err || { tail -3 /tmp/b.err; exit 1; }
  python -c "import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print('streams $s steps ${2:-20} rep $rep', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"
done; done
