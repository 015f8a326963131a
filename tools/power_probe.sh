#!/bin/bash
# GPU box: sample clocks / power with rocm-smi while the cfg3 workload runs (is the fp64 loop power-limited?)
python bench.py --workload cfg3 --no-cpu-baseline --steps 150 --warmup 3 > /tmp/pp.json 2>/dev/null &
BP=$!
sleep 2.5
for i in 1 2 3 4; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction)" | head -8
  echo "--"
  sleep 0.7
done
wait $BP
tail -1 /tmp/pp.json | cut -c1-120
echo "idle:"; sleep 1; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -4
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -3
