#!/bin/bash
# GPU box: dynamic VALU instructions per lock-step round (steady-state workload), one PMC pass per (gate, k).
# usage: tools/valu_per_round.sh "<gates>" "<ks>"   [SLAM_HIP_LIB honoured]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for g in ${1:-sqiswap}; do for k in ${2:-1 2 3}; do
  OUT=gpurun_out/vpr_tmp; rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/pmc1 -- python3 tools/steady.py $g $k 8 > $OUT/run.txt 2> $OUT/err.txt || { tail -3 $OUT/err.txt; exit 1; }
  python3 - "$OUT" "$g" "$k" <<'PY'
import sys, csv, glob, collections
out, g, k = sys.argv[1:4]
run = open(f"{out}/run.txt").read().split()
wr = int(run[run.index("wave_rounds") + 1]); ev = int(run[run.index("evals") + 1]); ms = float(run[run.index("ms") + 1])
agg = collections.defaultdict(list)
for f in glob.glob(f"{out}/pmc1/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "minimize_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
v = {c: sum(x) / len(x) for c, x in agg.items()}
print(f"{g} k={k}: VALU/round {v['SQ_INSTS_VALU']/wr:7.1f}  SALU/round {v['SQ_INSTS_SALU']/wr:6.1f}  LDS/round {v['SQ_INSTS_LDS']/wr:5.1f}  (last run: {ms:.2f} ms, {ev/ms/1e6:.3f} G evals/s)")
PY
done; done
rm -rf gpurun_out/vpr_tmp
