#!/bin/bash
# GPU box: vector instructions per 16 evaluations (one lock-step round of a full wavefront) of the CircuitTemplateV2 optimizer kernels beside
# the fixed-gate ones, same targets: SQ_INSTS_VALU summed over all launches of tools/r5_v2_valu.py / its evaluations per span.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/v2_valu; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT/pmc -- python3 tools/r5_v2_valu.py > $OUT/run.json 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
ev = json.loads(open(f"{out}/run.json").read().strip().splitlines()[-1])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"{out}/pmc/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for fam, tag in (("v2", "minimize_v2_kernel<"), ("fixed", "minimize_kernel<")):
            if tag in k:
                span = int(k[k.index(tag) + len(tag)])
                agg[(fam, span)][r["Counter_Name"]] += float(r["Counter_Value"])
F = {"fixed": lambda k: 3036 * k + 1247, "v2": lambda k: 3036 * k + 1247 + 488 * k}
for (fam, span), c in sorted(agg.items()):
    e = ev[fam][span]
    valu = c["SQ_INSTS_VALU"] / (e / 16.0)
    alg = F[fam](span) * 16 / 64 / 2
    print(f"{fam:5s} k={span}: {e:11d} evaluations, VALU per 16 evaluations {valu:7.1f}, SALU {c['SQ_INSTS_SALU'] / (e / 16.0):6.1f}; algorithmic FMA-equivalents {alg:6.1f} -> {alg / valu:.3f}")
PY
