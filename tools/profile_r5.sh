#!/bin/bash
# Runs on the GPU box (via gpurun): the rocprofv3 evidence behind bench.py's roofline numbers (round 5).
#   headline:       kernel-trace stats of the driver's command (5 batches in flight; cfg3)
#   single_stream:  the same workload with ONE batch in flight -- every launch alone on the chip, so the per-kernel
#                   averages here are what roofline.per_span[k].hip_event_ms reports (bench.py's per-span pass)
#   pmc1..3:        counter passes over the single-stream run (SQ_*; FETCH_SIZE; WRITE_SIZE + LDS), never combined
#                   with tracing (gpurun refuses --pmc together with hip/hsa traces)
#   v2, cfg2:       kernel-trace stats of the secondary measurements (CircuitTemplateV2; 1024 x 16 CNOT grouped 20 per call)
# writes pmc.json (valu_busy / hbm_gbps per span: what bench.py puts on its line) and traffic.json
# usage: tools/profile_r5.sh <tag> [workload]
set -o pipefail
TAG=${1:-r5}; WL=${2:-cfg3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
COMMON="--workload $WL --no-cpu-baseline --no-secondary"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_headline -- python3 bench.py --steps 20 --warmup 5 $COMMON > $OUT/headline_bench_under_trace.json 2> $OUT/trace_headline.err || { tail -5 $OUT/trace_headline.err; exit 1; }
cp $OUT/trace_headline/*/*_kernel_stats.csv $OUT/headline_kernel_stats.csv
# profile-derived versions of the line's roofline figures (union of the optimizer launches' intervals in a timed repetition; the
# single-stream launches at the end): VERDICT r3 item 1a
python3 tools/r4_trace_summary.py $(ls $OUT/trace_headline/*/*_kernel_trace.csv | head -1) $OUT/headline_bench_under_trace.json $OUT/trace_summary.json > /dev/null
python3 tools/trace_concurrency.py $(ls $OUT/trace_headline/*/*_kernel_trace.csv | head -1) > $OUT/headline_concurrency.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_single -- python3 bench.py --streams 1 --steps 6 --warmup 2 --repeats 1 $COMMON > $OUT/single_stream_bench_under_trace.json 2> $OUT/trace_single.err || { tail -5 $OUT/trace_single.err; exit 1; }
cp $OUT/trace_single/*/*_kernel_stats.csv $OUT/single_stream_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_v2 -- python3 bench.py --v2-only > $OUT/v2_bench_under_trace.json 2> $OUT/trace_v2.err || { tail -5 $OUT/trace_v2.err; exit 1; }
cp $OUT/trace_v2/*/*_kernel_stats.csv $OUT/v2_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg2 -- python3 bench.py --workload cfg2 --no-cpu-baseline --per-span-steps 0 > $OUT/cfg2_bench_under_trace.json 2> $OUT/trace_cfg2.err || { tail -5 $OUT/trace_cfg2.err; exit 1; }
cp $OUT/trace_cfg2/*/*_kernel_stats.csv $OUT/cfg2_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg5 -- python3 bench.py --workload cfg5 --steps 160 --warmup 16 --no-cpu-baseline --no-secondary --per-span-steps 0 > $OUT/cfg5_bench_under_trace.json 2> $OUT/trace_cfg5.err || { tail -5 $OUT/trace_cfg5.err; exit 1; }
cp $OUT/trace_cfg5/*/*_kernel_stats.csv $OUT/cfg5_kernel_stats.csv
# the long-template kernels (one stage of 8 / 12 weak gates) and the API paths
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_long -- python3 bench.py --long-only > $OUT/long_bench_under_trace.json 2> $OUT/trace_long.err || { tail -5 $OUT/trace_long.err; exit 1; }
cp $OUT/trace_long/*/*_kernel_stats.csv $OUT/long_kernel_stats.csv
bash tools/valu_per_round.sh "sqiswap cx" "1 2 3" > $OUT/valu_per_round.txt 2> $OUT/valu_per_round.err || { tail -5 $OUT/valu_per_round.err; exit 1; }
P1="SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"
P2="FETCH_SIZE GRBM_GUI_ACTIVE"
P3="WRITE_SIZE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc$i -- python3 bench.py --streams 1 --steps 2 --warmup 1 --repeats 1 $COMMON --per-span-steps 0 > $OUT/bench_pmc$i.json 2> $OUT/pmc$i.err || { tail -5 $OUT/pmc$i.err; exit 1; }
done
python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt
python3 - "$OUT" "$WL" <<'PY'
import collections, csv, glob, json, sys
out, wl = sys.argv[1:3]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(f"{out}/pmc*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "minimize_kernel<" in k:
            span = k[k.index("minimize_kernel<") + 16]
            agg[span][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "FETCH_SIZE":
                dur[span].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
mean = lambda v: sum(v) / len(v)
traffic = {s: int(1024 * (2 * mean(v["FETCH_SIZE"]) + mean(v["WRITE_SIZE"]))) for s, v in sorted(agg.items())}
json.dump({"unit": "bytes per minimize_kernel launch (HBM: 2 x FETCH_SIZE + WRITE_SIZE in KB -> bytes, MI355X_MICROARCH.md gfx950 correction), rocprofv3 --pmc, separate passes, one batch in flight",
           wl: traffic}, open(f"{out}/traffic.json", "w"), indent=1)
pmc = {"unit": {"valu_active_per_wave_cycle": "SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES", "valu_busy": "that x resident waves per SIMD (fraction of a SIMD's cycles with a VALU instruction issuing)",
                "hbm_gbps": "(2 x FETCH_SIZE + WRITE_SIZE) / dispatch duration of the FETCH_SIZE pass, GB/s (MI355X_MICROARCH.md gfx950 correction)"},
       "source": "tools/profile_r5.sh: rocprofv3 --pmc over `bench.py --streams 1` (one batch in flight), separate passes", wl: {}}
for s, v in sorted(agg.items()):
    va = mean(v["SQ_ACTIVE_INST_VALU"]) / mean(v["SQ_WAVE_CYCLES"])
    wps = 2 if s in "12" else 1
    pmc[wl][s] = {"valu_active_per_wave_cycle": round(va, 4), "waves_per_simd": wps, "valu_busy": round(va * wps, 4), "hbm_bytes_per_launch": traffic[s],
                  "hbm_gbps": round(traffic[s] / mean(dur[s]) / 1e9, 2), "launch_us_under_pmc": round(mean(dur[s]) * 1e6, 1),
                  "wait_any_per_wave_cycle": round(mean(v["SQ_WAIT_ANY"]) / mean(v["SQ_WAVE_CYCLES"]), 4),
                  "lds_bank_conflict_frac": round(mean(v["SQ_LDS_BANK_CONFLICT"]) / mean(v["SQ_LDS_IDX_ACTIVE"]), 4)}
json.dump(pmc, open(f"{out}/pmc.json", "w"), indent=1)
print(json.dumps(pmc[wl]))
PY
rm -rf $OUT/trace_headline $OUT/trace_single $OUT/trace_v2 $OUT/trace_cfg2 $OUT/trace_cfg5 $OUT/trace_long $OUT/pmc?
cat $OUT/trace_summary.json | head -60; cat $OUT/valu_per_round.txt; head -6 $OUT/long_kernel_stats.csv; head -8 $OUT/headline_kernel_stats.csv; head -6 $OUT/single_stream_kernel_stats.csv; head -5 $OUT/v2_kernel_stats.csv; head -6 $OUT/cfg2_kernel_stats.csv
