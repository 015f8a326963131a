#!/bin/bash
# GPU box: HBM traffic of the wavefront-per-item kernels (tools/r5_long_probe.py: launches in the order k = 6, 6, 8, 8, 12, 12, 16, 16, then
# the real 8-gate stage) -- FETCH_SIZE and WRITE_SIZE in separate passes, HBM bytes = 1024 (2 FETCH_SIZE + WRITE_SIZE) as in
# tools/profile_r5.sh (MI355X_MICROARCH.md, gfx950 correction), against the 8 n^2 bytes per evaluation the metric's pass moves.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/long_pmc; rm -rf $OUT; mkdir -p $OUT
for P in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES"; do
  tag=$(echo $P | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $P --output-format csv -d $OUT/$tag -- python3 tools/r5_long_probe.py > $OUT/$tag.txt 2> $OUT/$tag.err || { tail -5 $OUT/$tag.err; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, re, sys
out = sys.argv[1]
def rows(tag, counter):
    f = glob.glob(f"{out}/{tag}/*/*_counter_collection.csv")[0]
    r = [x for x in csv.DictReader(open(f)) if "minimize_long_kernel" in x["Kernel_Name"] and x["Counter_Name"] == counter]
    r.sort(key=lambda x: int(x["Start_Timestamp"]))
    return r
fe, wr, va = rows("FETCH_SIZE", "FETCH_SIZE"), rows("WRITE_SIZE", "WRITE_SIZE"), rows("SQ_INSTS_VALU", "SQ_INSTS_VALU")
evals = [int(m.group(1)) for m in re.finditer(r"steady k=\d+: [\d.]+ ms, (\d+) evals", open(f"{out}/FETCH_SIZE.txt").read())]
ks = [6, 6, 8, 8, 12, 12, 16, 16, 8]
for i, k in enumerate(ks[: len(fe)]):
    n = 6 * (k + 1)
    dur = (int(fe[i]["End_Timestamp"]) - int(fe[i]["Start_Timestamp"])) * 1e-9
    hbm = 1024.0 * (2 * float(fe[i]["Counter_Value"]) + float(wr[i]["Counter_Value"]))
    ev = evals[i // 2] if i < 8 else None
    line = f"launch {i} k={k:2d}: {1e3 * dur:7.3f} ms  HBM {hbm / 1e6:9.1f} MB = {hbm / dur / 1e12:5.2f} TB/s  VALU instr/wave-launch {float(va[i]['Counter_Value']):.3g}"
    if ev:
        line += f"  | per evaluation: HBM {hbm / ev:8.0f} B, metric pass 8 n^2 = {8 * n * n} B, ratio {hbm / ev / (8 * n * n):.2f}"
    print(line)
PY
