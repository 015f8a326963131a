"""Summarise rocprofv3 --pmc CSVs per kernel: per-dispatch averages of each counter."""
import collections
import csv
import glob
import sys

out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
dur = collections.defaultdict(list)
for f in glob.glob(f"{out}/pmc*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "minimize_kernel" not in k:
            continue
        k = k[k.index("minimize_kernel") : k.index("minimize_kernel") + 18]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
        if r["Counter_Name"] in ("SQ_WAVES", "FETCH_SIZE", "WRITE_SIZE"):
            dur[(k, r["Counter_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        print(f"   {c:24s} avg/dispatch {agg[k][c] / cnt[k][c]:16.1f}   dispatches {cnt[k][c]}")
    for (kk, c), d in dur.items():
        if kk == k:
            print(f"   avg dispatch duration under --pmc ({c} pass): {sum(d) / len(d):.1f} us")
    a = agg[k]
    n = cnt[k]
    if "SQ_ACTIVE_INST_VALU" in a and "SQ_WAVE_CYCLES" in a:
        print(f"   VALU-active / wave-cycles = {a['SQ_ACTIVE_INST_VALU'] / a['SQ_WAVE_CYCLES']:.3f}   wait_any / wave-cycles = {a['SQ_WAIT_ANY'] / a['SQ_WAVE_CYCLES']:.3f}")
    if "FETCH_SIZE" in a:
        print(f"   FETCH_SIZE avg {a['FETCH_SIZE'] / n['FETCH_SIZE']:.1f} KB/dispatch (x2 on gfx950 for wide streaming reads, MI355X_MICROARCH.md)")
    if "WRITE_SIZE" in a:
        print(f"   WRITE_SIZE avg {a['WRITE_SIZE'] / n['WRITE_SIZE']:.1f} KB/dispatch")
