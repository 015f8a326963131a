"""Dev tool: from a rocprofv3 kernel trace of tools/r5_api_timeline.py, the kernel timeline of the LAST job (per queue) and its totals."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void slamdev::", "").replace("slamdev::", ""), r.get("Queue_Id", "?")) for r in rows))
# jobs: separated by idle gaps > 0.8 ms
jobs, cur, end = [], [], 0
for e in ev:
    if cur and e[0] - end > 0.8e6:
        jobs.append(cur); cur = []
    cur.append(e); end = max(end, e[1])
jobs.append(cur)
jobs = [j for j in jobs if sum(1 for e in j if "minimize" in e[2]) >= 10]
print(len(jobs), "jobs; durations (ms):", [round((max(e[1] for e in j) - j[0][0]) / 1e6, 2) for j in jobs])
j = jobs[int(sys.argv[2]) if len(sys.argv) > 2 else -1]
j0 = j[0][0]
qs = sorted({e[3] for e in j})
for q in qs:
    print("queue", q)
    for e in j:
        if e[3] == q and (e[1] - e[0] > 0.05e6 or "haar" in e[2]):
            print("   %-30s %7.2f -> %7.2f  (%.2f ms)" % (e[2][:30], (e[0] - j0) / 1e6, (e[1] - j0) / 1e6, (e[1] - e[0]) / 1e6))
tot = {}
for e in j:
    tot[e[2][:30]] = tot.get(e[2][:30], 0) + (e[1] - e[0]) / 1e6
print({k: round(v, 2) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])})
