"""Dev tool: concurrency figures from a rocprofv3 kernel trace CSV (several streams in flight)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
per = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][:60]
    per[name][0] += 1
    per[name][1] += (b - a) * 1e-6
    ev.append((a, 1)); ev.append((b, -1))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
busy = 0; cur = 0; last = t0; hist = collections.Counter()
for t, d in ev:
    if cur > 0: busy += t - last
    hist[min(cur, 20)] += t - last
    cur += d; last = t
span = (t1 - t0) * 1e-6
print(f"span {span:.1f} ms, >=1 kernel running {busy*1e-6:.1f} ms ({100*busy/(t1-t0):.1f} %), sum of kernel durations {sum(v[1] for v in per.values()):.1f} ms")
for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"  {k:60s} n={v[0]:6d} total {v[1]:9.1f} ms avg {1e3*v[1]/v[0]:8.1f} us")
print("time share by number of kernels in flight:", {k: round(100 * v / (t1 - t0), 1) for k, v in sorted(hist.items())})
