"""Dev tool (CPU): profile-derived versions of the roofline figures on bench.py's line, from the rocprofv3 kernel trace of the SAME
command ("done when the profile-derived figure is within 3 % of the line's").

usage: tools/r4_trace_summary.py <..._kernel_trace.csv> <bench line .json> [out.json]

The trace is cut into BURSTS (maximal stretches with at least one kernel running, separated by more than `gap_us` of idle chip).
  * a timed repetition of the driver's command = a burst with steps x 3 optimizer launches (several batches in flight): the union of
    its optimizer-kernel intervals is the time the chip spent on them; `frac_union` = the line's algorithmic flops / that union --
    to be compared with `roofline.frac` (flops / wall clock of the timed region);
  * the single-stream pass (bench.py `per_span`) = the burst with (warm_steps + per_span_steps) x 3 launches, one batch in flight, back to
    back: per span the mean duration of the last per_span_steps launches -- to be compared with `per_span[k].hip_event_ms`, and
    `frac_kernel` (dominant kernel, k = 1) recomputed from the trace's duration and the line's evaluations per launch.
"""
import collections
import csv
import json
import sys

PEAK = 78.6e12


def f_eval(k):
    return 3036 * k + 1247


def main():
    trace, line = sys.argv[1], sys.argv[2]
    out_path = sys.argv[3] if len(sys.argv) > 3 else None
    d = json.loads(open(line).read().strip().splitlines()[-1])
    steps = d["steps"]
    rows = []
    for r in csv.DictReader(open(trace)):
        name = r["Kernel_Name"]
        k = 0
        if "minimize_kernel<" in name:
            k = int(name[name.index("minimize_kernel<") + 16])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, name.split("(")[0][:48]))
    rows.sort()
    gap_ns = 300_000
    bursts, cur, end = [], [], None
    for a, b, k, n in rows:
        if cur and a > end + gap_ns:
            bursts.append(cur)
            cur = []
            end = None
        cur.append((a, b, k, n))
        end = b if end is None else max(end, b)
    if cur:
        bursts.append(cur)

    def union(iv):
        tot, e = 0, None
        for a, b in sorted(iv):
            if e is None or a > e:
                tot += b - a
                e = b
            elif b > e:
                tot += b - e
                e = b
        return tot

    rl = d["roofline"]
    flops = sum(rl["evals_per_span"][str(k)] * f_eval(k) for k in (1, 2, 3))
    timed = []
    for b in bursts:
        nk = sum(1 for x in b if x[2] in (1, 2, 3))
        if nk == 3 * steps:
            opt = [(a, e) for a, e, k, _ in b if k]
            allk = [(a, e) for a, e, _, _ in b]
            timed.append({"optimizer_union_ms": union(opt) * 1e-6, "all_kernels_union_ms": union(allk) * 1e-6,
                          "span_ms": (max(e for _, e in allk) - min(a for a, _ in allk)) * 1e-6,
                          "sum_optimizer_durations_ms": sum(e - a for a, e in opt) * 1e-6,
                          "sum_epilogue_durations_ms": sum(e - a for a, e, k, n in b if "epilogue" in n) * 1e-6})
    res = {"source_trace": trace.split("/")[-1], "line": {"ms_per_step": d["ms_per_step"], "frac": rl["frac"], "frac_kernel": rl.get("frac_kernel")},
           "bursts": len(bursts), "timed_repetitions_found": len(timed)}
    if timed:
        timed.sort(key=lambda t: t["all_kernels_union_ms"])
        med = timed[(len(timed) - 1) // 2]
        res["timed_region"] = {
            **{k: round(v, 3) for k, v in med.items()},
            "wall_ms_on_the_line": round(d["ms_per_step"] * steps, 3),
            "frac_union": flops / (med["optimizer_union_ms"] * 1e-3) / PEAK,
            "frac_all_kernels_union": flops / (med["all_kernels_union_ms"] * 1e-3) / PEAK,
            "frac_on_the_line": rl["frac"],
            "ratio_union_to_line": (flops / (med["optimizer_union_ms"] * 1e-3) / PEAK) / rl["frac"],
            "epilogue_share_of_summed_kernel_time": med["sum_epilogue_durations_ms"] / (med["sum_optimizer_durations_ms"] + med["sum_epilogue_durations_ms"]),
        }
    ps = rl.get("per_span")
    if ps:
        n_solo = ps["1"]["launches"]
        n_warm = int(ps.get("all", {}).get("warm_steps", 1))
        for b in bursts:
            ks = [x for x in b if x[2] in (1, 2, 3)]
            if len(ks) == 3 * (n_solo + n_warm) and len(ks) != 3 * steps:
                solo = {}
                for k in (1, 2, 3):
                    dur = [(e - a) * 1e-6 for a, e, kk, _ in ks if kk == k][n_warm:]  # the first launches of the pass are its untimed warm steps
                    ms = sum(dur) / len(dur)
                    ev = ps[str(k)]["evals_per_launch"]
                    solo[str(k)] = {"trace_ms": round(ms, 4), "line_hip_event_ms": round(ps[str(k)]["hip_event_ms"], 4),
                                    "ratio": ms / ps[str(k)]["hip_event_ms"], "frac_from_trace": ev * f_eval(k) / (ms * 1e-3) / PEAK,
                                    "frac_on_the_line": ps[str(k)]["frac"]}
                res["single_stream_pass"] = solo
                res["frac_kernel_from_trace"] = solo["1"]["frac_from_trace"]
                break
    per = collections.defaultdict(lambda: [0, 0.0])
    for a, e, k, n in rows:
        per[n][0] += 1
        per[n][1] += (e - a) * 1e-6
    res["kernels"] = {n: {"calls": c, "total_ms": round(t, 2), "avg_us": round(1e3 * t / c, 1)} for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:8]}
    txt = json.dumps(res, indent=1)
    print(txt)
    if out_path:
        open(out_path, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
