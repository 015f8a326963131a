"""Dev tool (CPU): rewrite the numeric cells of DESIGN.md's round-3 numbers table from the committed bench lines
(profiles/r3_bench_*.json), so that the table always says what the files say.  usage: python tools/r3_tables.py [--check]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def line(name):
    return json.loads(open(os.path.join(ROOT, "profiles", f"r3_bench_{name}.json")).read().strip().splitlines()[-1])


def sci(v):
    m, e = f"{v:.2e}".split("e")
    return f"{m}e{int(e)}"


def ms(d, digits=2):
    return f"{d['ms_per_step']:.{digits}f} ({d['ms_per_step_min']:.{digits}f} … {d['ms_per_step_max']:.{digits}f})"


def cells(name, bold=False, digits=2):
    d = line(name)
    r = d["roofline"]
    b = "**" if bold else ""
    return [f"{b}{sci(d['value'])}{b}", ms(d, digits), f"{b}{r['frac']:.3f}{b}", f"{r['frac_accepted']:.3f}"]


def main():
    d = line("default")
    ps = d["roofline"]["per_span"]
    v2 = d["secondary"]["v2"]
    cb = d["cpu_baseline"]
    two = line("2ranks_one_gpu_filecomm")
    two_s = line("2ranks_one_gpu_filecomm_strong")
    rows = {
        "| **cfg3 (default)**": cells("default", bold=True),
        "| cfg3, one batch in flight": [
            f"{sci(65536 / (ps['all']['hip_event_ms_per_step'] * 1e-3))} ({ps['all']['hip_event_ms_per_step']:.1f} ms of kernels)", "—",
            "**" + " / ".join(f"{ps[k]['frac']:.3f}" for k in "123") + "** at k = 1 / 2 / 3", "—"],
        "| launcher + RCCL, one rank": cells("launcher_rccl_world1"),
        "| cfg3, `--fast-exit`": cells("cfg3_fast_exit"),
        "| cfg3, `--span-rules`": cells("cfg3_span_rules"),
        "| cfg2: CNOT, 1024 × 16, 20 steps per library call": cells("cfg2", bold=True, digits=3),
        "| cfg2 with the driver's `--steps 20 --warmup 5`": cells("cfg2_20", digits=3),
        "| cfg2, one step per call, one in flight": cells("cfg2_1stream"),
        "| cfg4 shard": cells("cfg4"),
        "| cfg5 shard": cells("cfg5"),
        "| 2 ranks sharing the one GPU": [f"{sci(two['value'])} / {sci(two_s['value'])} (whole job)", f"{two['ms_per_step']:.1f} / {two_s['ms_per_step']:.1f}", "—", "—"],
        "| `secondary.v2`": [f"**{sci(v2['value'])}**", f"{v2['ms_per_step']:.3f}", f"**{v2['roofline_frac']:.3f}**", "—"],
        "| CPU port": [f"{cb['value']:.1f} (finite differences, as the reference) / {cb['analytic_jac']['value']:.1f} (analytic gradient)", "—", "—", "—"],
    }
    path = os.path.join(ROOT, "HISTORY.md")  # (the round-3 document; DESIGN.md is the current design)
    text = open(path).read()
    start = text.index("Numbers measured on MI355X (round 3 build")
    end = text.index("**Why the k = 1 launch writes", start)
    block = text[start:end].split("\n")
    changed = 0
    for i, row in enumerate(block):
        for prefix, new in rows.items():
            if row.startswith(prefix):
                parts = row.split(" | ")
                old = parts[1:5]
                parts[1:5] = new
                if old != new:
                    changed += 1
                block[i] = " | ".join(parts)
    new_text = text[:start] + "\n".join(block) + text[end:]
    if "--check" in sys.argv:
        print("rows that differ from the committed bench lines:", changed)
        return 1 if changed else 0
    open(path, "w").write(new_text)
    print("rows rewritten:", changed)
    return 0


if __name__ == "__main__":
    sys.exit(main())
