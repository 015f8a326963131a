"""bench: figures read from the committed rocprofv3 passes under profiles/ (counters cannot be collected inside an unprofiled run)."""
from __future__ import annotations

import json
import os

from .workloads import ROOT


def traffic_per_launch(workload: str):
    """HBM bytes per optimizer-kernel launch (mean over the three spans) from the committed PMC passes
    (profiles/r2_traffic.json, else r1d_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md HBM section);
    None for workloads that were not profiled."""
    for name in ("r5_traffic.json", "r4_traffic.json", "r3_traffic.json", "r2_traffic.json", "r1d_traffic.json"):
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", name)))[workload]
            return sum(t.values()) / len(t)
        except (OSError, KeyError, ValueError):
            continue
    return None


def pmc_figures(workload: str):
    """VALU-busy and achieved HBM GB/s of the optimizer launches, per span, from the committed rocprofv3 --pmc passes
    (profiles/r3_pmc.json, else r2_pmc.json; tools/profile_r3.sh writes them).  Counters cannot be collected inside an
    unprofiled run: these are the figures of the committed profile of the same command, named in `source`."""
    for name in ("r5_pmc.json", "r4_pmc.json", "r3_pmc.json", "r2_pmc.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            per = d[workload]
            return {"source": f"profiles/{name}", "valu_busy": {k: v["valu_busy"] for k, v in per.items()},
                    "hbm_gbps": {k: v["hbm_gbps"] for k, v in per.items()}}
        except (OSError, KeyError, ValueError):
            continue
    return None
