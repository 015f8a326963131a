#!/usr/bin/env python3
"""Writes tests/golden/kat1_riswap_sweep.json: the 25 recorded ``c1c2c3`` triples of the KAT-1 circuit with its LAST
RiSwapGate swept, copied as DATA from the reference's recorded notebook output.

Source (reference checkout, read as text; needs /root/reference, runs in the build container only):
    src/slam/scripts/decomp_trajectory.ipynb cell 12 output = ``coordinate_list[-25:]`` of cell 10 with ``end = 11``:
    the full bound circuit of cells 5-6 (KAT-1 parameters, tests/golden/kat1.json) with RiSwap gates 1 and 2 at
    alpha = 0.5 and gate 3 at ``t in np.linspace(0, 0.5, 25)``; every triple mirrored on the x axis
    (``if c[0] > 0.5: c[0] = 1 - c[0]``).  It is the only recorded data that pins ``RiSwapGate(alpha)`` for
    alpha != 1/2, i.e. the forward path of CircuitTemplateV2's gate parameters (custom_gates.py:582-595)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NB = "/root/reference/src/slam/scripts/decomp_trajectory.ipynb"


def main():
    nb = json.load(open(NB))
    cell = nb["cells"][12]
    assert "".join(cell["source"]).strip() == "coordinate_list[-25:]"
    text = "".join(cell["outputs"][0]["data"]["text/plain"])
    triples = json.loads(text)
    assert len(triples) == 25 and all(len(t) == 3 for t in triples)
    out = {
        "source": "src/slam/scripts/decomp_trajectory.ipynb cell 12 output (coordinate_list[-25:] of cell 10, end = 11)",
        "circuit": "KAT-1 (tests/golden/kat1.json): gates 1, 2 = RiSwapGate(0.5), gate 3 = RiSwapGate(t)",
        "t": "numpy.linspace(0, 0.5, 25)",
        "mirror": "c1 > 0.5 -> 1 - c1 (cell 10)",
        "c1c2c3": triples,
    }
    path = os.path.join(ROOT, "tests", "golden", "kat1_riswap_sweep.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    sys.exit(main())
