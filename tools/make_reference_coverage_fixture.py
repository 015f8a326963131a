"""Makes tests/golden/reference_coverage_polytopes.json from the coverage sets the reference ships as DATA
(/root/reference/src/slam/data/polytopes/polytope_coverage_[...].pkl: what ``MixedOrderBasisCircuitTemplate`` loads,
src/slam/basis.py:276-290 -- monodromy's output for ConversionGainGate bases, precomputed by the reference's authors).

The files are pickles of ``monodromy.coverage.CircuitPolytope`` objects; monodromy is not installed and nothing of it (or of the
reference) is imported or executed: a restricted unpickler maps every class named in the files (monodromy's polytopes, the reference's and
qiskit's gate objects) to an inert record that only keeps the attribute dictionary; ``fractions.Fraction`` and NumPy scalars are rebuilt.  What is kept per coverage entry: ``operations``
(gate keys), ``cost`` and the inequality / equality rows (integers) of its convex subpolytopes -- numbers, no code.

usage (in the build container only; the GPU box has no /root/reference): python3 tools/make_reference_coverage_fixture.py
"""
import fractions
import glob
import json
import os
import pickle
import re
import sys

SRC = "/root/reference/src/slam/data/polytopes"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_coverage_polytopes.json")


class _Record:
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {"state": state})


class _Restricted(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) == ("fractions", "Fraction"):
            return fractions.Fraction
        if (module, name) in (("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar")):
            import numpy._core.multiarray as m

            return m.scalar
        if (module, name) == ("numpy", "dtype"):
            import numpy

            return numpy.dtype
        # everything else (monodromy's polytope classes, the reference's gate objects, qiskit's): an inert record -- no code of theirs runs
        return type(name, (_Record,), {"_module": module})


def _num(v):
    if isinstance(v, fractions.Fraction):
        return [v.numerator, v.denominator] if v.denominator != 1 else v.numerator
    if isinstance(v, (int, float)):
        return v
    if hasattr(v, "item"):
        return v.item()
    raise TypeError(type(v))


def _entry(e):
    d = e.__dict__
    return {
        "operations": list(d.get("operations", [])),
        "cost": _num(d.get("cost")) if d.get("cost") is not None else None,
        "convex_subpolytopes": [
            {"inequalities": [[_num(x) for x in row] for row in cp.__dict__.get("inequalities", [])],
             "equalities": [[_num(x) for x in row] for row in cp.__dict__.get("equalities", [])]}
            for cp in d.get("convex_subpolytopes", [])
        ],
    }


def main():
    out = {}
    for path in sorted(glob.glob(os.path.join(SRC, "polytope_coverage_*.pkl"))):
        name = os.path.basename(path)
        if "smush" in name:
            continue  # time-sliced "smush" gates: out of scope
        with open(path, "rb") as f:
            obj = _Restricted(f).load()
        coverage, gate_hash = obj[0], obj[1]
        m = re.findall(r"2QGate\(([0-9.]+), ([0-9.]+), ([0-9.]+)\)", name)
        out[name] = {"gates": [[float(v) for v in g] for g in m], "gate_keys": list(gate_hash.keys()) if isinstance(gate_hash, dict) else None,
                     "coverage": [_entry(e) for e in coverage]}
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(f"{len(out)} coverage sets -> {OUT} ({os.path.getsize(OUT)} bytes)")
    for name, v in out.items():
        print(name, [(len(e["operations"]), e["cost"], len(e["convex_subpolytopes"])) for e in v["coverage"]])


def haar_volumes():
    """tests/golden/reference_haar_volumes.json: the Haar volumes of k-gate coverage sets the reference recorded
    (src/slam/data/extended_results.json, written by utils/gates/parallel_drive_volume.py:340-348,407,447-451: ``base_vol`` = monodromy's
    Haar integral of the coverage polytope of k applications of ConversionGainGate(0, 0, gc, gg, t)), with the gates' definitions
    (parallel_drive_volume.py:91-96)."""
    import math

    src = "/root/reference/src/slam/data/extended_results.json"
    rec = json.load(open(src))
    gates = {  # name: (gc, gg, t)   parallel_drive_volume.py:91-96
        "iSwap": (math.pi / 2, 0.0, 1.0), "sqiSwap": (math.pi / 2, 0.0, 0.5), "CNOT": (math.pi / 4, math.pi / 4, 1.0),
        "sqCNOT": (math.pi / 4, math.pi / 4, 0.5), "B": (3 * math.pi / 8, math.pi / 8, 1.0), "sqB": (3 * math.pi / 8, math.pi / 8, 0.5),
    }
    out = {}
    for name, per_k in rec.items():
        gc, gg, t = gates[name]
        out[name] = {"gc": gc, "gg": gg, "t": t, "base_vol": {k: float(v[0]) for k, v in per_k.items()}}
    path = os.path.join(os.path.dirname(OUT), "reference_haar_volumes.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(f"{len(out)} gates -> {path}")


if __name__ == "__main__":
    main()
    haar_volumes()
