#!/usr/bin/env python3
"""Generates tests/golden/quad_golden.json: frozen inputs + expected outputs (hex floats) of the function-argument forms of
cumtrapz / cumsimpson (integrate.nim:138-175, 377-400) from the CPU oracle, after it has been pinned to the reference's own
tests for them (tests/test_oracle_reference_kats.py::test_cumquad_fn_reference_tests).  Integrand: the arithmetic-only
polynomial ((a x + b) x)(1 + c) + d for component c (oracle kind RHS_POLY_T), so device results can be bit-exact.
Re-run:  python tests/golden/make_quad_golden.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def hx(a):
    return [float(v).hex() for v in np.asarray(a, dtype=np.float64).ravel()]


def main():
    import math
    Xs = {
        "harness17": O.linspace(0.0, 1.5 * math.pi, 17),         # tests/test_integrate.nim:19
        "unsorted": [0.3, -0.2, 1.7, 0.9, 1.7, -0.2],
        "sorted_dup_max": [0.0, 0.5, 0.5, 1.0, 1.0],
        "descending": [2.0, 1.0, 0.0],
        "negative": [-3.0, -1.0, -2.0, -1.5],
        "single_interval": [0.0, 1.0],
    }
    cases = []
    for xname, X in Xs.items():
        for rule in ("trapz", "simpson"):
            for dx in (0.01, 0.1, 0.37):
                for dim in (0, 3):
                    for params in ([0.75, -1.25, 0.5], [-2.0, 0.125, 3.0]):
                        r = O.cumquad_fn(rule, O.RHS_POLY_T, params, dim, X, dx)
                        cases.append({"name": f"{rule}_{xname}_dx{dx}_dim{dim}_a{params[0]}", "rule": rule, "X": hx(X), "dx": float(dx).hex(),
                                      "dim": dim, "params": hx(params), "rows": int(r.shape[0]), "out": hx(r)})
    out = {"generator": "tests/golden/make_quad_golden.py (oracle/ode_oracle.cpp: oracle_cumquad_fn)", "cases": cases}
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "quad_golden.json")
    json.dump(out, open(p, "w"), separators=(",", ":"))
    print(len(cases), "cases ->", p, os.path.getsize(p), "bytes")


if __name__ == "__main__":
    main()
