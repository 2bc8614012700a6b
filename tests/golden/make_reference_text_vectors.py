#!/usr/bin/env python3
"""Generates tests/golden/reference_text_vectors.json by EXECUTING THE REFERENCE'S OWN SOURCE TEXT.

The reference is Nim and the image has no Nim compiler; oracle/nim_subset.py is an interpreter for the subset of Nim that
src/numericalnim/ode.nim (solveODE, ODESolver, the 14 *_step procs, commonAdaptiveMethodCode, newODEoptions) and
utils.nim's hermiteSpline, linspace and Vector[T] operators are written in.  This script reads those files from /root/reference (build container only: the
reference does not travel), runs solveODE from the text on the INPUTS of every case of tests/golden/ode_golden.json (the
fixtures the oracle generated) and also one IntegratorProc call per integrator, and stores the outputs as hex floats.

What is stored is data (inputs + expected outputs), never reference text.  Consumers:
  tests/test_reference_text_pin.py   oracle == these vectors, bit for bit (any host); interpreter re-run == these vectors (here)
  tests/test_gpu_golden.py           HIP path == these vectors, bit for bit (GPU box)
Re-run:  python tests/golden/make_reference_text_vectors.py        (about ten minutes: utils.nim's Vector operators are interpreted too)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nim_subset as N  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "reference_text_vectors.json")


def fh(xs):
    return [float.fromhex(x) for x in xs]


def hx(xs):
    return [float(v).hex() for v in xs]


# The user's right-hand sides f(t, y, ctx) of the fixtures (enum nnhip_rhs_kind / oracle RHS_*), written as a numericalnim user would write
# them in Nim over float / Vector[float] — same expression order as include/nnhip_ode.h documents.  These are the CALLER's procs, not
# reference code.  A Vector[float] state is the reference's own object (built and operated on by utils.nim's procs, interpreted) unless the
# interpreter was loaded with interpret_vector=False.
def make_rhs(it, kind, p, dim):
    V = lambda comps: N.vector(it, comps)
    C = N.components
    if kind == 0:  # NEG_Y: -y
        return (lambda t, y, ctx: -y) if dim == 0 else (lambda t, y, ctx: V([-a for a in C(y)]))
    if kind == 1:  # LINEAR: y * p0
        return (lambda t, y, ctx: y * p[0]) if dim == 0 else (lambda t, y, ctx: V([a * p[0] for a in C(y)]))
    if kind == 2:  # LORENZ
        def lorenz(t, y, ctx):
            c = C(y)
            return V([p[0] * (c[1] - c[0]), c[0] * (p[1] - c[2]) - c[1], c[0] * c[1] - p[2] * c[2]])
        return lorenz
    if kind == 3:  # RING: -((c+1)/d)*y_c + p0*y_{(c+1) mod d}
        def ring(t, y, ctx):
            c = C(y)
            return V([-((k + 1) / dim) * c[k] + p[0] * c[(k + 1) % dim] for k in range(dim)])
        return ring
    if kind == 4:  # AFFINE_T: p0*y + p1*t
        return (lambda t, y, ctx: p[0] * y + p[1] * t) if dim == 0 else (lambda t, y, ctx: V([p[0] * a + p[1] * t for a in C(y)]))
    if kind == 5:  # VANDERPOL
        def vdp(t, y, ctx):
            c = C(y)
            return V([c[1], p[0] * ((1.0 - c[0] * c[0]) * c[1]) - c[0]])
        return vdp
    raise ValueError(kind)


def to_state(it, vals, dim):
    return vals[0] if dim == 0 else N.vector(it, vals)


def flat(v):
    return [v] if isinstance(v, float) else N.components(v)


def solve_case(it, c):
    """solveODE(f, y0, tspan, options, integrator = ...) from the reference's text for every IVP of one fixture."""
    dim, p = c["dim"], fh(c["params"])
    f = make_rhs(it, c["rhs_kind"], p, dim)
    opt = it.call("newODEoptions", **c["options"])
    tspan = fh(c["tspan"])
    out, t = [], None
    for y0 in c["y0"]:
        t, ys = it.call("solveODE", f, to_state(it, fh(y0), dim), list(tspan), opt, integrator=c["integrator"])
        out.append({"y": hx([x for row in ys for x in flat(row)]), "n_y": len(ys)})
    return {"name": c["name"], "t": hx(t), "ivps": out}


STEP_INPUTS = [  # (t, y, dt, options) for one IntegratorProc call per integrator on Lorenz: accepted at once / in-step retries / dtMin double hit
    (0.3, [-8.1, -7.9, 27.2], 1e-3, {}),
    (0.3, [-8.1, -7.9, 27.2], 0.05, dict(absTol=1e-8, relTol=1e-8, dtMin=1e-7, dtMax=1.0)),
    (-1.5, [3.0, 4.5, 20.0], 0.5, dict(absTol=1e-12, relTol=1e-12, dtMin=1e-1, dtMax=1.0)),
]
LORENZ_P = [10.0, 28.0, 8.0 / 3.0]


def step_cases(it):
    f = make_rhs(it, 2, LORENZ_P, 3)
    out = []
    for name, proc in N.STEP_PROCS.items():
        for k, (t, y, dt, okw) in enumerate(STEP_INPUTS):
            opt = it.call("newODEoptions", **okw)
            yv = N.vector(it, y)
            fsal = f(t, yv, None)
            yNew, fs, dtUsed, err = it.call(proc, f, t, yv, fsal, dt, opt, None)
            out.append({"integrator": name, "input": k, "t": float(t).hex(), "y": hx(y), "fsal": hx(flat(fsal)), "dt": float(dt).hex(), "options": okw,
                        "yNew": hx(flat(yNew)), "fsalOut": hx(flat(fs)), "dtUsed": float(dtUsed).hex(), "error": float(err).hex()})
    return out


def main():
    if not N.reference_available():
        sys.exit("needs /root/reference (build container only)")
    it = N.load_reference_ode()
    golden = json.load(open(os.path.join(HERE, "ode_golden.json")))["cases"]
    t0 = time.time()
    cases = []
    for c in golden:
        t1 = time.time()
        cases.append(solve_case(it, c))
        print(f"{c['name']:40s} {time.time() - t1:6.2f} s", flush=True)
    doc = {"generator": "tests/golden/make_reference_text_vectors.py: the reference's ode.nim / utils.nim (Vector arithmetic included) executed by oracle/nim_subset.py",
           "inputs": "tests/golden/ode_golden.json (by case name)", "cases": cases, "steps": step_cases(it),
           "linspace_m10_10_100": hx(it.call("linspace", -10.0, 10.0, 100))}
    json.dump(doc, open(OUT, "w"), indent=0)
    print(f"{len(cases)} solve cases + {len(doc['steps'])} single steps -> {OUT} ({os.path.getsize(OUT)} bytes) in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
