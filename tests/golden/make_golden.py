#!/usr/bin/env python3
"""Generates tests/golden/ode_golden.json from the CPU oracle (oracle/ode_oracle.cpp).

The reference is Nim (cannot run here) and stores no golden vectors, so these fixtures are produced by the
oracle AFTER it has been pinned to the reference's own known-answer tests
(tests/test_oracle_reference_kats.py).  They freeze inputs + expected outputs (hex floats, bit-exact) so
that (a) the oracle cannot drift silently and (b) the HIP path is checked against data, not only against
a live oracle run.  Re-run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def hx(a):
    return [float(v).hex() for v in np.asarray(a, dtype=np.float64).ravel()]


def case(name, rhs, params, y0_list, tspan, opt_kw, integrator):
    """y0_list: list of IVPs; each a float (scalar path) or list (vector path)."""
    opt = O.new_options(**opt_kw)
    outs = []
    for y0 in y0_list:
        t, y, st = O.solve_ode(rhs, params, y0, tspan, opt, integrator)
        outs.append({"y": hx(y), "n_y": int(st.n_y), "steps": int(st.steps), "rejected": int(st.rejected)})
    return {"name": name, "rhs_kind": rhs, "params": hx(params), "dim": 0 if np.isscalar(y0_list[0]) else len(y0_list[0]),
            "y0": [hx(np.atleast_1d(y)) for y in y0_list], "tspan": hx(tspan), "options": opt_kw, "integrator": integrator,
            "t": hx(t), "ivps": outs}


def main():
    cases = []
    lin = O.linspace(-10.0, 10.0, 100)
    tight = dict(absTol=1e-10, relTol=1e-10, dtMin=1e-6, dtMax=1e-1)
    sc = [1.0, 0.5, -1.25, 1.0009765625]
    for m in ["rk4", "dopri54", "tsit54"]:
        cases.append(case(f"harness_scalar_{m}", O.RHS_LINEAR, [-0.1], sc, lin, dict(dt=1e-2), m))
        cases.append(case(f"harness_vec3_{m}", O.RHS_LINEAR, [-0.1], [[1.0, 1.0, 1.0], [1.0, 2.0, -0.5]], lin, dict(dt=1e-2), m))
        cases.append(case(f"c1_{m}", O.RHS_NEG_Y, [], [1.0 + i * 2.0 ** -10 for i in (0, 1, 513, 1023)], [0.0, 0.9765625],
                          dict(dt=2.0 ** -10), m))
        cases.append(case(f"affine_t_{m}", O.RHS_AFFINE_T, [-0.5, 0.25], [1.0, -2.0], [-1.0, -0.25, 0.0, 0.5, 2.0],
                          dict(dt=1e-3, tStart=0.0), m))
        cases.append(case(f"lorenz_default_{m}", O.RHS_LORENZ, [10.0, 28.0, 8.0 / 3.0],
                          [[1.0 + k * 2.0 ** -20, 1.0, 1.0] for k in (0, 1, 1023)], [0.0, 1.0], dict(dt=1e-3), m))
        cases.append(case(f"vdp_{m}", O.RHS_VANDERPOL, [1.5], [[2.0, 0.0], [0.5, -0.5]], [0.0, 0.5, 1.0, 2.0], dict(dt=1e-3), m))
        cases.append(case(f"ring4_{m}", O.RHS_RING, [0.1], [[1.0, 1.25, 1.5, 1.75]], [0.0, 1.0], dict(dt=1e-3), m))
        cases.append(case(f"ring16_{m}", O.RHS_RING, [0.1], [[1 + i / 16 + s * 2.0 ** -20 for i in range(16)] for s in (0, 7)],
                          [0.0, 1.0], dict(dt=1e-3), m))
        cases.append(case(f"tstart_shift_{m}", O.RHS_LINEAR, [0.3], [1.0, 2.0], [0.5, 1.0, 1.5, 2.5, 3.0],
                          dict(dt=1e-2, tStart=1.5), m))
    # the other 11 integrators (ode.nim:107-178, 191-234, 377-468): a smaller set each
    for m in [x for x in O.ALL_ODE if x not in ("rk4", "dopri54", "tsit54")]:
        cases.append(case(f"harness_scalar_{m}", O.RHS_LINEAR, [-0.1], [1.0, -1.25], lin, dict(dt=1e-2), m))
        cases.append(case(f"harness_vec3_{m}", O.RHS_LINEAR, [-0.1], [[1.0, 2.0, -0.5]], lin, dict(dt=1e-2), m))
        cases.append(case(f"affine_t_{m}", O.RHS_AFFINE_T, [-0.5, 0.25], [1.0, -2.0], [-1.0, -0.25, 0.0, 0.5, 2.0], dict(dt=1e-3), m))
        cases.append(case(f"lorenz_default_{m}", O.RHS_LORENZ, [10.0, 28.0, 8.0 / 3.0], [[1.0, 1.0, 1.0], [1.0 + 2.0 ** -10, 1.0, 1.0]],
                          [0.0, 1.0], dict(dt=1e-3), m))
        cases.append(case(f"ring16_{m}", O.RHS_RING, [0.1], [[1 + i / 16 for i in range(16)]], [0.0, 0.5, 1.0], dict(dt=1e-3), m))
    for m in ["vern65", "bs32", "rk21"]:
        cases.append(case(f"rejecting_vdp_{m}", O.RHS_VANDERPOL, [5.0], [[2.0, 0.0]], [0.0, 20.0],
                          dict(absTol=1e-6, relTol=1e-6, dtMin=1e-9, dtMax=2.0), m))
        cases.append(case(f"dtmin_escape_{m}", O.RHS_LINEAR, [-200.0], [1.0], [0.0, 0.2],
                          dict(absTol=1e-12, relTol=1e-12, dtMin=1e-2, dtMax=1e-1), m))
    for m in ["dopri54", "tsit54"]:
        cases.append(case(f"lorenz_tight_{m}", O.RHS_LORENZ, [10.0, 28.0, 8.0 / 3.0],
                          [[1.0 + k * 2.0 ** -20, 1.0, 1.0] for k in (0, 1023)], [0.0, 1.0], tight, m))
        cases.append(case(f"ring16_tight_{m}", O.RHS_RING, [0.1], [[1 + i / 16 for i in range(16)]], [0.0, 1.0], tight, m))
        # in-step rejections (ode.nim:58-76 retry loop): relaxation oscillator with a large dtMax
        cases.append(case(f"rejecting_vdp_{m}", O.RHS_VANDERPOL, [5.0], [[2.0, 0.0], [1.0, 1.0]], [0.0, 20.0],
                          dict(absTol=1e-6, relTol=1e-6, dtMin=1e-9, dtMax=2.0), m))
        cases.append(case(f"rejecting_lorenz_{m}", O.RHS_LORENZ, [10.0, 28.0, 8.0 / 3.0], [[1.0, 1.0, 1.0]], [0.0, 2.0, 5.0],
                          dict(absTol=1e-6, relTol=1e-6, dtMin=1e-9, dtMax=1.0), m))
        # dtMin double-hit escape (ode.nim:72-74): error stays > 1 but the step is accepted
        cases.append(case(f"dtmin_escape_{m}", O.RHS_LINEAR, [-200.0], [1.0, -3.0], [0.0, 0.2],
                          dict(absTol=1e-12, relTol=1e-12, dtMin=1e-2, dtMax=1e-1), m))
    # reference quirks: dropped dense points inside the last step; 2-point tspan on one side of tStart
    cases.append(case("quirk_dense_tail_rk4", O.RHS_NEG_Y, [], [1.0, 2.0], O.linspace(0.0, 1.0, 101), dict(dt=7e-2), "rk4"))
    cases.append(case("quirk_two_points_one_side_rk4", O.RHS_NEG_Y, [], [1.0], [1.0, 2.0], dict(dt=1e-2), "rk4"))
    cases.append(case("quirk_dense_tail_backward_dopri54", O.RHS_NEG_Y, [], [1.0], O.linspace(-1.0, 0.0, 301), dict(dt=1e-2), "dopri54"))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ode_golden.json")
    json.dump({"generator": "tests/golden/make_golden.py (oracle/ode_oracle.cpp, g++ -O2 -ffp-contract=off)", "cases": cases},
              open(out, "w"), indent=0)
    print(f"{len(cases)} cases -> {out} ({os.path.getsize(out)} bytes)")


if __name__ == "__main__":
    main()
