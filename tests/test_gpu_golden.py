"""HIP path vs the committed golden fixtures (tests/golden/ode_golden.json) and vs a live oracle run.
Tolerances (BASELINE.json north_star): fixed-step 1e-10 abs, adaptive 1e-6 abs.  We require more: bit equality for
fixed-step AND adaptive runs (the controller's pow is glibc's, bit for bit: glibc_pow.hpp) and identical accepted/rejected
step counts."""
import numpy as np
import pytest

import json
import os

from golden_util import fh, load_cases

# what the REFERENCE'S OWN TEXT returns on the same inputs (tests/golden/make_reference_text_vectors.py; oracle/nim_subset.py)
_REFTEXT = {c["name"]: c for c in json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_text_vectors.json")))["cases"]}

pytestmark = pytest.mark.gpu
TOL_FIXED, TOL_ADAPTIVE = 1e-10, 1e-6
FIXED = {"rk4", "heun2", "ralston2", "kutta3", "heun3", "ralston3", "ssprk3", "ralston4", "kutta4"}
KEYS = {1: ("a",), 2: ("sigma", "rho", "beta"), 3: ("c",), 4: ("a", "b"), 5: ("mu",)}


def _rhs(nn, kind, params):
    return nn.Rhs(kind, KEYS.get(kind, ()), dict(zip(KEYS.get(kind, ()), params)))


@pytest.mark.parametrize("layout", [0, 1], ids=["soa", "aos"])
@pytest.mark.parametrize("case", load_cases(), ids=lambda c: c["name"])
def test_hip_matches_golden(nn, dev, case, layout):
    import torch
    dim = max(case["dim"], 1)
    integ = nn._lib.lib().nnhip_ode_integrator_id(case["integrator"].encode())
    if not nn._lib.lib().nnhip_ode_supported(integ, case["rhs_kind"], dim, layout, 0):
        pytest.skip("no kernel for this (integrator, rhs, dim) yet")
    if case["dim"] == 0 and layout == 1:
        pytest.skip("scalar states have one layout")
    n = len(case["y0"])
    y0 = np.stack([fh(y) for y in case["y0"]])  # [n, dim]
    if case["dim"] == 0:
        y0t = torch.from_numpy(y0[:, 0].copy()).to(dev)
    elif layout == 0:
        y0t = torch.from_numpy(np.ascontiguousarray(y0.T)).to(dev)  # [dim, n]
    else:
        y0t = torch.from_numpy(y0.copy()).to(dev)                   # [n, dim]
    opt = nn.newODEoptions(**case["options"])
    t, y, cnt = nn.solveODE(_rhs(nn, case["rhs_kind"], fh(case["params"])), y0t, fh(case["tspan"]), opt,
                            integrator=case["integrator"], layout=layout, return_counts=True)
    assert np.array_equal(t, fh(case["t"]))
    reftext = _REFTEXT[case["name"]]
    assert [float(v).hex() for v in t] == reftext["t"]
    got = y.cpu().numpy()
    ny, steps, rej = (cnt[k].cpu().numpy() for k in ("ny", "steps", "rejected"))
    for i, exp in enumerate(case["ivps"]):
        if case["dim"] == 0:
            g = got[:, i]
        elif layout == 0:
            g = got[:, :, i]
        else:
            g = got[:, i, :]
        want = fh(exp["y"]).reshape(exp["n_y"], -1)
        assert ny[i] == exp["n_y"]
        gi = g.reshape(len(t), -1)
        assert np.isnan(gi[exp["n_y"]:]).all()
        gi = gi[:exp["n_y"]]
        # the HIP path against the reference's text directly (no oracle in between): rows returned and every bit of them
        assert reftext["ivps"][i]["n_y"] == ny[i] and [float(v).hex() for v in gi.ravel()] == reftext["ivps"][i]["y"], "differs from the reference's text"
        if case["integrator"] in FIXED:
            assert np.abs(gi - want).max() <= TOL_FIXED
            assert np.array_equal(gi, want), "fixed-step results must be bit-exact"
            assert steps[i] == exp["steps"]
        else:
            assert np.abs(gi - want).max() <= TOL_ADAPTIVE
            assert np.array_equal(gi, want), "adaptive results must be bit-exact too"
            assert (steps[i], rej[i]) == (exp["steps"], exp["rejected"])
