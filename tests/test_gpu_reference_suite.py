"""The reference's own ODE test file, /root/reference/tests/test_ode.nim (all 39 cases), run through the HIP path: same right-hand
side (f = -0.1 y, :5-7), initial states (:12-14), tspan = linspace(-10, 10, 100) (:15), option sets (:9-11), tolerances and checks
(`t == tspan`, isClose per state type) — one test per reference test, in the reference's order.  On top of the reference's own
(analytic) expectation every case must carry the oracle's bits.  The case table is the one the oracle is pinned with
(tests/test_oracle_reference_kats.py)."""
import numpy as np
import pytest

from test_oracle_reference_kats import SCALAR, VECTOR

pytestmark = pytest.mark.gpu


def _opts(mod, key):
    if key is None:
        return mod.newODEoptions() if hasattr(mod, "newODEoptions") else mod.new_options()
    kw = dict(relTol=1e-8, dt=1e-6) if key == "oo" else dict(relTol=1e-8, dt=1e-2)   # test_ode.nim:9-11
    return mod.newODEoptions(**kw) if hasattr(mod, "newODEoptions") else mod.new_options(**kw)


@pytest.mark.parametrize("name,line,integrator,okey,tol", SCALAR, ids=[s[0] for s in SCALAR])
def test_scalar_state(nn, oracle, dev, name, line, integrator, okey, tol):
    import torch
    O = oracle
    tspan = O.linspace(-10.0, 10.0, 100)
    correct = np.exp(-0.1 * tspan)
    ctx = nn.newNumContext()
    ctx.setF("a", -0.1)
    y0 = torch.tensor([1.0], dtype=torch.float64, device=dev)                      # y0 = 1.0 (:12)
    t, y = nn.solveODE(nn.Rhs.linear(), y0, tspan, _opts(nn, okey), ctx, integrator)
    assert np.array_equal(t, tspan)                                                 # check t == tspan
    got = y[:, 0].cpu().numpy()
    assert len(got) == len(tspan) and np.all(np.abs(got - correct) <= tol)          # isClose(y[i], correct[i], tol)
    rt, ry, st = O.solve_ode(O.RHS_LINEAR, [-0.1], 1.0, tspan, _opts(O, okey), integrator)
    assert np.array_equal(got, np.asarray(ry)), name                                # and the reference's bits


@pytest.mark.parametrize("kind", ["Vector", "Tensor"])
@pytest.mark.parametrize("name,integrator,okey,tol", VECTOR, ids=[s[0] for s in VECTOR])
def test_vector_and_tensor_state(nn, oracle, dev, kind, name, integrator, okey, tol):
    """Vector[float] (:139-197) -> SoA batch of one 3-component system; Tensor[float] (:199-257) -> the AoS layout (one contiguous
    row per system, arraymancer's storage).  isClose: norm2(a - b) / len for Vector (utils.nim:252), mean squared error for Tensor."""
    import torch
    O = oracle
    tspan = O.linspace(-10.0, 10.0, 100)
    correct = np.exp(-0.1 * tspan)
    layout = 0 if kind == "Vector" else 1
    y0 = torch.ones((3, 1) if layout == 0 else (1, 3), dtype=torch.float64, device=dev)   # @[1.0, 1.0, 1.0] (:13-14)
    t, y = nn.solveODE(nn.Rhs.linear(-0.1), y0, tspan, _opts(nn, okey), integrator=integrator, layout=layout)
    assert np.array_equal(t, tspan)
    got = y.cpu().numpy().reshape(100, 3)
    if kind == "Vector":
        err = np.sqrt(((got - correct[:, None]) ** 2).sum(axis=1)) / 3.0
    else:
        err = ((got - correct[:, None]) ** 2).mean(axis=1)
    assert np.all(err <= tol)
    rt, ry, st = O.solve_ode(O.RHS_LINEAR, [-0.1], [1.0, 1.0, 1.0], tspan, _opts(O, okey), integrator)
    assert np.array_equal(got, np.asarray(ry)), (kind, name)
