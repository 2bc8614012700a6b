"""Pin of the oracle.  The reference cannot be executed here (Nim, no toolchain) and stores no golden vectors, so
oracle/ode_oracle.cpp is pinned by (i) the reference's analytic known-answer tests (tests/test_oracle_reference_kats.py) and
(ii) THIS file: a second restatement written independently from ode.nim / utils.nim in plain Python floats
(oracle/py_restatement.py) must agree with the C++ one BIT FOR BIT — all 14 integrators, scalar and Vector states, both
directions, dense output, default and tight options, the reference's harness.  Two separately written restatements that agree
to the last bit on every operation order, controller decision and emitted row leave little room for a shared misreading."""
import numpy as np
import pytest

from oracle import py_restatement as P

ALL = ["heun2", "ralston2", "kutta3", "heun3", "ralston3", "ssprk3", "ralston4", "kutta4", "rk4", "rk21", "bs32", "dopri54", "tsit54", "vern65"]
LOR = [10.0, 28.0, 8.0 / 3.0]


def lorenz(t, y):
    return [LOR[0] * (y[1] - y[0]), y[0] * (LOR[1] - y[2]) - y[1], y[0] * y[1] - LOR[2] * y[2]]


def vdp(t, y):
    return [y[1], 1.5 * ((1.0 - y[0] * y[0]) * y[1]) - y[0]]


@pytest.mark.parametrize("integrator", ALL)
def test_reference_harness_scalar(oracle, integrator):
    """tests/test_ode.nim:5-46: f = -0.1 y, y0 = 1, tspan = linspace(-10, 10, 100), default options and (relTol 1e-8, dt 1e-2)."""
    O = oracle
    ts = list(O.linspace(-10.0, 10.0, 100))
    for kw in ({}, dict(relTol=1e-8, dt=1e-2)):
        if integrator in ("rk4", "heun2", "ralston2", "kutta3", "heun3", "ralston3", "ssprk3", "ralston4", "kutta4") and not kw:
            kw = dict(dt=1e-3)   # the default dt = 1e-4 means 2e5 pure-Python steps per direction: keep the CPU suite quick
        t, y, steps = P.solve_ode(lambda t, y: y * -0.1, 1.0, ts, P.new_options(**kw), integrator)
        rt, ry, st = O.solve_ode(O.RHS_LINEAR, [-0.1], 1.0, ts, O.new_options(**kw), integrator)
        assert t == list(rt) and len(y) == st.n_y == 100
        assert np.array_equal(np.array(y), np.asarray(ry)), (integrator, kw)
        assert steps == st.steps


@pytest.mark.parametrize("integrator", ALL)
def test_vector_states_tight_and_default(oracle, integrator):
    """Lorenz (3 components) and Van der Pol (2): forward + backward, dense rows incl. times closer than the steps (rows the
    reference drops), tight tolerances with rejected steps."""
    O = oracle
    for f, kind, params, y0, ts in ((lorenz, O.RHS_LORENZ, LOR, [1.0, 1.0, 20.0], [-0.05, 0.0, 0.1, 0.10001, 0.3]),
                                    (vdp, O.RHS_VANDERPOL, [1.5], [2.0, 0.0], [0.5, -0.25, 0.75, 0.2]),
                                    (lorenz, O.RHS_LORENZ, LOR, [1.0, 1.0, 1.0], [0.0, 0.25])):
        for kw in (dict(dt=2.0 ** -7), dict(dt=2.0 ** -7, absTol=1e-9, relTol=1e-9, dtMin=1e-6, dtMax=1e-1, tStart=0.05)):
            t, y, steps = P.solve_ode(f, y0, ts, P.new_options(**kw), integrator)
            rt, ry, st = O.solve_ode(kind, params, y0, ts, O.new_options(**kw), integrator)
            assert t == list(rt) and len(y) == st.n_y
            assert np.array_equal(np.array(y).reshape(st.n_y, -1), np.asarray(ry).reshape(st.n_y, -1)), (integrator, kw, ts)
            assert steps == st.steps


def test_survey_appendix_b_values():
    """The hex values SURVEY.md Appendix B lists (computed independently of both restatements in the survey session)."""
    t, y, steps = P.solve_ode(lambda t, y: -y, 1.0, [0.0, 0.9765625], P.new_options(dt=2.0 ** -10), "rk4")
    assert steps == 1000 and y[-1] == float.fromhex("0x1.81a455c174b97p-2")
    t, y, steps = P.solve_ode(lambda t, y: -y, 1.0, [0.0, 1.0], P.new_options(dt=1e-3), "rk4")
    assert steps == 1000 and y[-1] == float.fromhex("0x1.78b56362cef86p-2")
    t, y, steps = P.solve_ode(lorenz, [1.0, 1.0, 1.0], [0.0, 1.0], P.new_options(), "dopri54")
    assert steps == 102 and y[-1] == [-9.3785717636247341, -8.3570334212910602, 29.362329869304119]
    t, y, steps = P.solve_ode(lorenz, [1.0, 1.0, 1.0], [0.0, 1.0], P.new_options(absTol=1e-10, relTol=1e-10, dtMin=1e-6, dtMax=1e-1), "tsit54")
    assert steps == 279 and y[-1] == [-9.3785700108506358, -8.3570337884060368, 29.362325337220842]
