"""Pins the CPU oracle (oracle/ode_oracle.cpp) against every known-answer test the reference holds for the
ODE path — /root/reference/tests/test_ode.nim (all 39 cases; the Tensor[float] cases run through the same
Vector arithmetic: arraymancer element-wise ops are IEEE-identical), the Vector operator cases of
tests/test_vector.nim the steppers rely on, and tests/test_utils.nim's linspace/isClose cases.
The expectations are the reference's own (analytic answer exp(-0.1 t), its tolerances, `t == tspan`)."""
import numpy as np
import pytest

# (test name in test_ode.nim, line, integrator, options kwargs, tol) — scalar `float` state
SCALAR = [
    ("DOPRI54, default", 24, "dopri54", None, 1e-4), ("DOPRI54, tol = 1e-8", 30, "dopri54", "oo", 1e-8),
    ("RK4, default", 36, "rk4", None, 1e-4), ("RK4, dt = 1e-6", 42, "rk4", "oo", 1e-8),
    ("Heun2, default", 48, "heun2", None, 1e-10), ("Heun2, dt = 1e-3", 54, "heun2", None, 1e-8),
    ("Ralston2, default", 60, "ralston2", None, 1e-10), ("Kutta3, default", 66, "kutta3", None, 1e-10),
    ("Heun3, default", 72, "heun3", None, 1e-10), ("Ralston3, default", 78, "ralston3", None, 1e-10),
    ("SSPRK3, default", 84, "ssprk3", None, 1e-10), ("Ralston4, default", 90, "ralston4", None, 1e-10),
    ("Kutta4, default", 96, "kutta4", None, 1e-10), ("RK21, default", 102, "rk21", None, 1e-6),
    ("BS32, default", 108, "bs32", None, 1e-6), ("Tsit54, default", 114, "tsit54", None, 1e-4),
    ("Tsit54, tol = 1e-8", 120, "tsit54", "oo", 1e-8), ("Vern65, default", 126, "vern65", None, 1e-4),
    ("Vern65, tol = 1e-8", 132, "vern65", "oo", 1e-8),
]
# Vector (test_ode.nim:139-197) and Tensor (:199-257) cases: same (integrator, options, tol) list for both
VECTOR = [
    ("DOPRI54", "dopri54", None, 1e-4), ("DOPRI54 tol", "dopri54", "ooV", 1e-8), ("RK4", "rk4", None, 1e-4),
    ("RK4 dt=1e-2", "rk4", "ooV", 1e-8), ("Heun2", "heun2", None, 1e-8), ("Heun2 dt=1e-2", "heun2", "ooV", 1e-5),
    ("Tsit54", "tsit54", None, 1e-4), ("Tsit54 dt=1e-2", "tsit54", "ooV", 1e-8), ("Vern65", "vern65", None, 1e-4),
    ("Vern65 tol", "vern65", "ooV", 1e-8),
]


def _opts(O, key):
    if key is None:
        return O.new_options()
    if key == "oo":
        return O.new_options(relTol=1e-8, dt=1e-6)  # test_ode.nim:9
    return O.new_options(relTol=1e-8, dt=1e-2)      # test_ode.nim:10-11


@pytest.mark.parametrize("name,line,integrator,okey,tol", SCALAR, ids=[s[0] for s in SCALAR])
def test_ode_nim_scalar(oracle, name, line, integrator, okey, tol):
    O = oracle
    tspan = O.linspace(-10.0, 10.0, 100)            # test_ode.nim:15
    correct = np.exp(-0.1 * tspan)                  # test_ode.nim:8,16
    t, y, st = O.solve_ode(O.RHS_LINEAR, [-0.1], 1.0, tspan, _opts(O, okey), integrator)
    assert len(t) == len(tspan) and np.array_equal(t, tspan)      # check t == tspan
    assert len(y) == len(tspan)
    assert np.all(np.abs(y - correct) <= tol)                    # isClose float: abs(a-b) <= tol (utils.nim:270,474-479)


@pytest.mark.parametrize("kind", ["Vector", "Tensor"])
@pytest.mark.parametrize("name,integrator,okey,tol", VECTOR, ids=[s[0] for s in VECTOR])
def test_ode_nim_vector_and_tensor(oracle, kind, name, integrator, okey, tol):
    O = oracle
    tspan = O.linspace(-10.0, 10.0, 100)
    correct = np.exp(-0.1 * tspan)
    t, y, st = O.solve_ode(O.RHS_LINEAR, [-0.1], [1.0, 1.0, 1.0], tspan, _opts(O, okey), integrator)  # test_ode.nim:13-14
    assert np.array_equal(t, tspan)
    assert y.shape == (100, 3)
    if kind == "Vector":   # isClose -> norm2(a-b)/len (utils.nim:252)
        err = np.sqrt(((y - correct[:, None]) ** 2).sum(axis=1)) / 3.0
    else:                  # arraymancer mean_squared_error: mean((a-b)^2)
        err = ((y - correct[:, None]) ** 2).mean(axis=1)
    assert np.all(err <= tol)
    assert np.array_equal(y[:, 0], y[:, 1]) and np.array_equal(y[:, 0], y[:, 2])  # identical components stay identical


def test_scalar_and_vector_paths_agree_bitwise(oracle):
    """float path (size=1, sum=id: ode.nim:54-55) == Vector path of length 1."""
    O = oracle
    ts = O.linspace(-1.0, 2.0, 17)
    for m in O.ALL_ODE:
        _, ys, _ = O.solve_ode(O.RHS_LINEAR, [-0.1], 1.25, ts, integrator=m)
        _, yv, _ = O.solve_ode(O.RHS_LINEAR, [-0.1], [1.25], ts, integrator=m)
        assert np.array_equal(ys, yv[:, 0]), m


def test_invalid_integrator_and_options(oracle):
    O = oracle
    with pytest.raises(ValueError):
        O.solve_ode(O.RHS_NEG_Y, [], 1.0, [0.0, 1.0], integrator="rk5")   # ode.nim:651
    assert O.lib().oracle_integrator_id(b"DoPrI54") == 1                    # toLower (ode.nim:607)
    with pytest.raises(ValueError):
        O.new_options(dtMax=1e-5, dtMin=1e-4)                               # ode.nim:95-96
    with pytest.raises(ValueError):
        O.new_options(scaleMax=0.5)                                         # ode.nim:97-98
    with pytest.raises(ValueError):
        O.new_options(scaleMin=2.0)                                         # ode.nim:99-100
    o = O.new_options(dt=-1e-3, absTol=-1e-5, tStart=-2.0)                   # abs() of all but tStart (ode.nim:101-102)
    assert (o.dt, o.absTol, o.tStart) == (1e-3, 1e-5, -2.0)


# ---- tests/test_vector.nim operator semantics used by the steppers -----------------------------------------
def test_vector_ops(oracle):
    O = oracle
    v1, v2 = [1.1, 2.2, 3.3], [3.3, 2.2, 1.0]
    assert np.array_equal(O.vector_op("+", v1, v2), np.array([1.1 + 3.3, 2.2 + 2.2, 3.3 + 1.0]))  # test_vector.nim:26-30
    assert np.array_equal(O.vector_op("-", v1, v2), np.array([1.1 - 3.3, 2.2 - 2.2, 3.3 - 1.0]))  # :47-52
    assert np.array_equal(O.vector_op("+.", v1, d=8.98), np.array([1.1 + 8.98, 2.2 + 8.98, 3.3 + 8.98]))  # :32-39
    assert np.array_equal(O.vector_op("s*", v1, d=2.5), np.array([1.1 * 2.5, 2.2 * 2.5, 3.3 * 2.5]))
    assert np.array_equal(O.vector_op("abs", [1.0, -2.5, -3.34]), np.array([1.0, 2.5, 3.34]))
    assert np.array_equal(O.vector_op("*.", v1, v2), np.array(v1) * np.array(v2))
    assert np.array_equal(O.vector_op("/.", v1, v2), np.array(v1) / np.array(v2))
    assert O.vector_op("sum", [0.1, 0.2, 0.3])[0] == (0.0 + 0.1 + 0.2) + 0.3   # left-to-right (utils.nim:233-235)
    with pytest.raises(ValueError):                                             # test_vector.nim:41-45
        O.vector_op("+", [1.0, 2.0, 4.0, 1.34, 9.9], [3.3, 2.2, 1.1, 5.67])


# ---- tests/test_utils.nim:15-23 ---------------------------------------------------------------------------------
def test_linspace(oracle):
    O = oracle
    assert list(O.linspace(0.0, 10.0, 11)) == [float(i) for i in range(11)]
    assert list(O.linspace(10.0, 0.0, 11)) == [float(i) for i in range(10, -1, -1)]
    with pytest.raises(ValueError):
        O.linspace(0.0, 1.0, 0)


def test_hermite_spline_endpoints(oracle):
    O = oracle  # utils.nim:273-279: interpolates the end points and derivatives exactly
    assert O.hermite_spline(0.0, 0.0, 2.0, 3.0, 5.0, 1.0, -1.0) == 3.0
    assert O.hermite_spline(2.0, 0.0, 2.0, 3.0, 5.0, 1.0, -1.0) == 5.0
    # cubic reproduction: y = x^3 on [0, 2]
    for x in (0.25, 1.0, 1.75):
        assert abs(O.hermite_spline(x, 0.0, 2.0, 0.0, 8.0, 0.0, 12.0) - x ** 3) < 1e-14


# ---- independent scratch restatement (bit-level cross-check) --------------------------
def test_survey_appendix_b_known_answers(oracle):
    O = oracle
    o = O.new_options(dt=2.0 ** -10)
    for y0, hx in [(1.0, "0x1.81a455c174b97p-2"), (1.0009765625, "0x1.8204bed6e5189p-2"),
                   (1.5009765625, "0x1.216b74dbcfbaep-1"), (1.9990234375, "0x1.81742136bc8b1p-1")]:
        t, y, st = O.solve_ode(O.RHS_NEG_Y, [], y0, [0.0, 0.9765625], o, "rk4")
        assert float(y[-1]).hex() == hx and st.steps == 1000
    t, y, st = O.solve_ode(O.RHS_NEG_Y, [], 1.0, [0.0, 1.0], O.new_options(dt=1e-3), "rk4")
    assert float(y[-1]).hex() == "0x1.78b56362cef86p-2" and st.steps == 1000
    ts = O.linspace(-10.0, 10.0, 100)
    t, y, st = O.solve_ode(O.RHS_LINEAR, [-0.1], 1.0, ts, integrator="rk4")
    assert st.steps == 200002 and y[0] == 2.7182818284617611 and y[99] == 0.36787944117107235
    t, y, st = O.solve_ode(O.RHS_LINEAR, [-0.1], 1.0, ts, integrator="dopri54")
    assert (st.steps, st.rejected) == (2004, 0) and y[0] == 2.7182818284590859 and y[99] == 0.36787944117143523
    t, y, st = O.solve_ode(O.RHS_LINEAR, [-0.1], 1.0, ts, integrator="tsit54")
    assert (st.steps, st.rejected) == (2004, 0) and y[0] == 2.7182818284590935 and y[99] == 0.36787944117143584
    lor = [10.0, 28.0, 8.0 / 3.0]
    tight = O.new_options(absTol=1e-10, relTol=1e-10, dtMin=1e-6, dtMax=1e-1)
    exp = {("dopri54", "d"): (102, (-9.3785717636247341, -8.3570334212910602, 29.362329869304119)),
           ("tsit54", "d"): (102, (-9.378570199590694, -8.3570337492400384, 29.362325823515047)),
           ("dopri54", "t"): (301, (-9.3785700111308472, -8.3570337880787147, 29.362325338335523)),
           ("tsit54", "t"): (279, (-9.3785700108506358, -8.3570337884060368, 29.362325337220842))}
    for (m, k), (steps, vals) in exp.items():
        t, y, st = O.solve_ode(O.RHS_LORENZ, lor, [1.0, 1.0, 1.0], [0.0, 1.0], tight if k == "t" else O.new_options(), m)
        assert st.steps == steps and st.rejected == 0
        assert np.allclose(y[-1], vals, rtol=0, atol=1e-11)   # Appendix B: "<= few-ulp for adaptive"
    y0 = [1 + i / 16 for i in range(16)]
    t, y, st = O.solve_ode(O.RHS_RING, [0.1], y0, [0.0, 1.0], O.new_options(), "tsit54")
    assert st.steps == 102 and abs(y[-1][0] - 1.0413099763026807) < 1e-14 and abs(y[-1][15] - 0.77752377989849464) < 1e-14
    t, y, st = O.solve_ode(O.RHS_RING, [0.1], y0, [0.0, 1.0], tight, "tsit54")
    assert st.steps == 20 and abs(y[-1][0] - 1.0413099763026816) < 1e-14 and abs(y[-1][15] - 0.77752377993260124) < 1e-14


def test_reference_quirks(oracle):
    """SURVEY.md Appendix A quirks the oracle must reproduce."""
    O = oracle
    # A.4: t accumulates in floating point: dt=1e-4 over [0,10] takes 100001 steps
    t, y, st = O.solve_ode(O.RHS_NEG_Y, [], 1.0, [0.0, 10.0], O.new_options(dt=1e-4), "rk4")
    assert st.steps == 100001
    # A.8: requested times strictly inside the last step are dropped -> y shorter than t
    ts = O.linspace(0.0, 1.0, 1001)
    t, y, st = O.solve_ode(O.RHS_NEG_Y, [], 1.0, ts, O.new_options(dt=1e-2), "rk4")
    assert len(t) == 1001 and st.n_y < 1001 and st.n_y >= 990
    # A.8: tspan.len == 2 with both points on one side of tStart -> 2 times, 1 state
    t, y, st = O.solve_ode(O.RHS_NEG_Y, [], 1.0, [1.0, 2.0], O.new_options(dt=1e-2), "rk4")
    assert len(t) == 2 and st.n_y == 1
    # tspan is sorted (ode.nim:609) and tStart inside tspan is emitted verbatim
    t, y, st = O.solve_ode(O.RHS_NEG_Y, [], 3.0, [1.0, -1.0, 0.0], O.new_options(dt=1e-2), "rk4")
    assert list(t) == [-1.0, 0.0, 1.0] and y[1] == 3.0


# ---- tests/test_integrate.nim:72-95: the function-argument forms of cumtrapz / cumsimpson -----------------------------------
@pytest.mark.parametrize("rule,tol", [("trapz", 1e-1), ("simpson", 1e-3)])
@pytest.mark.parametrize("dx", [1e-5, 0.1])
@pytest.mark.parametrize("dim", [0, 3])
def test_cumquad_fn_reference_tests(oracle, rule, tol, dx, dim):
    """"cumtrapz func, discrete points" / "... dx = 0.1" / "cumsimpson func ..." (f = a cos x, a = 2, X = linspace(0, 3pi/2, 17),
    checked against 2 sin x at the reference's tolerances); dim 3 = the Vector flavour of the same integrand."""
    import math
    O = oracle
    X = np.array(O.linspace(0.0, 1.5 * math.pi, 17))
    r = O.cumquad_fn(rule, O.RHS_COS_T, [2.0], dim, X, dx)
    assert r.shape[0] == 17
    exact = 2.0 * np.sin(X)
    if dim == 0:
        assert np.abs(r - exact).max() < tol
    else:
        for c in range(dim):
            assert np.abs(r[:, c] - exact).max() < tol


def test_cumquad_fn_matches_its_discrete_building_blocks(oracle):
    """cumsimpson(f, X, dx) == hermiteInterpolate(X, t, cumsimpson(f(t), t), f(t)) with t the linspace the reference builds
    (integrate.nim:395-400): ties the function form to the already pinned discrete cumsimpson."""
    O = oracle
    X = np.array([0.0, 0.3, 0.55, 1.0])
    dx = 0.05
    n = int(round((X.max() - X.min()) / dx)) + 2
    t = np.array(O.linspace(X.min(), X.max(), n))
    dy = np.array([((0.5 * x + 2.0) * x) * 1.0 - 1.0 for x in t])
    ys = O.cumsimpson(dy, t)
    r = O.cumquad_fn("simpson", O.RHS_POLY_T, [0.5, 2.0, -1.0], 0, X, dx)
    assert len(r) == 4 and r[0] == ys[0] and r[-1] == ys[-1]
    # a query sitting on a grid point reproduces that grid value (h00 = 1, every other weight 0)
    k = int(np.argmin(np.abs(t - 0.55)))
    if t[k] == 0.55:
        assert r[2] == ys[k]


# ---- tests/test_interpolate.nim:5-18, 104-145: newHermiteSpline(t, y, dy) eval / derivEval -----------------------------------
def _arange(x1, x2, dx):
    """utils.nim:480-493 with includeStart = true, includeEnd = false."""
    n = abs(int(np.floor((x2 - x1) / dx)))
    return np.array([x1] + [x1 + float(i) * dx for i in range(1, n + 1)])


def test_hermite_spline_reference_tests(oracle):
    O = oracle
    t = np.array(O.linspace(0.0, 10.0, 100))
    y, dy = np.sin(t), np.cos(t)
    t_test = _arange(0.0, 10.0, 0.2345)
    # "HermiteSpline Eval in input points, direct" / "... for loop" / ".toProc, single value": isClose(val, y[i], 1e-15)
    assert np.all(np.abs(O.hermite_interp(t, y, dy, t) - y) <= 1e-15)
    for x, yy in zip(t, y):
        assert abs(O.hermite_interp(t, y, dy, [x])[0] - yy) <= 1e-15
    # "HermiteSpline Eval between input points": isClose(val, yTest[i], 1e-4)
    assert np.all(np.abs(O.hermite_interp(t, y, dy, t_test) - np.sin(t_test)) <= 1e-4)
    # "HermiteSpline derivEval, single value": abs(res - cos(t[20])) < 1e-9 ; "... seq input": < 1e-5
    assert abs(O.hermite_interp(t, y, dy, [t[20]], deriv=True)[0] - np.cos(t[20])) < 1e-9
    assert np.all(np.abs(O.hermite_interp(t, y, dy, t_test, deriv=True) - np.cos(t_test)) < 1e-5)


# ---- tests/test_integrate.nim:67-70, 82-85: cumtrapz / cumsimpson for discrete points -----------------------------------------
def test_cumulative_quadrature_discrete_reference_tests(oracle):
    import math
    O = oracle
    X = np.array(O.linspace(0.0, 1.5 * math.pi, 17))
    Y = 2.0 * np.cos(X)
    cum = 2.0 * np.sin(X)
    assert np.all(np.abs(O.cumtrapz(Y, X) - cum) <= 1e-1)      # "cumtrapz discrete points"
    assert np.all(np.abs(O.cumsimpson(Y, X) - cum) <= 1e-3)    # "cumsimpson discrete points"
    assert abs(O.cumtrapz(Y, X)[-1] - 2.0 * math.sin(1.5 * math.pi)) <= 1e-1   # "trapz discrete points" (:55-57) = last cumulative value


def test_hermite_spline_without_dy_reference_tests(oracle):
    """tests/test_interpolate.nim:11, 59-103: newHermiteSpline(t, y) estimates the slopes by three-point differences."""
    O = oracle
    t = np.array(O.linspace(0.0, 10.0, 100))
    y = np.sin(t)
    dy = O.hermite_slopes(t, y)
    t_test = _arange(0.0, 10.0, 0.2345)
    assert np.all(np.abs(O.hermite_interp(t, y, dy, t) - y) <= 1e-15)                       # "... Eval in input points"
    assert np.all(np.abs(O.hermite_interp(t, y, dy, t_test) - np.sin(t_test)) <= 1e-4)        # "... Eval between input points"
    assert abs(O.hermite_interp(t, y, dy, [t[20]], deriv=True)[0] - np.cos(t[20])) < 1e-3     # "... derivEval, single value"
    assert np.all(np.abs(O.hermite_interp(t, y, dy, t_test, deriv=True) - np.cos(t_test)) < 2e-3)  # "... derivEval, seq input"
