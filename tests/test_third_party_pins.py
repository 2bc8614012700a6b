"""Pins that do not go through our own transcriptions at all: a third-party Dormand-Prince implementation, the algebraic order
conditions of Runge-Kutta methods, and measured convergence orders on a NONLINEAR problem.

The reference's own known-answer tests integrate y' = -0.1*y (tests/test_ode.nim:5-8): a linear scalar right-hand side satisfies every
order condition that differs only in the shape of its rooted tree, so those tests cannot see most coefficient errors.  These can:
  * SciPy's RK45 *is* the Dormand-Prince 5(4) pair (scipy.integrate._ivp.rk): one step of the oracle's DOPRI54_step on Lorenz must agree with
    scipy's rk_step to a few ulp, the embedded error estimate included (ode.nim:240-305); its RK23 is the Bogacki-Shampine pair (bs32, :212-234): the same;
  * the tableaux the HIP kernels are compiled with (nnhip_ode_tableau_f64) must satisfy all 17 / 37 rooted-tree order conditions of order
    5 / 6, their embedded weights those of order 4 / 5, and the row-sum conditions (ode.nim:240-282, 310-352, 380-443);
  * all 14 methods must converge at their nominal order — 2/3/4/5/6 — on Van der Pol against a 30-digit Taylor-series solution (mpmath), and the
    embedded error estimates must shrink at theirs; through the oracle here, through nnhip_ode_step_batch_f64_dev on the GPU box.
"""
import ctypes as C
import math

import numpy as np
import pytest

LORENZ_P = [10.0, 28.0, 8.0 / 3.0]
ORDER = {"heun2": 2, "ralston2": 2, "kutta3": 3, "heun3": 3, "ralston3": 3, "ssprk3": 3, "ralston4": 4, "kutta4": 4, "rk4": 4,
         "rk21": 2, "bs32": 3, "dopri54": 5, "tsit54": 5, "vern65": 6}                    # ode.nim:608-649 `order =`
EST_ORDER = {"rk21": 2, "bs32": 3, "dopri54": 5, "tsit54": 5, "vern65": 6}                # local size of yNew - yLow: h^(min(p, pHat) + 1)
LOOSE = dict(absTol=1.0, relTol=0.0, dtMin=1e-12, dtMax=10.0)  # error = RMS(error_y) (ode.nim:61-65): far below 1, every step is accepted as given


# ---- SciPy's Dormand-Prince ----------------------------------------------------------------------------------------------------------
def _ulps(a, b):
    return abs(a - b) / np.spacing(max(abs(a), abs(b)))


@pytest.mark.parametrize("t,y,h", [(0.3, [-8.1, -7.9, 27.2], 1e-2), (0.0, [1.0, 1.0, 1.0], 1e-3), (-2.0, [3.0, 4.5, 20.0], 5e-2)])
def test_dopri54_step_agrees_with_scipy_rk45(oracle, t, y, h):
    from scipy.integrate._ivp import rk
    O = oracle
    s, r, b = LORENZ_P

    def f(_t, v):
        return np.array([s * (v[1] - v[0]), v[0] * (r - v[2]) - v[1], v[0] * v[1] - b * v[2]])
    y = np.array(y)
    K = np.empty((rk.RK45.n_stages + 1, 3))
    y_sp, f_sp = rk.rk_step(f, t, y, f(t, y), h, rk.RK45.A, rk.RK45.B, rk.RK45.C, K)
    err_sp = np.dot(K.T, rk.RK45.E) * h                      # = yLow - yNew
    yN, fs, dtU, err = O.step(O.RHS_LORENZ, LORENZ_P, "dopri54", O.new_options(**LOOSE), t, y, f(t, y), h)
    assert dtU == h
    for c in range(3):
        assert _ulps(yN[c], y_sp[c]) <= 4, (c, yN[c], y_sp[c])
        assert _ulps(fs[c], f_sp[c]) <= 16                   # FSAL = f(t + h, yNew): inherits yNew's ulps times the local Lipschitz factor
    # error = sqrt(1/3 * sum((error_y / 1.0)^2)) with error_y = yNew - yLow (ode.nim:61-65, 303); yNew - yLow cancels ~1e5-fold, so the
    # comparison is relative to the rounding of the terms that cancel: 64 ulp of |dt * f|
    want = math.sqrt(float(np.mean(err_sp ** 2)))
    assert abs(err - want) <= 64 * np.spacing(h * np.abs(f(t, y)).max()), (err, want)
    assert err > 0 and abs(err - want) <= 1e-6 * want + 64 * np.spacing(h * np.abs(f(t, y)).max())


@pytest.mark.parametrize("t,y,h", [(0.3, [-8.1, -7.9, 27.2], 1e-2), (0.0, [1.0, 1.0, 1.0], 1e-3), (-2.0, [3.0, 4.5, 20.0], 5e-2)])
def test_bs32_step_agrees_with_scipy_rk23(oracle, t, y, h):
    """SciPy's RK23 is the Bogacki-Shampine 3(2) pair the reference calls bs32 (ode.nim:212-234): same stages, same weights, the same embedded estimate."""
    from scipy.integrate._ivp import rk
    O = oracle
    s, r, b = LORENZ_P

    def f(_t, v):
        return np.array([s * (v[1] - v[0]), v[0] * (r - v[2]) - v[1], v[0] * v[1] - b * v[2]])
    y = np.array(y)
    K = np.empty((rk.RK23.n_stages + 1, 3))
    y_sp, f_sp = rk.rk_step(f, t, y, f(t, y), h, rk.RK23.A, rk.RK23.B, rk.RK23.C, K)
    err_sp = np.dot(K.T, rk.RK23.E) * h
    yN, fs, dtU, err = O.step(O.RHS_LORENZ, LORENZ_P, "bs32", O.new_options(**LOOSE), t, y, f(t, y), h)
    assert dtU == h
    for c in range(3):
        assert _ulps(yN[c], y_sp[c]) <= 4, (c, yN[c], y_sp[c])
        assert _ulps(fs[c], f_sp[c]) <= 16
    want = math.sqrt(float(np.mean(err_sp ** 2)))
    assert err > 0 and abs(err - want) <= 1e-6 * want + 64 * np.spacing(h * np.abs(f(t, y)).max()), (err, want)


def test_dopri54_tableau_is_scipys(nn):
    """The table the HIP kernels read against scipy's RK45.A / B / C / E, entry by entry (<= 1 ulp: scipy stores the same fractions)."""
    from scipy.integrate._ivp import rk
    tab = _tableau(nn, "dopri54")
    for s_ in range(6):
        assert _ulps(tab["c"][s_], rk.RK45.C[s_]) <= 1 or tab["c"][s_] == rk.RK45.C[s_]
        for j in range(s_):
            assert tab["A"][s_, j] == rk.RK45.A[s_, j] or _ulps(tab["A"][s_, j], rk.RK45.A[s_, j]) <= 1
    for j in range(6):
        assert tab["b"][j] == rk.RK45.B[j] or _ulps(tab["b"][j], rk.RK45.B[j]) <= 1
        assert tab["A"][6, j] == tab["b"][j]                 # b_i = a_7i (ode.nim:269-274)
    e = np.array(list(tab["bhat"])) - np.append(tab["b"], 0.0)
    assert np.abs(e - rk.RK45.E).max() <= 1e-16


# ---- rooted-tree order conditions ------------------------------------------------------------------------------------------------------
def _tableau(nn, integrator):
    L = nn._lib.lib()
    out = np.full(128, np.nan)
    k = L.nnhip_ode_tableau_f64(nn.ode.integrator_id(integrator), -1, out.ctypes.data_as(C.POINTER(C.c_double)), 128)
    assert k > 0
    S, NB = int(out[0]), int(out[1])
    p = 2
    c = out[p:p + S].copy(); p += S
    A = np.zeros((S, S))
    for s in range(1, S):
        A[s, :s] = out[p:p + s]; p += s
    b = out[p:p + NB].copy(); p += NB
    bhat = out[p:p + S].copy(); p += S
    return dict(S=S, c=c, A=A, b=b, bhat=bhat)


def _trees(order):
    """All rooted trees with `order` vertices as sorted tuples of subtrees (the empty tuple is the single vertex)."""
    if order == 1:
        return [()]
    out = set()

    def parts(n, maxpart):
        if n == 0:
            yield (); return
        for k in range(min(n, maxpart), 0, -1):
            for rest in parts(n - k, k):
                yield (k,) + rest
    from itertools import product
    for part in parts(order - 1, order - 1):
        for combo in product(*[_trees(k) for k in part]):
            out.add(tuple(sorted(combo)))
    return sorted(out)


def _gamma(t):
    size = lambda u: 1 + sum(size(v) for v in u)
    g = size(t)
    for u in t:
        g *= _gamma(u)
    return g


def _phi(t, A):
    v = np.ones(A.shape[0])
    for u in t:
        v = v * (A @ _phi(u, A))
    return v


def test_tree_counts():
    assert [len(_trees(k)) for k in range(1, 7)] == [1, 1, 2, 4, 9, 20]


@pytest.mark.parametrize("integrator,order,emb_order", [("dopri54", 5, 4), ("tsit54", 5, 4), ("vern65", 6, 5)])
def test_device_tableaux_satisfy_the_order_conditions(nn, integrator, order, emb_order):
    T = _tableau(nn, integrator)
    S, A = T["S"], T["A"]
    # the literals are printed to 13-16 significant digits (ode.nim:310-352, 380-443): residuals are at that level, times the size of the weights
    # (Vern65's a_9j / b_j reach 176 and cancel).  DOPRI54's fractions are exact to an ulp.
    tol = {"dopri54": 5e-16, "tsit54": 2e-14, "vern65": 1e-11}[integrator]
    assert np.abs(A.sum(axis=1) - T["c"]).max() <= tol * 10, "row sums: c_i = sum_j a_ij"
    b = np.zeros(S); b[:len(T["b"])] = T["b"]
    worst = 0.0
    for k in range(1, order + 1):
        for t in _trees(k):
            worst = max(worst, abs(b @ _phi(t, A) - 1.0 / _gamma(t)))
    assert worst <= tol * 50, f"b fails an order-{order} condition by {worst:.3g}"
    worst = 0.0
    for k in range(1, emb_order + 1):
        for t in _trees(k):
            worst = max(worst, abs(T["bhat"] @ _phi(t, A) - 1.0 / _gamma(t)))
    # Tsit54's bHat_i are the DIFFERENCES b_i - bLow_i (error_y = dt * sum bHat_i k_i, ode.nim:372): they must annihilate every tree up to order 4
    if integrator == "tsit54":
        worst = 0.0
        for k in range(1, emb_order + 1):
            for t in _trees(k):
                worst = max(worst, abs(T["bhat"] @ _phi(t, A)))
    # bHat is printed to 12-15 digits only (e.g. -0.001780011052226, 0.04909967648382): that is the accuracy of the embedded solution
    assert worst <= {"dopri54": 1e-15, "tsit54": 5e-15, "vern65": 1e-11}[integrator], f"bHat fails an order-{emb_order} condition by {worst:.3g}"
    # and the next order is NOT satisfied: the pair really is p(p-1), not a lucky higher one
    miss = max(abs(b @ _phi(t, A) - 1.0 / _gamma(t)) for t in _trees(order + 1))
    assert miss > 1e-6


# ---- measured convergence on Van der Pol (mu = 1.5) against a 30-digit Taylor-series solution ----------------------------------------------
MU, T_END, Y0 = 1.5, 1.0, (2.0, 0.5)


@pytest.fixture(scope="module")
def vdp_exact():
    import mpmath
    with mpmath.workdps(30):
        sol = mpmath.odefun(lambda t, y: [y[1], MU * ((1 - y[0] * y[0]) * y[1]) - y[0]], 0, [mpmath.mpf(Y0[0]), mpmath.mpf(Y0[1])])
        v = sol(mpmath.mpf(T_END))
        return np.array([float(v[0]), float(v[1])])


def _steps_for(integrator):  # step counts whose errors sit between the method's asymptotic regime and round-off
    return {2: (200, 400, 800, 1600), 3: (100, 200, 400, 800), 4: (25, 50, 100, 200), 5: (8, 16, 32, 64), 6: (4, 8, 16, 32)}[ORDER[integrator]]


def _slopes(hs, errs):
    return [math.log(errs[i] / errs[i + 1]) / math.log(hs[i] / hs[i + 1]) for i in range(len(hs) - 1)]


def _check_order(integrator, hs, errs):
    """Orders 2-4: the measured slope is the nominal order +- 0.15 (first interval +- 0.25).  Orders 5-6: DOPRI54, Tsit54 and Vern65 are
    constructed with minimised principal error coefficients, so between the pre-asymptotic regime and double-precision round-off (their
    errors reach 1e-12 within a few halvings) the higher-order terms still dominate: measured slopes come out ABOVE the nominal order (5.5-10); required is
    slope >= p - 0.3 — a wrong coefficient costs whole orders.  The exact statement for them is algebraic: test_device_tableaux_satisfy_the_order_conditions."""
    p = ORDER[integrator]
    sl = _slopes(hs, errs)
    used = 0
    for k, s in enumerate(sl):
        if errs[k + 1] < 3e-12:  # too close to round-off (and to the accuracy of the reference's 14-digit Vern65 literals) to measure a slope
            continue
        used += 1
        if p <= 4:
            assert abs(s - p) <= (0.25 if k == 0 else 0.15), (integrator, sl, errs)
        else:
            assert s >= p - 0.3, (integrator, sl, errs)
    assert used >= (3 if p <= 4 else 2), (integrator, sl, errs)


def _oracle_march(O, integrator, n):
    opt = O.new_options(**LOOSE)
    h = T_END / n
    y = np.array(Y0)
    f = O.rhs(O.RHS_VANDERPOL, [MU], 0.0, y)
    t = 0.0
    for _ in range(n):
        y, f, dtU, err = O.step(O.RHS_VANDERPOL, [MU], integrator, opt, t, y, f, h)
        assert dtU == h
        if integrator not in ("dopri54", "tsit54", "vern65", "bs32"):
            f = y  # unused FSAL slot of the non-FSAL methods
        t += h
    return y


@pytest.mark.parametrize("integrator", sorted(ORDER))
def test_oracle_converges_at_the_nominal_order(oracle, vdp_exact, integrator):
    ns = _steps_for(integrator)
    errs = [float(np.abs(_oracle_march(oracle, integrator, n) - vdp_exact).max()) for n in ns]
    _check_order(integrator, [T_END / n for n in ns], errs)


@pytest.mark.parametrize("integrator", sorted(EST_ORDER))
def test_oracle_error_estimate_shrinks_at_its_order(oracle, integrator):
    """error (ode.nim:61-65) of one step from the same state with h halved: RMS(yNew - yLow) = O(h^(q+1)), q the lower order of the pair
    (Vern65 compares its order-6 and order-5 solutions: h^6)."""
    O = oracle
    opt = O.new_options(**LOOSE)
    y = np.array(Y0)
    f = O.rhs(O.RHS_VANDERPOL, [MU], 0.0, y)
    hs = [0.1 * 2.0 ** -k for k in range(1, 5)] if EST_ORDER[integrator] >= 5 else [0.02 * 2.0 ** -k for k in range(1, 5)]
    errs = [O.step(O.RHS_VANDERPOL, [MU], integrator, opt, 0.0, y, f, h)[3] for h in hs]
    sl = _slopes(hs, errs)
    assert min(errs) > 1e-14
    assert all(abs(s - EST_ORDER[integrator]) <= 0.2 for s in sl), (integrator, sl, errs)


@pytest.mark.gpu
@pytest.mark.parametrize("integrator", sorted(ORDER))
def test_hip_converges_at_the_nominal_order(nn, dev, vdp_exact, integrator):
    """The same measurement through nnhip_ode_step_batch_f64_dev: all four step sizes march side by side, one IVP each (per-IVP dt)."""
    import torch
    ns = _steps_for(integrator)
    opt = nn.newODEoptions(**LOOSE)
    rhs = nn.Rhs.vanderpol(MU)
    errs = []
    for n in ns:
        h = T_END / n
        y = torch.tensor([[Y0[0]] * 2, [Y0[1]] * 2], dtype=torch.float64, device=dev)
        f = nn.rhsBatch(rhs, 0.0, y) if hasattr(nn, "rhsBatch") else None
        if f is None:
            f = torch.stack([y[1], MU * ((1.0 - y[0] * y[0]) * y[1]) - y[0]])
        t = 0.0
        for _ in range(n):
            y, fs, dtU, err = nn.integratorStep(rhs, t, y, f, h, opt, integrator=integrator)
            if fs is not None and integrator in ("dopri54", "tsit54", "vern65", "bs32"):
                f = fs
            t += h
        errs.append(float(np.abs(y[:, 0].cpu().numpy() - vdp_exact).max()))
    _check_order(integrator, [T_END / n for n in ns], errs)


@pytest.mark.gpu
@pytest.mark.parametrize("integrator", sorted(EST_ORDER))
def test_hip_error_estimate_shrinks_at_its_order(nn, dev, integrator):
    import torch
    hs = [0.1 * 2.0 ** -k for k in range(1, 5)] if EST_ORDER[integrator] >= 5 else [0.02 * 2.0 ** -k for k in range(1, 5)]
    y = torch.tensor([[Y0[0]] * 4, [Y0[1]] * 4], dtype=torch.float64, device=dev)
    f = torch.stack([y[1], MU * ((1.0 - y[0] * y[0]) * y[1]) - y[0]])
    dt = torch.tensor(hs, dtype=torch.float64, device=dev)
    _y, _f, dtU, err = nn.integratorStep(nn.Rhs.vanderpol(MU), 0.0, y, f, dt, nn.newODEoptions(**LOOSE), integrator=integrator)
    assert torch.equal(dtU, dt)
    sl = _slopes(hs, [float(e) for e in err])
    assert all(abs(s - EST_ORDER[integrator]) <= 0.2 for s in sl), (integrator, sl)
