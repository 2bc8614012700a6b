"""§8 f4 consumer: batched newHermiteSpline(t, y, dy).eval / .derivEval (interpolate.nim:186-240, 299-390) on the
solver's trajectory tensor vs the oracle's restatement, bit-exact, every ExtrapolateKind."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_hermite_spline_on_trajectories(nn, oracle, dev):
    import torch
    O = oracle
    n = 200
    rng = np.random.default_rng(9)
    y0 = np.stack([1.0 + rng.uniform(0, 1, n), np.ones(n), np.ones(n)])
    ts = O.linspace(0.0, 1.0, 21)
    f = nn.Rhs.lorenz()
    t, y = nn.solveODE(f, torch.from_numpy(y0).to(dev), ts, nn.newODEoptions(dt=1e-3), integrator="rk4")
    dy = torch.stack([nn.rhsBatch(f, t[j], y[j]) for j in range(len(t))])          # the README's (t, y, dy) recipe
    spl = nn.newHermiteSpline(t, y, dy)
    xq = np.concatenate([rng.uniform(-0.2, 1.2, 40), [0.0, 1.0, 0.5, t[3], -0.5, 1.5]])
    Yh, dYh = y.cpu().numpy().reshape(len(t), -1), dy.cpu().numpy().reshape(len(t), -1)
    for deriv in (False, True):
        for extrap, val in (("Native", None), ("Edge", None), ("Linear", None), ("Constant", 7.25)):
            got = (spl.derivEval if deriv else spl.eval)(xq, extrap=extrap, extrapValue=val).cpu().numpy().reshape(len(xq), -1)
            for m in range(0, Yh.shape[1], 37):
                ref = O.hermite_interp(t, Yh[:, m], dYh[:, m], xq, deriv=deriv, extrap=extrap, extrap_value=val or 0.0)
                assert np.array_equal(got[:, m], ref), (deriv, extrap, m)
    # knots are reproduced exactly; Error raises like the reference's ValueError
    assert torch.equal(spl.eval(t), y)
    with pytest.raises(ValueError):
        spl.eval([1.5], extrap="Error")
    # interpolation error of the cubic Hermite spline against a fine-grid solve is small
    tf, yf = nn.solveODE(f, torch.from_numpy(y0).to(dev), [0.0, 0.525], nn.newODEoptions(dt=1e-3), integrator="rk4")
    assert float((spl.eval(0.525) - yf[-1]).abs().max()) < 2e-2  # h^4 error of a cubic Hermite with knot spacing 0.05 on Lorenz


def test_cumtrapz_on_trajectories(nn, oracle, dev):
    """cumtrapz(Y, X) / trapz (integrate.nim:104-135) over a trajectory tensor, bit-exact vs the oracle; > one weight chunk."""
    import torch
    O = oracle
    n = 150
    rng = np.random.default_rng(10)
    y0 = rng.uniform(0.5, 2.0, n)
    ts = np.sort(np.unique(np.round(rng.uniform(0.0, 2.0, 500), 4)))
    ts[0] = 0.0
    t, y = nn.solveODE(nn.Rhs.linear(-0.8), torch.from_numpy(y0).to(dev), ts, nn.newODEoptions(dt=1e-3), integrator="rk4")
    assert len(t) > 400
    c = nn.cumtrapz(y, t).cpu().numpy()
    yh = y.cpu().numpy()
    for m in range(0, n, 13):
        assert np.array_equal(c[:, m], O.cumtrapz(yh[:, m], t))
    assert np.array_equal(nn.trapz(y, t).cpu().numpy(), c[-1])
    exact = y0 * (1 - np.exp(-0.8 * t[-1])) / 0.8
    assert np.abs(c[-1] - exact).max() < 1e-4


@pytest.mark.parametrize("n_t", [3, 4, 5, 10, 11, 200, 201])
def test_cumsimpson_on_trajectories(nn, oracle, dev, n_t):
    """cumsimpson(Y, X) (integrate.nim:329-375) over a trajectory tensor, odd and even point counts, non-uniform grid,
    bit-exact vs the oracle."""
    import torch
    O = oracle
    n = 90
    rng = np.random.default_rng(n_t)
    y0 = rng.uniform(0.5, 2.0, n)
    ts = np.sort(np.unique(np.round(rng.uniform(0.0, 2.0, 3 * n_t), 5)))[:n_t]
    ts[0] = 0.0
    assert len(ts) == n_t
    t, y = nn.solveODE(nn.Rhs.linear(-0.8), torch.from_numpy(y0).to(dev), ts, nn.newODEoptions(dt=1e-3), integrator="rk4")
    c = nn.cumsimpson(y, t).cpu().numpy()
    yh = y.cpu().numpy()
    for m in range(0, n, 7):
        # (rows the solver dropped — reference quirk, NaN — propagate identically through both)
        assert np.array_equal(c[:, m], O.cumsimpson(yh[:, m], t), equal_nan=True), (n_t, m)
    if n_t >= 200 and np.isfinite(c[-1]).all():
        assert np.abs(c[-1] - y0 * (1 - np.exp(-0.8 * t[-1])) / 0.8).max() < 1e-3  # random non-uniform grid: loose sanity bound only
    with pytest.raises(ValueError):
        nn.cumsimpson(y[:2], t[:2])


def test_hermite_spline_without_dy(nn, oracle, dev):
    """newHermiteSpline(X, Y) (interpolate.nim:241-257): the slope estimates and everything evaluated from them are bit-identical to
    the oracle's; non-uniform knots, several series."""
    import torch
    O = oracle
    rng = np.random.default_rng(4)
    X = np.cumsum(0.05 + rng.random(60))
    Yh = np.stack([np.sin(X) * (1 + k) + 0.1 * k * X for k in range(5)], axis=1)  # [knots, series]
    Y = torch.from_numpy(Yh).to(dev)
    sp = nn.newHermiteSpline(X, Y)
    xq = np.concatenate([X[[0, 7, -1]], X[0] + (X[-1] - X[0]) * rng.random(40)])
    ev, dv = sp.eval(xq).cpu().numpy(), sp.derivEval(xq).cpu().numpy()
    for m in range(5):
        slopes = O.hermite_slopes(X, Yh[:, m])
        assert np.array_equal(sp.dY.cpu().numpy()[:, m], slopes)
        assert np.array_equal(ev[:, m], O.hermite_interp(X, Yh[:, m], slopes, xq))
        assert np.array_equal(dv[:, m], O.hermite_interp(X, Yh[:, m], slopes, xq, deriv=True))
    with pytest.raises(ValueError):
        nn.newHermiteSpline(X[:1], Y[:1])


def test_host_pointer_forms_equal_the_device_forms(nn, dev):
    """The consumers called with host arrays (what the Nim shim would do) stage through the device and return the same bits:
    HermiteSpline with and without dY, cumtrapz, cumsimpson, and the function forms with a host parameter sweep."""
    import torch
    rng = np.random.default_rng(12)
    X = np.cumsum(0.05 + rng.random(41))
    Yh = np.stack([np.cos(X) * (1 + k) for k in range(6)], axis=1)
    dYh = np.stack([-np.sin(X) * (1 + k) for k in range(6)], axis=1)
    Y, dY = torch.from_numpy(Yh).to(dev), torch.from_numpy(dYh).to(dev)
    xq = X[0] + (X[-1] - X[0]) * rng.random(30)
    for args_h, args_d in (((Yh, dYh), (Y, dY)), ((Yh,), (Y,))):
        sh, sd = nn.newHermiteSpline(X, *args_h), nn.newHermiteSpline(X, *args_d)
        assert isinstance(sh.eval(xq), np.ndarray)
        assert np.array_equal(sh.eval(xq), sd.eval(xq).cpu().numpy()) and np.array_equal(sh.derivEval(xq), sd.derivEval(xq).cpu().numpy())
        assert np.array_equal(sh.eval(X[-1] + 1.0, extrap="Linear"), sd.eval(X[-1] + 1.0, extrap="Linear").cpu().numpy())
    assert np.array_equal(nn.cumtrapz(Yh, X), nn.cumtrapz(Y, X).cpu().numpy())
    assert np.array_equal(nn.cumsimpson(Yh, X), nn.cumsimpson(Y, X).cpu().numpy())
    f = nn.Rhs.custom(2, "dy[0] = p[0] * t * t; dy[1] = p[1] * (1.0 - t);", keys=("a", "b"), defaults={"a": 0.0, "b": 0.0}, name="host_forms")
    sw = rng.uniform(-1, 1, (2, 500))
    Xq = np.array([0.0, 0.4, 0.1, 1.0])
    for fn in (nn.cumtrapz, nn.cumsimpson):
        h = fn(f, Xq, dx=1e-2, sweep=sw)
        d = fn(f, Xq, dx=1e-2, sweep=torch.from_numpy(sw).to(dev))
        assert isinstance(h, np.ndarray) and np.array_equal(h, d.cpu().numpy())
        assert np.array_equal(fn(f, Xq, dx=1e-2, n=3, device="host", ctx=nn.newNumContext({"a": 0.5, "b": 2.0})),
                              fn(f, Xq, dx=1e-2, n=3, ctx=nn.newNumContext({"a": 0.5, "b": 2.0})).cpu().numpy())
    with pytest.raises(ValueError):
        nn.cumsimpson(Yh[:2], X[:2])


def test_descending_abscissae_are_sorted_like_the_reference(nn, oracle, dev):
    """Until round 5 these calls were refused (the product required strictly ascending X); the reference sorts and trims first (integrate.nim:131, 340;
    interpolate.nim:231, 244), and since round 6 so does the backend: a trajectory handed over with its time axis reversed — device entries and host-pointer
    entries — against the oracle, which is fed the same reversed arrays.  (A first-contact test: tests/conftest.py runs it after the recorded ones.)"""
    import torch
    O = oracle
    rng = np.random.default_rng(21)
    X = np.cumsum(0.05 + rng.random(41))
    Yh = np.stack([np.cos(X) * (1 + k) for k in range(6)], axis=1)
    dYh = np.stack([-np.sin(X) * (1 + k) for k in range(6)], axis=1)
    Xr = X[::-1].copy()
    Y, dY = torch.from_numpy(Yh).to(dev), torch.from_numpy(dYh).to(dev)
    xq = X[0] + (X[-1] - X[0]) * rng.random(30)
    ct, cs = nn.cumtrapz(Y, Xr).cpu().numpy(), nn.cumsimpson(Y, Xr).cpu().numpy()
    ev = nn.newHermiteSpline(Xr, Y, dY).eval(xq).cpu().numpy()
    es = nn.newHermiteSpline(Xr, Y).derivEval(xq).cpu().numpy()
    for m in range(6):
        assert np.array_equal(ct[:, m], O.cumtrapz(Yh[:, m], Xr)) and np.array_equal(cs[:, m], O.cumsimpson(Yh[:, m], Xr))
        assert np.array_equal(ev[:, m], O.hermite_interp(Xr, Yh[:, m], dYh[:, m], xq))
        xs, (ys,) = O.sort_and_trim(Xr, Yh[:, m])
        assert np.array_equal(es[:, m], O.hermite_interp(xs, ys, O.hermite_slopes(Xr, Yh[:, m]), xq, deriv=True))
    assert np.array_equal(nn.cumtrapz(Yh, Xr), ct) and np.array_equal(nn.cumsimpson(Yh, Xr), cs)      # the host-pointer entries
    assert np.array_equal(nn.newHermiteSpline(Xr, Yh, dYh).eval(xq), ev) and np.array_equal(nn.newHermiteSpline(Xr, Yh).derivEval(xq), es)
