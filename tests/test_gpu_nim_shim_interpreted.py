"""nim/numericalnim_hip.nim EXECUTED — by the Nim-subset interpreter of tests/nimrun.py, its `{.importc.}` procs bound to the built
libnnhip_ode.so — and compared bit for bit with the Python mirror (numericalnim_amd), which the rest of the suite pins to the oracle and the
reference's text.  Every public proc of the shim runs: the three `solveODE` overloads (batch; one tEnd + options object per IVP; one tspan per
IVP), parameter sweeps, divergence binning, right-hand sides from source (plain, per component with a halo, with a ctx block), the cumulative
quadratures and the Hermite spline.  See tests/nimrun.py for what an interpreted run shows and what only `nim c` can."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import numericalnim_amd as nn
    import nimrun
    assert torch.cuda.is_available()
    return nn, nimrun.load(), nimrun


def _batch(it, y0, layout="layoutSoA"):
    y0 = np.asarray(y0, dtype=np.float64)
    if y0.ndim == 1:
        n, dim = y0.shape[0], 1
    elif layout == "layoutSoA":
        dim, n = y0.shape
    else:
        n, dim = y0.shape
    return it.expr(f"OdeBatch(n: {n}, dim: {dim}, layout: {layout}, data: d)", d=[float(v) for v in y0.ravel()])


def _rows(ys, shape):
    return np.stack([np.array(b.get("data")).reshape(shape) for b in ys]) if ys else np.empty((0,) + shape)


def _ctx(it, **values):
    ctx = it.call("newNumContext")
    ctx.get("fvalues").update({k: float(v) for k, v in values.items()})
    return ctx


@pytest.mark.parametrize("integ", ["rk4", "DOPRI54", "tsit54", "vern65", "bs32", "heun2"])
def test_batch_solveode_on_the_references_harness(env, integ):
    """tests/test_ode.nim:5-46 shape: f = a y with a = -0.1 read from ctx.fValues, tspan = linspace(-10, 10, 100) — both directions, dense output."""
    nn, it, _ = env
    tspan = [float(v) for v in np.linspace(-10.0, 10.0, 100)]
    y0 = np.linspace(0.5, 1.5, 7)
    spec = it.expr('RhsSpec(kind: rhsLinear, keys: @["a"])')
    t, ys = it.call("solveODE", spec, _batch(it, y0), tspan, ctx=_ctx(it, a=-0.1), integrator=integ)
    tr, yr = nn.solveODE(nn.Rhs.linear(), y0, tspan, ctx=nn.newNumContext({"a": -0.1}), integrator=integ)
    assert np.array_equal(np.array(t), np.asarray(tr)) and np.array_equal(_rows(ys, y0.shape), np.asarray(yr))
    assert it.ffi_log[-2:] == ["nnhip_ode_integrator_id", "nnhip_ode_solve_batch_sweep_f64"]      # what the proc called, in its order


def test_batch_solveode_systems_options_sweep_and_layouts(env):
    nn, it, _ = env
    rng = np.random.default_rng(3)
    n = 33
    y0 = np.stack([rng.uniform(-10, 10, n), rng.uniform(-10, 10, n), rng.uniform(5, 30, n)])
    tspan = [0.0, 0.1, 0.25, 0.4]
    opt = it.call("newODEoptions", absTol=1e-8, relTol=1e-8, dtMax=1.0, dtMin=1e-9)
    popt = nn.newODEoptions(absTol=1e-8, relTol=1e-8, dtMax=1.0, dtMin=1e-9)
    spec = it.expr('RhsSpec(kind: rhsLorenz, keys: @["sigma", "rho", "beta"])')
    ctx = _ctx(it, sigma=10.0, rho=28.0, beta=8.0 / 3.0)
    pctx = nn.newNumContext({"sigma": 10.0, "rho": 28.0, "beta": 8.0 / 3.0})
    lor = nn.Rhs.lorenz()
    for integ in ("dopri54", "tsit54"):
        t, ys = it.call("solveODE", spec, _batch(it, y0), tspan, opt, ctx, integ)
        tr, yr = nn.solveODE(lor, y0, tspan, popt, ctx=pctx, integrator=integ)
        assert np.array_equal(np.array(t), np.asarray(tr)) and np.array_equal(_rows(ys, y0.shape), np.asarray(yr)), integ
        # the same batch as [N][dim]
        t2, ys2 = it.call("solveODE", spec, _batch(it, y0.T.copy(), "layoutAoS"), tspan, opt, ctx, integ)
        assert np.array_equal(_rows(ys2, (n, 3)), np.asarray(yr).transpose(0, 2, 1)), integ
    # a sweep of rho, every IVP its own ctx; in the caller's order, sorted by a key, and sorted by the automatic probe
    rho = rng.uniform(20.0, 35.0, n)
    sweep = [[10.0] * n, [float(v) for v in rho]]
    _, yr = nn.solveODE(lor, y0, tspan, popt, ctx=pctx, integrator="tsit54", sweep=np.array(sweep))
    for kw in (dict(), dict(sortBy=[float(-v) for v in rho]), dict(autoSort=True)):
        _, ys = it.call("solveODE", spec, _batch(it, y0), tspan, opt, ctx, "tsit54", sweep=sweep, **kw)
        assert np.array_equal(_rows(ys, y0.shape), np.asarray(yr)), kw
    with pytest.raises(Exception, match="one key per IVP"):
        it.call("solveODE", spec, _batch(it, y0), tspan, opt, ctx, "tsit54", sortBy=[1.0, 2.0])
    with pytest.raises(Exception, match="not a valid integrator"):       # ode.nim:651
        it.call("solveODE", spec, _batch(it, y0), tspan, opt, ctx, "rk5")
    with pytest.raises(Exception, match="KeyError|key not found"):      # a ctx without the parameter the right-hand side reads (tables.`[]`)
        it.call("solveODE", spec, _batch(it, y0), tspan, opt, _ctx(it, sigma=10.0), "tsit54")


def test_per_ivp_calls_overloads(env):
    """solveODE(f, y0, tEnd: openArray[float], options: openArray[ODEoptions]) and solveODE(f, y0, tspans: seq[seq[float]], options)"""
    nn, it, _ = env
    rng = np.random.default_rng(5)
    n = 19
    y0 = np.stack([rng.uniform(1.5, 2.5, n), np.zeros(n)])
    spec = it.expr('RhsSpec(kind: rhsVanDerPol, keys: @["mu"])')
    ctx, pctx = _ctx(it, mu=2.0), nn.newNumContext({"mu": 2.0})
    t_end = rng.uniform(-1.0, 2.0, n)
    t_end[3] = 0.25                                                       # = its tStart below: an empty span
    okw = [dict(absTol=10.0 ** -rng.integers(4, 9), relTol=1e-6, dtMax=0.5, dtMin=1e-8, tStart=float(ts)) for ts in rng.choice([0.0, 0.25, -0.5], n)]
    okw[3]["tStart"] = 0.25
    opts = [it.call("newODEoptions", **k) for k in okw]
    popts = [nn.newODEoptions(**k) for k in okw]
    ys, ny = it.call("solveODE", spec, _batch(it, y0), [float(v) for v in t_end], opts, ctx, "dopri54")
    yr, cr = nn.solveODECalls(nn.Rhs.vanderpol(), y0, t_end, popts, ctx=pctx, integrator="dopri54")
    a, b = _rows(ys, y0.shape), np.asarray(yr)
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a), np.nan_to_num(b)) and list(ny) == list(cr["ny"])
    one = it.call("solveODE", spec, _batch(it, y0), [float(v) for v in t_end], [opts[0]], ctx, "dopri54")       # one options object for all
    yr1, cr1 = nn.solveODECalls(nn.Rhs.vanderpol(), y0, t_end, popts[0], ctx=pctx, integrator="dopri54")
    assert np.array_equal(np.nan_to_num(_rows(one[0], y0.shape)), np.nan_to_num(np.asarray(yr1))) and list(one[1]) == list(cr1["ny"])
    with pytest.raises(Exception, match="one value per IVP"):
        it.call("solveODE", spec, _batch(it, y0), [0.5], opts, ctx, "dopri54")
    # ... with a sweep of mu on top (every call its own ctx as well)
    mus = [[float(v) for v in rng.uniform(0.5, 3.0, n)]]
    ysw, nysw = it.call("solveODE", spec, _batch(it, y0), [float(v) for v in t_end], opts, ctx, "dopri54", mus)
    yrw, crw = nn.solveODECalls(nn.Rhs.vanderpol(), y0, t_end, popts, ctx=pctx, integrator="dopri54", sweep=np.array(mus))
    assert np.array_equal(np.nan_to_num(_rows(ysw, y0.shape)), np.nan_to_num(np.asarray(yrw))) and list(nysw) == list(crw["ny"])
    # what the library refuses comes back through `check` as the reference's kind of exception: a requested time that is not finite
    with pytest.raises(Exception, match="ValueError|nnhip error"):
        it.call("solveODE", spec, _batch(it, y0), [0.0, float("nan")], ctx=ctx, integrator="dopri54")
    # every IVP its own tspan: any order, both sides of tStart, duplicates
    tspans = np.stack([rng.permutation(np.concatenate([rng.uniform(-1, 1.5, 4), [okw[i]["tStart"]]])) for i in range(n)])
    t, ys, ny = it.call("solveODE", spec, _batch(it, y0), [[float(v) for v in row] for row in tspans], opts, ctx, "tsit54")
    tr, yr, cr = nn.solveODECallsTspan(nn.Rhs.vanderpol(), y0, tspans, popts, ctx=pctx, integrator="tsit54")
    assert list(ny) == list(cr["ny"])
    for i in range(n):
        assert t[i] == [float(v) for v in tr[i] if v == v]
    a, b = _rows(ys, y0.shape), np.asarray(yr)
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a), np.nan_to_num(b))


def test_right_hand_sides_from_source(env):
    nn, it, _ = env
    rng = np.random.default_rng(9)
    n = 12
    y0 = np.stack([rng.uniform(-5, 5, n) for _ in range(3)])
    tspan = [0.0, 0.2, 0.5]
    ctx = _ctx(it, sigma=10.0, rho=28.0, beta=8.0 / 3.0)
    body = "dy[0] = p[0] * (y[1] - y[0]); dy[1] = y[0] * (p[1] - y[2]) - y[1]; dy[2] = y[0] * y[1] - p[2] * y[2];"
    f = it.call("rhsFromSource", 3, body, ["sigma", "rho", "beta"], "lorenz_from_nim")
    assert f.get("userkind") > 0
    _, ys = it.call("solveODE", f, _batch(it, y0), tspan, ctx=ctx, integrator="tsit54")
    _, yr = nn.solveODE(nn.Rhs.lorenz(), y0, tspan, integrator="tsit54")
    assert np.array_equal(_rows(ys, y0.shape), np.asarray(yr))
    # per component, with its halo declared: the ring of 16 (include/nnhip_ode.h: NNHIP_RHS_RING)
    d = 16
    yr0 = rng.uniform(-1, 1, (n, d))
    ring = it.call("rhsFromSourcePerComponent", d, f"return -((double)(c + 1) / {d}.0) * y[c] + p[0] * y[(c + 1) % {d}];", ["c"], "ring_from_nim", 0, 1)
    _, ys = it.call("solveODE", ring, _batch(it, yr0, "layoutAoS"), tspan, ctx=_ctx(it, c=0.1), integrator="dopri54")
    _, yref = nn.solveODE(nn.Rhs.ring(), yr0, tspan, ctx=nn.newNumContext({"c": 0.1}), integrator="dopri54", layout=1)
    assert np.array_equal(_rows(ys, (n, d)), np.asarray(yref))
    with pytest.raises(Exception):
        it.call("rhsFromSource", 1, "this is not C++", [], "broken")
    # NumContext in full: a shared vector, a per-IVP vector and a mutable slot counting the calls (ode.nim:599)
    vecs = [it.expr('CtxVector(name: "w", len: 2, perIvp: false)'), it.expr('CtxVector(name: "g", len: 1, perIvp: true)')]
    g = it.call("rhsFromSourceCtx", 1, "aux(0) += 1.0; dy[0] = p[0] * y[0] * w[1] + g(0) * w[0];", ["a"], vecs, 1, "ctx_from_nim")
    m = 6
    gains = [float(v) for v in rng.uniform(-1, 1, m)]
    it.call("bindCtx", g, [0.5, 2.0], gains, [0.0] * m, 1, m)
    yc0 = rng.uniform(0.5, 1.5, m)
    _, ys = it.call("solveODE", g, _batch(it, yc0), [0.0, 0.3], ctx=_ctx(it, a=-0.7), integrator="rk4")
    calls = it.call("readAux", g, 1, m)
    pg = nn.Rhs.custom(1, "aux(0) += 1.0; dy[0] = p[0] * y[0] * w[1] + g(0) * w[0];", keys=("a",), tvalues={"w": 2, "g": 1}, per_ivp=("g",), n_aux=1, name="ctx_py")
    import torch
    aux = torch.zeros((1, m), dtype=torch.float64, device="cuda")
    pctx = nn.newNumContext({"a": -0.7}, {"w": np.array([0.5, 2.0]), "g": np.array([gains]), "aux": aux})
    _, yr = nn.solveODE(pg, torch.from_numpy(yc0).cuda(), [0.0, 0.3], ctx=pctx, integrator="rk4")
    assert np.array_equal(_rows(ys, yc0.shape), yr.cpu().numpy())
    assert calls == aux.cpu().numpy()[0].tolist() and calls[0] > 4.0     # the mutable ctx slot: as many right-hand-side calls as the mirror counted


def test_consumers(env):
    nn, it, _ = env
    rng = np.random.default_rng(11)
    X = [float(v) for v in np.linspace(0.0, 2.0, 9)]
    n = 5
    Y = rng.uniform(-1, 1, (len(X), n))
    dY = rng.uniform(-1, 1, (len(X), n))
    Yb = [_batch(it, row) for row in Y]
    dYb = [_batch(it, row) for row in dY]
    for name, ref in (("cumtrapz", nn.cumtrapz(Y, X)), ("cumsimpson", nn.cumsimpson(Y, X))):
        got = it.call(name, Yb, X)
        assert np.array_equal(_rows(got, (n,)), np.asarray(ref)), name
    xq = [float(v) for v in np.linspace(-0.25, 2.25, 14)]
    for built, pref in ((it.call("newHermiteSpline", X, Yb, dYb), nn.newHermiteSpline(X, Y, dY)), (it.call("newHermiteSpline", X, Yb), nn.newHermiteSpline(X, Y))):
        assert np.array_equal(_rows(it.call("eval", built, xq), (n,)), pref.eval(xq))
        assert np.array_equal(_rows(it.call("derivEval", built, xq), (n,)), pref.derivEval(xq))
        assert np.array_equal(_rows(it.call("eval", built, xq, 1), (n,)), pref.eval(xq, extrap="Edge"))
    with pytest.raises(Exception, match="same length"):
        it.call("newHermiteSpline", X[:-1], Yb)
    # cumtrapz(f, X, ctx, dx) with a polynomial integrand (bit-exact against the mirror), one parameter set and a sweep of three
    poly = it.call("rhsFromSource", 1, "dy[0] = (p[0] * t + p[1]) * t + p[2];", ["a", "b", "c"], "poly_from_nim")
    ppoly = nn.Rhs.custom(1, "dy[0] = (p[0] * t + p[1]) * t + p[2];", keys=("a", "b", "c"), name="poly_py")
    ctx, pctx = _ctx(it, a=0.75, b=-1.25, c=0.5), nn.newNumContext({"a": 0.75, "b": -1.25, "c": 0.5})
    for name, fn in (("cumtrapz", nn.cumtrapz), ("cumsimpson", nn.cumsimpson)):
        got = it.call(name, poly, X, ctx, 0.1)
        assert np.array_equal(_rows(got, (1,))[:, 0], fn(ppoly, X, ctx=pctx, dx=0.1).cpu().numpy()[:, 0]), name
        sw = [[0.5, 0.75, 1.0]]
        got = it.call(name, poly, X, ctx, 0.1, sw)
        ref = fn(ppoly, X, ctx=pctx, dx=0.1, sweep=np.array(sw), n=3)
        assert np.array_equal(_rows(got, (3,)), ref if isinstance(ref, np.ndarray) else ref.cpu().numpy()), name


def test_right_hand_sides_written_in_nim(env):
    """nim/rhs_macro.nim: `deviceRhs(dim, keys): body` — the user's f(t, y, ctx) (ode.nim:36) as a restricted Nim body, translated by the macro's own
    procs (interpreted), compiled by the backend, solved through the interpreted shim: the bits of the compiled-in kinds written the same way."""
    nn, _, nimrun = env
    it = nimrun.load(macros=True)
    rng = np.random.default_rng(21)
    n = 10
    y0 = np.stack([rng.uniform(-5, 5, n) for _ in range(3)])
    tspan = [0.0, 0.2, 0.5]
    f = it.call("deviceRhs", 3, ["sigma", "rho", "beta"], body=nimrun.nim_ast('''
dy[0] = ctx.fValues["sigma"] * (y[1] - y[0])
dy[1] = y[0] * (ctx.fValues["rho"] - y[2]) - y[1]
dy[2] = y[0] * y[1] - ctx.fValues["beta"] * y[2]
'''))
    _, ys = it.call("solveODE", f, _batch(it, y0), tspan, ctx=_ctx(it, sigma=10.0, rho=28.0, beta=8.0 / 3.0), integrator="dopri54")
    _, yr = nn.solveODE(nn.Rhs.lorenz(), y0, tspan, integrator="dopri54")
    assert np.array_equal(_rows(ys, y0.shape), np.asarray(yr))
    # a loop over the components, a `let`, an integer-to-float conversion, `mod` in an index: the ring of 16
    yr0 = rng.uniform(-1, 1, (16, n))
    ring = it.call("deviceRhs", 16, ["c"], body=nimrun.nim_ast('''
for i in 0 ..< 16:
  dy[i] = -(float(i + 1) / 16.0) * y[i] + ctx.fValues["c"] * y[(i + 1) mod 16]
'''))
    _, ys = it.call("solveODE", ring, _batch(it, yr0), tspan, ctx=_ctx(it, c=0.1), integrator="tsit54")
    _, yref = nn.solveODE(nn.Rhs.ring(), yr0, tspan, ctx=nn.newNumContext({"c": 0.1}), integrator="tsit54")
    assert np.array_equal(_rows(ys, yr0.shape), np.asarray(yref))
    # ctx.tValues: a shared vector and a per-IVP one, their layout given next to the body (deviceRhsCtx) — against the same right-hand side as C++ source
    import torch
    m = 6
    gains = [float(v) for v in rng.uniform(-1, 1, m)]
    g = it.call("deviceRhsCtx", 1, ["a"], [2, 1], [False, True],
                body=nimrun.nim_ast('dy[0] = ctx.fValues["a"] * y[0] * ctx.tValues["w"][1] + ctx.tValues["g"][0] * ctx.tValues["w"][0]'))
    it.call("bindCtx", g, [0.5, 2.0], gains, [], 0, m)
    yc0 = rng.uniform(0.5, 1.5, m)
    _, ys = it.call("solveODE", g, _batch(it, yc0), [0.0, 0.3], ctx=_ctx(it, a=-0.7), integrator="tsit54")
    pg = nn.Rhs.custom(1, "dy[0] = (((p[0] * y[0]) * w[1]) + (g(0) * w[0]));", keys=("a",), tvalues={"w": 2, "g": 1}, per_ivp=("g",), name="ctx_macro_py")
    pctx = nn.newNumContext({"a": -0.7}, {"w": np.array([0.5, 2.0]), "g": np.array([gains])})
    _, yr = nn.solveODE(pg, torch.from_numpy(yc0).cuda(), [0.0, 0.3], ctx=pctx, integrator="tsit54")
    assert np.array_equal(_rows(ys, yc0.shape), yr.cpu().numpy())


def test_shim_plus_library_return_what_the_references_text_returns(env):
    """The closing of the loop: nim/numericalnim_hip.nim (interpreted) over the library (GPU) against tests/golden/reference_text_vectors.json — what
    the reference's OWN ode.nim returns, executed by the same interpreter, on the inputs of all 101 fixtures: the times and every bit of every row."""
    import json
    import os
    from golden_util import fh, load_cases
    nn, it, _ = env
    here = os.path.dirname(os.path.abspath(__file__))
    reftext = {c["name"]: c for c in json.load(open(os.path.join(here, "golden", "reference_text_vectors.json")))["cases"]}
    kinds = {0: ("rhsNegY", ()), 1: ("rhsLinear", ("a",)), 2: ("rhsLorenz", ("sigma", "rho", "beta")), 3: ("rhsRing", ("c",)), 4: ("rhsAffineT", ("a", "b")),
             5: ("rhsVanDerPol", ("mu",))}
    done = 0
    for case in load_cases():
        dim = max(case["dim"], 1)
        integ = nn._lib.lib().nnhip_ode_integrator_id(case["integrator"].encode())
        if not nn._lib.lib().nnhip_ode_supported(integ, case["rhs_kind"], dim, 0, 0):
            continue
        kname, keys = kinds[case["rhs_kind"]]
        spec = it.expr(f"RhsSpec(kind: {kname}, keys: k)", k=list(keys))
        ctx = _ctx(it, **dict(zip(keys, fh(case["params"]))))
        y0 = np.stack([fh(y) for y in case["y0"]])                      # [n, dim]
        batch = _batch(it, y0[:, 0] if case["dim"] == 0 else np.ascontiguousarray(y0.T))
        opt = it.call("newODEoptions", **case["options"])
        t, ys = it.call("solveODE", spec, batch, [float(v) for v in fh(case["tspan"])], opt, ctx, case["integrator"])
        ref = reftext[case["name"]]
        assert [float(v).hex() for v in t] == ref["t"], case["name"]
        rows = _rows(ys, (dim, len(case["y0"])))                          # [n_t, dim, n]
        for i, r in enumerate(ref["ivps"]):
            got = rows[:r["n_y"], :, i]
            assert [float(v).hex() for v in got.ravel()] == r["y"], (case["name"], i)
            assert np.isnan(rows[r["n_y"]:, :, i]).all(), (case["name"], i)
        done += 1
    assert done >= 95, done


def test_the_shims_multi_gpu_branch(env):
    """solveODE(..., nGpus = k): contiguous shards of the batch (and of the sweep table) per device, nnhip_ode_solve_batch_multi_gpu_sweep_f64.  On a one-GPU box the
    shards over-subscribe the device (knob multi_gpu_oversubscribe, as tests/test_gpu_multi_gpu_entry.py does); with k devices they run one per device."""
    import torch
    nn, it, _ = env
    L = nn._lib.lib()
    rng = np.random.default_rng(31)
    n = 1001                                                                # ragged for 2, 3 and 8 shards
    y0 = np.stack([rng.uniform(-10, 10, n), rng.uniform(-10, 10, n), rng.uniform(5, 30, n)])
    tspan = [-0.1, 0.0, 0.2, 0.3]
    spec = it.expr('RhsSpec(kind: rhsLorenz, keys: @["sigma", "rho", "beta"])')
    ctx = _ctx(it, sigma=10.0, rho=28.0, beta=8.0 / 3.0)
    rho = [float(v) for v in rng.uniform(20.0, 35.0, n)]
    t1, y1 = it.call("solveODE", spec, _batch(it, y0), tspan, ctx=ctx, integrator="tsit54")
    _, y1s = it.call("solveODE", spec, _batch(it, y0), tspan, ctx=ctx, integrator="tsit54", sweep=[[10.0] * n, rho])
    ndev = torch.cuda.device_count()
    for k in (2, 3, 8):
        assert L.nnhip_tune_set(b"multi_gpu_oversubscribe", 1 if k > ndev else 0) == 0
        try:
            tk, yk = it.call("solveODE", spec, _batch(it, y0), tspan, ctx=ctx, integrator="tsit54", nGpus=k)
            _, yks = it.call("solveODE", spec, _batch(it, y0), tspan, ctx=ctx, integrator="tsit54", nGpus=k, sweep=[[10.0] * n, rho])
        finally:
            assert L.nnhip_tune_set(b"multi_gpu_oversubscribe", 0) == 0
        assert it.ffi_log[-1] == "nnhip_ode_solve_batch_multi_gpu_sweep_f64"
        assert tk == t1 and np.array_equal(_rows(yk, y0.shape), _rows(y1, y0.shape)) and np.array_equal(_rows(yks, y0.shape), _rows(y1s, y0.shape)), k
    with pytest.raises(Exception, match="not available together with nGpus"):
        it.call("solveODE", spec, _batch(it, y0), tspan, ctx=ctx, integrator="tsit54", nGpus=2, autoSort=True)


def test_zz_how_much_of_the_shim_ran(env):
    """Statement coverage of the interpreted runs above (this module's tests, in this process): which lines of nim/numericalnim_hip.nim and of
    nim/rhs_macro.nim's translating procs were executed at least once.  What stays unexecuted is named in the failure message, not hidden."""
    _, _, nimrun = env
    for unit, floor in (("numericalnim_hip.nim", 0.95), ("rhs_macro.nim", 0.75)):
        lines, ran = nimrun.coverage(unit)
        missed = sorted(lines - ran)
        assert len(lines) > 50 and len(ran & lines) >= floor * len(lines), f"{unit}: {len(ran & lines)} of {len(lines)} statement lines ran; not run: {missed}"
