"""Function-argument forms of cumtrapz / cumsimpson on the device (integrate.nim:138-175, 377-400) against the
oracle's op-for-op restatement: bit-exact for arithmetic-only integrands, 1e-12 for integrands that call cos (the device's
and glibc's cos are both < 1 ulp but not the same function)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import numericalnim_amd as nn
    from oracle import oracle as O
    assert torch.cuda.is_available()
    return nn, O, torch


POLY_SRC = "for (int c = 0; c < {d}; ++c) dy[c] = ((p[0] * t + p[1]) * t) * (1.0 + (double)c) + p[2];"


def _poly(nn, d):
    return nn.Rhs.custom(d, POLY_SRC.format(d=d), keys=("a", "b", "c"), name=f"poly{d}")


@pytest.mark.parametrize("rule", ["trapz", "simpson"])
@pytest.mark.parametrize("dx", [1e-3, 0.1, 0.37])
@pytest.mark.parametrize("dim", [1, 3])
def test_reference_harness_shape_bit_exact(env, rule, dx, dim):
    """tests/test_integrate.nim:72-95 shape (X = linspace(0, 3pi/2, 17), dx default / 0.1) with a polynomial integrand: bit-exact."""
    nn, O, torch = env
    X = np.array(O.linspace(0.0, 1.5 * math.pi, 17))
    params = [0.75, -1.25, 0.5]
    f = _poly(nn, dim)
    ctx = nn.newNumContext({"a": params[0], "b": params[1], "c": params[2]})
    fn = nn.cumtrapz if rule == "trapz" else nn.cumsimpson
    got = fn(f, X, ctx=ctx, dx=dx, n=5).cpu().numpy()
    ref = O.cumquad_fn(rule, O.RHS_POLY_T, params, 0 if dim == 1 else dim, X, dx)
    assert got.shape[0] == len(ref) == 17
    for i in range(5):
        g = got[:, i] if dim == 1 else got[:, :, i]
        assert np.array_equal(g, ref), (rule, dx, dim, np.abs(g - ref).max())


@pytest.mark.parametrize("rule,tol", [("trapz", 1e-1), ("simpson", 1e-3)])
@pytest.mark.parametrize("dx", [1e-5, 0.1])
def test_reference_cos_kat(env, rule, tol, dx):
    """The reference's own four tests (tests/test_integrate.nim:72-95): f = a cos(x), a = 2, against 2 sin(x) at the reference's
    tolerance, and against the oracle at 1e-12."""
    nn, O, torch = env
    X = np.array(O.linspace(0.0, 1.5 * math.pi, 17))
    f = nn.Rhs.custom(1, "dy[0] = p[0] * cos(t);", keys=("a",), name="acos")
    fn = nn.cumtrapz if rule == "trapz" else nn.cumsimpson
    got = fn(f, X, ctx=nn.newNumContext({"a": 2.0}), dx=dx).cpu().numpy()[:, 0]
    assert got.shape == (17,)
    assert np.abs(got - 2.0 * np.sin(X)).max() < tol
    ref = O.cumquad_fn(rule, O.RHS_COS_T, [2.0], 0, X, dx)
    assert np.abs(got - ref).max() < 1e-12


@pytest.mark.parametrize("rule", ["trapz", "simpson"])
def test_parameter_sweep_and_builtin_kind(env, rule):
    """Batch axis = parameter sweep (every item its own ctx); compiled-in AFFINE_T kind (f = b*x at y = 0); SoA and AoS."""
    nn, O, torch = env
    dev = torch.device("cuda", 0)
    N = 1000
    rng = np.random.default_rng(5)
    a = rng.uniform(-2, 2, N)
    b = rng.uniform(-2, 2, N)
    X = np.sort(rng.uniform(-1.0, 2.0, 23))
    fn = nn.cumtrapz if rule == "trapz" else nn.cumsimpson
    sweep = torch.tensor(np.stack([a, b]), device=dev)
    for layout in (nn.LAYOUT_SOA, nn.LAYOUT_AOS):
        got = fn(nn.Rhs.affine_t(0.0, 0.0), X, dx=0.01, sweep=sweep, dim=2, layout=layout).cpu().numpy()
        for i in (0, 1, 499, 999):
            ref = O.cumquad_fn(rule, O.RHS_AFFINE_T, [a[i], b[i]], 2, X, 0.01)
            g = got[:, :, i] if layout == nn.LAYOUT_SOA else got[:, i, :]
            assert np.array_equal(g, ref)


@pytest.mark.parametrize("rule", ["trapz", "simpson"])
def test_unsorted_duplicate_and_degenerate_queries(env, rule):
    """hermiteInterpolate's two branches (utils.nim:290-311): unsorted X keeps X's order; sorted X with repeated maxima drops
    rows exactly as the reference does; row counts equal the oracle's."""
    nn, O, torch = env
    f = _poly(nn, 1)
    ctx = nn.newNumContext({"a": 0.5, "b": 2.0, "c": -1.0})
    fn = nn.cumtrapz if rule == "trapz" else nn.cumsimpson
    cases = [np.array([0.3, -0.2, 1.7, 0.9, 1.7, -0.2]), np.array([0.0, 0.5, 0.5, 1.0, 1.0]), np.array([2.0, 1.0, 0.0]),
             np.array([0.0, 1.0]), np.array([-3.0, -1.0, -2.0, -1.5])]
    for X in cases:
        for dx in (0.01, 0.25):
            ref = O.cumquad_fn(rule, O.RHS_POLY_T, [0.5, 2.0, -1.0], 0, X, dx)
            got = fn(f, X, ctx=ctx, dx=dx).cpu().numpy()[:, 0]
            assert got.shape == ref.shape, (rule, X, dx, got.shape, ref.shape)
            assert np.array_equal(got, ref), (rule, X, dx)


def test_errors(env):
    nn, O, torch = env
    f = _poly(nn, 1)
    ctx = nn.newNumContext({"a": 0.5, "b": 2.0, "c": -1.0})
    with pytest.raises(ValueError):
        nn.cumtrapz(f, [0.0, 1.0], ctx=ctx, dx=0.0)       # the reference would never terminate
    with pytest.raises(ValueError):
        nn.cumsimpson(f, [0.0, 1.0], ctx=ctx, dx=-1.0)
    with pytest.raises(ValueError):
        nn.cumtrapz(f, [0.0, float("nan")], ctx=ctx)
    with pytest.raises(ValueError):
        nn.cumsimpson(f, [1.0, 1.0], ctx=ctx)              # 2 grid points: "at least 3 elements" ValueError (integrate.nim:345-346)
    with pytest.raises(ValueError):
        O.cumquad_fn("simpson", O.RHS_POLY_T, [0.5, 2.0, -1.0], 0, [1.0, 1.0], 1e-5)
    with pytest.raises(NotImplementedError):
        nn.cumtrapz(nn.Rhs.ring(), [0.0, 1.0], dim=16)     # no thread-per-item kernel for a 16-component built-in kind


def test_large_sweep_matches_closed_form(env):
    """1e5 parameter sets x 1e4 grid points: the result of integrating a*x^2 + b*x + c is (a/3)x^3 + (b/2)x^2 + c x."""
    nn, O, torch = env
    dev = torch.device("cuda", 0)
    N = 100000
    g = torch.Generator(device="cpu").manual_seed(3)
    sw = torch.rand(3, N, generator=g, dtype=torch.float64).to(dev) * 2 - 1
    f = nn.Rhs.custom(1, POLY_SRC.format(d=1), keys=("a", "b", "c"), defaults={"a": 0.0, "b": 0.0, "c": 0.0}, name="poly_sweep")
    X = np.linspace(0.0, 1.0, 11)
    got = nn.cumsimpson(f, X, dx=1e-4, sweep=sw)
    xs = torch.tensor(X, device=dev)[:, None]
    exact = sw[0] / 3 * xs ** 3 + sw[1] / 2 * xs ** 2 + sw[2] * xs
    assert float((got - exact).abs().max()) < 1e-12
    got = nn.cumtrapz(f, X, dx=1e-4, sweep=sw)
    assert float((got - exact).abs().max()) < 1e-8


def test_against_committed_golden_vectors(env):
    """tests/golden/quad_golden.json (144 frozen cases: both rules, sorted / unsorted / duplicated X, dx 0.01 / 0.1 / 0.37, scalar and
    3-component integrands): the device result is bit-identical, row count included — data, not a live oracle run."""
    import json
    import os
    nn, O, torch = env
    cases = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "quad_golden.json")))["cases"]
    fh = lambda xs: np.array([float.fromhex(x) for x in xs])  # noqa: E731
    fs = {1: _poly(nn, 1), 3: _poly(nn, 3)}
    for c in cases:
        p = fh(c["params"])
        d = max(c["dim"], 1)
        ctx = nn.newNumContext({"a": p[0], "b": p[1], "c": p[2]})
        fn = nn.cumtrapz if c["rule"] == "trapz" else nn.cumsimpson
        got = fn(fs[d], fh(c["X"]), ctx=ctx, dx=float.fromhex(c["dx"]), n=2).cpu().numpy()
        assert got.shape[0] == c["rows"], c["name"]
        g = got[:, 0] if d == 1 else got[:, :, 0]
        assert np.array_equal(g.ravel(), fh(c["out"])), c["name"]
