"""The reference's Vector[float] state has any length.  Sizes without an ahead-of-time kernel run through run-time instantiation
of the same per-component expressions (ode_rtc.hip: rtc_builtin_kind); results must equal the oracle's exactly as for the
ahead-of-time sizes: bit-exact with identical step counts for fixed-step and adaptive methods."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KINDS = {"neg_y": (0, []), "linear": (1, [-0.35]), "affine_t": (4, [-0.5, 0.25]), "ring": (3, [0.1])}


@pytest.fixture(scope="module")
def env():
    import torch
    import numericalnim_amd as nn
    from oracle import oracle as O
    assert torch.cuda.is_available()
    return nn, O, torch


def _rhs(nn, name, params):
    if name == "neg_y":
        return nn.Rhs.neg_y(), None
    if name == "linear":
        return nn.Rhs.linear(params[0]), None
    if name == "affine_t":
        return nn.Rhs.affine_t(params[0], params[1]), None
    return nn.Rhs.ring(params[0]), None


@pytest.mark.parametrize("name", list(KINDS))
@pytest.mark.parametrize("dim", [5, 7, 12, 20, 32, 33, 64, 100])
@pytest.mark.parametrize("integrator", ["rk4", "tsit54"])
def test_any_dim_solve_matches_oracle(env, name, dim, integrator):
    nn, O, torch = env
    kind, params = KINDS[name]
    n = 300
    rng = np.random.default_rng(dim * 7 + kind)
    y0 = 0.5 + rng.random((dim, n))
    ts = [0.0, 0.4, 1.0]
    kw = dict(dt=1e-2) if integrator == "rk4" else dict(absTol=1e-9, relTol=1e-9, dtMax=0.1, dtMin=1e-6)
    f, _ = _rhs(nn, name, params)
    assert nn._lib.lib().nnhip_ode_supported(nn.ode.integrator_id(integrator), kind, dim, 0, 0) == 1
    t, y, cnt = nn.solveODE(f, torch.from_numpy(y0).cuda(), ts, nn.newODEoptions(**kw), integrator=integrator, return_counts=True)
    ref = O.solve_ode_batch(kind, params, y0, n, dim, ts, O.new_options(**kw), integrator, n_threads=8)
    got = y.cpu().numpy()
    assert np.array_equal(got, ref["y"]), float(np.abs(got - ref["y"]).max())
    assert np.array_equal(cnt["steps"].cpu().numpy(), ref["steps"])


def test_any_dim_aos_layout_and_step_entry(env):
    nn, O, torch = env
    dim, n = 6, 257
    rng = np.random.default_rng(3)
    y0 = 0.5 + rng.random((n, dim))  # AoS
    kind, params = KINDS["ring"]
    ts = [0.0, 1.0]
    t, y = nn.solveODE(nn.Rhs.ring(params[0]), torch.from_numpy(y0).cuda(), ts, nn.newODEoptions(dt=1e-2), integrator="kutta4", layout=nn.LAYOUT_AOS)
    ref = O.solve_ode_batch(kind, params, y0, n, dim, ts, O.new_options(dt=1e-2), "kutta4", layout=1, n_threads=8)
    assert np.array_equal(y.cpu().numpy(), ref["y"])
    # one IntegratorProc call (ode.nim:38) at dim 6
    ysoa = torch.from_numpy(np.ascontiguousarray(y0.T)).cuda()
    fsal = nn.rhsBatch(nn.Rhs.ring(params[0]), 0.0, ysoa)
    out = nn.integratorStep(nn.Rhs.ring(params[0]), 0.0, ysoa, fsal, 0.05, nn.newODEoptions(), integrator="dopri54")
    yn = out[0].cpu().numpy()
    for i in (0, 100, 256):
        r = O.step(kind, params, "dopri54", O.new_options(), 0.0, y0[i], O.rhs(kind, params, 0.0, y0[i]), 0.05)
        assert np.array_equal(yn[:, i], r[0]), i


HEAT = ("const double l = c > 0 ? y[c - 1] : 0.0; const double r = c + 1 < dim ? y[c + 1] : 0.0; "
        "return p[0] * ((l - 2.0 * y[c]) + r);")


@pytest.mark.parametrize("dim", [24, 128, 200, 256])
@pytest.mark.parametrize("integrator", ["rk4", "dopri54", "vern65"])
@pytest.mark.parametrize("layout", [0, 1])
def test_wide_user_system_method_of_lines(env, dim, integrator, layout):
    """A method-of-lines heat equation with up to 256 unknowns per system, given per component as source: lanes-per-system kernels
    with 4 components per lane (up to a full wavefront per system), dense output, both layouts; the step entry for systems wider
    than one wavefront."""
    nn, O, torch = env
    n = 70
    rng = np.random.default_rng(dim)
    y0 = rng.random((dim, n))
    f = nn.Rhs.custom(dim, HEAT, keys=("kappa",), defaults={"kappa": 0.4}, name=f"heat{dim}", per_component=True)
    ts = [0.0, 0.1, 0.25]
    kw = dict(dt=5e-3) if integrator == "rk4" else dict(absTol=1e-8, relTol=1e-8, dtMax=0.05, dtMin=1e-7)
    y0l = np.ascontiguousarray(y0.T) if layout == 1 else y0
    t, y, cnt = nn.solveODE(f, torch.from_numpy(y0l).cuda(), ts, nn.newODEoptions(**kw), integrator=integrator, layout=layout, return_counts=True)
    ref = O.solve_ode_batch(O.RHS_HEAT, [0.4], y0l, n, dim, ts, O.new_options(**kw), integrator, layout=layout, n_threads=8)
    got = y.cpu().numpy()
    assert np.array_equal(got, ref["y"]), float(np.abs(got - ref["y"]).max())
    assert np.array_equal(cnt["steps"].cpu().numpy(), ref["steps"])
    if layout == 0:
        ys = torch.from_numpy(y0).cuda()
        fsal = nn.rhsBatch(f, 0.0, ys)
        out = nn.integratorStep(f, 0.0, ys, fsal, 0.01, nn.newODEoptions(**kw), integrator=integrator)
        for i in (0, n - 1):
            r = O.step(O.RHS_HEAT, [0.4], integrator, O.new_options(**kw), 0.0, y0[:, i], O.rhs(O.RHS_HEAT, [0.4], 0.0, y0[:, i]), 0.01)
            assert np.abs(out[0].cpu().numpy()[:, i] - r[0]).max() <= (0 if integrator == "rk4" else 1e-12), (i, integrator)


def test_unsupported_sizes_still_refused(env):
    nn, O, torch = env
    y0 = torch.ones(257, 8, dtype=torch.float64, device="cuda")
    with pytest.raises(NotImplementedError):
        nn.solveODE(nn.Rhs.neg_y(), y0, [0.0, 1.0], nn.newODEoptions(dt=0.1), integrator="rk4")      # 257 components
    with pytest.raises(NotImplementedError):
        nn.solveODE(nn.Rhs.lorenz(), y0[:5], [0.0, 1.0], nn.newODEoptions(dt=0.1), integrator="rk4")  # Lorenz is 3-dimensional
    with pytest.raises(ValueError):
        nn.Rhs.custom(17, "dy[0] = 0;", name="too_wide_whole_vector")   # whole-vector bodies live in registers: <= 16 components
