"""User-supplied right-hand sides compiled at run time (hiprtc; nnhip_ode_rhs_compile) — the stand-in for the
reference's arbitrary closure f(t, y, ctx) (ODEProc[T], ode.nim:36)."""
import numpy as np
import pytest

LORENZ_SRC = "dy[0] = p[0]*(y[1]-y[0]); dy[1] = y[0]*(p[1]-y[2]) - y[1]; dy[2] = y[0]*y[1] - p[2]*y[2];"
DUFFING_SRC = "const double x = y[0], v = y[1]; dy[0] = v; dy[1] = ((-p[0]*v - p[1]*x) - p[2]*(x*x*x)) + p[3]*t;"
DUFF_P = dict(delta=0.3, alpha=-1.0, beta=1.0, gamma=0.37)


def test_registration_and_compile_errors_without_gpu(nn):
    """hiprtc targets gfx950 explicitly, so source errors surface at registration even with no device present."""
    f = nn.Rhs.custom(3, LORENZ_SRC, keys=("sigma", "rho", "beta"), defaults=dict(sigma=10.0, rho=28.0, beta=8.0 / 3.0))
    assert f.kind >= 1000 and f.params(None) == [10.0, 28.0, 8.0 / 3.0]
    assert nn._lib.lib().nnhip_ode_supported(1, f.kind, 3, 0, 0) == 1 and nn._lib.lib().nnhip_ode_supported(1, f.kind, 2, 0, 0) == 0
    with pytest.raises(ValueError, match="undeclared identifier"):
        nn.Rhs.custom(2, "dy[0] = y[1]; dy[1] = -p[0]*yy[0];", keys=("k",))
    with pytest.raises(ValueError):
        nn.Rhs.custom(0, "dy[0] = 0;")
    with pytest.raises(ValueError):
        nn.Rhs.custom(17, "dy[0] = 0;")
    assert nn._lib.lib().nnhip_ode_rhs_release(f.kind) == 0
    assert nn._lib.lib().nnhip_ode_rhs_release(f.kind) != 0


def _rhs_macro_expected_sources():
    """The HIP source nim/rhs_macro.nim's self-test says `deviceRhsSource` emits for the Lorenz body (the doAssert in its `when isMainModule`)."""
    import os
    import re
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nim", "rhs_macro.nim")).read()
    m = re.search(r'doAssert src == "((?:[^"\\]|\\.)*)"', text)
    assert m, "nim/rhs_macro.nim lost its self-test"
    return m.group(1).encode().decode("unicode_escape")


def test_rhs_macro_self_test_text_is_the_documented_form():
    src = _rhs_macro_expected_sources()
    assert src.count(";") == 3 and "p[0] * ((y[1] - y[0]))" in src and "nnhip" not in src  # one parenthesised C operation per Nim infix node


@pytest.mark.gpu
def test_source_the_nim_macro_emits_for_lorenz_is_the_builtin_bit_for_bit(nn, dev):
    """nim/rhs_macro.nim cannot be compiled here, but what it EMITS can: the source text its self-test pins for the Lorenz body goes through
    nnhip_ode_rhs_compile and must give the bits of the compiled-in kind (same association order, -ffp-contract=off)."""
    import torch
    f = nn.Rhs.custom(3, _rhs_macro_expected_sources(), keys=("sigma", "rho", "beta"), defaults=dict(sigma=10.0, rho=28.0, beta=8.0 / 3.0), name="lorenz_from_nim_macro")
    n = 300
    y0 = torch.from_numpy(np.stack([1.0 + (np.arange(n) % 1024) * 2.0 ** -20, np.ones(n), np.ones(n)])).to(dev)
    for m in ("rk4", "tsit54", "vern65"):
        ta, ya = nn.solveODE(f, y0, [-0.2, 0.0, 0.35, 0.5], nn.newODEoptions(dt=1e-3), integrator=m)
        tb, yb = nn.solveODE(nn.Rhs.lorenz(), y0, [-0.2, 0.0, 0.35, 0.5], nn.newODEoptions(dt=1e-3), integrator=m)
        assert np.array_equal(ta, tb) and torch.equal(ya, yb), m


@pytest.mark.gpu
def test_user_lorenz_equals_builtin_bitwise(nn, dev):
    import torch
    f = nn.Rhs.custom(3, LORENZ_SRC, keys=("sigma", "rho", "beta"), defaults=dict(sigma=10.0, rho=28.0, beta=8.0 / 3.0))
    n = 500
    y0 = torch.from_numpy(np.stack([1.0 + (np.arange(n) % 1024) * 2.0 ** -20, np.ones(n), np.ones(n)])).to(dev)
    ts = [-0.2, 0.0, 0.1, 0.35, 0.5]
    for m in nn.allODE:
        opt = nn.newODEoptions(dt=1e-3)
        ta, ya = nn.solveODE(f, y0, ts, opt, integrator=m)
        tb, yb = nn.solveODE(nn.Rhs.lorenz(), y0, ts, opt, integrator=m)
        assert np.array_equal(ta, tb) and torch.equal(ya, yb), m
    # IntegratorProc seam as well
    fs = torch.zeros_like(y0)
    a = nn.integratorStep(f, 0.0, y0, fs, 1e-2, integrator="tsit54")
    b = nn.integratorStep(nn.Rhs.lorenz(), 0.0, y0, fs, 1e-2, integrator="tsit54")
    assert all(torch.equal(x, y) for x, y in zip(a, b))


@pytest.mark.gpu
@pytest.mark.parametrize("integrator", ["rk4", "kutta4", "dopri54", "tsit54", "vern65", "bs32"])
def test_user_duffing_matches_oracle(nn, oracle, dev, integrator):
    """A RHS that exists nowhere in the library (time-dependent Duffing oscillator): HIP via hiprtc vs the oracle's closure."""
    import torch
    O = oracle
    f = nn.Rhs.custom(2, DUFFING_SRC, keys=("delta", "alpha", "beta", "gamma"), defaults=DUFF_P, name="duffing")
    rng = np.random.default_rng(4)
    n = 300
    y0 = rng.uniform(-1.5, 1.5, (2, n))
    ts = O.linspace(-1.0, 2.0, 31)
    kw = dict(dt=1e-2, absTol=1e-8, relTol=1e-8, dtMin=1e-6, dtMax=5e-2)
    t, y, cnt = nn.solveODE(f, torch.from_numpy(y0).to(dev), ts, nn.newODEoptions(**kw), integrator=integrator, return_counts=True)
    ref = O.solve_ode_batch(O.RHS_DUFFING, list(DUFF_P.values()), y0, n, 2, ts, O.new_options(**kw), integrator, n_threads=8)
    got = y.cpu().numpy()
    assert np.array_equal(t, ref["t"])
    if integrator in nn.fixedODE:
        assert np.array_equal(got, ref["y"])
    else:
        assert np.abs(got - ref["y"]).max() <= 1e-6
    assert np.array_equal(cnt["steps"].cpu().numpy(), ref["steps"])


RING_COMP_SRC = "return -((double)(c + 1) / (double)DIM_) * y[c] + p[0] * y[(c + 1) % DIM_];"


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [4, 8, 16, 32])
def test_user_per_component_rhs_equals_builtin_ring(nn, dev, dim):
    """Per-component user RHS (nnhip_ode_rhs_compile_comp): dims 8/16/32 run on the lanes-per-system kernels, dim 4
    thread-per-IVP; all must reproduce the built-in ring RHS bit for bit, fused solve and step entry."""
    import torch
    f = nn.Rhs.custom(dim, RING_COMP_SRC.replace("DIM_", str(dim)), keys=("c",), defaults=dict(c=0.1), per_component=True, name=f"ring{dim}")
    rng = np.random.default_rng(dim)
    y0 = torch.from_numpy(rng.uniform(0.5, 2.0, (333, dim))).to(dev)
    ts = [-0.25, 0.0, 0.2, 0.5]
    for m in ("rk4", "tsit54", "vern65", "heun3"):
        ta, ya = nn.solveODE(f, y0, ts, nn.newODEoptions(dt=1e-2), integrator=m, layout=1)
        tb, yb = nn.solveODE(nn.Rhs.ring(0.1), y0, ts, nn.newODEoptions(dt=1e-2), integrator=m, layout=1)
        assert torch.equal(ya, yb), (dim, m)
    fs = nn.rhsBatch(nn.Rhs.ring(0.1), 0.0, y0, layout=1)
    assert torch.equal(nn.rhsBatch(f, 0.0, y0, layout=1), fs)
    a = nn.integratorStep(f, 0.0, y0, fs, 5e-2, integrator="dopri54", layout=1)
    b = nn.integratorStep(nn.Rhs.ring(0.1), 0.0, y0, fs, 5e-2, integrator="dopri54", layout=1)
    assert all(torch.equal(x, y) for x, y in zip(a, b))


@pytest.mark.gpu
def test_user_rhs_reads_a_device_table_through_a_parameter_slot(nn, dev):
    """ctx "can be used ... to pass in big Tensors" (ode.nim:599).  On the device the idiom is: hand the table's device
    address to the RHS through a parameter slot (bit pattern of the pointer in a double) and index it in the RHS source."""
    import struct
    import torch
    table = torch.linspace(0.0, 2.0, 4097, dtype=torch.float64, device=dev) ** 2        # forcing samples on t in [0, 1]: g(t) = (2t)^2
    addr_as_double = struct.unpack("d", struct.pack("q", table.data_ptr()))[0]
    body = ("const double* tab = (const double*)__double_as_longlong(p[1]);\n"
            "const double s = t * 4096.0; int j = (int)s; j = j < 0 ? 0 : (j > 4095 ? 4095 : j);\n"
            "const double g = tab[j] + (s - (double)j) * (tab[j + 1] - tab[j]);\n"   # linear interpolation in the table
            "dy[0] = p[0] * y[0] + g;")
    f_tab = nn.Rhs.custom(1, body, keys=("a", "table"), defaults=dict(a=-1.5, table=addr_as_double), name="forced_table")
    f_ref = nn.Rhs.custom(1, "dy[0] = p[0] * y[0] + 4.0 * t * t;", keys=("a",), defaults=dict(a=-1.5), name="forced_exact")
    y0 = torch.linspace(0.5, 1.5, 1000, dtype=torch.float64, device=dev)
    ts = [0.0, 0.25, 0.5, 1.0]
    for m in ("rk4", "tsit54"):
        _, ya = nn.solveODE(f_tab, y0, ts, nn.newODEoptions(dt=1e-3), integrator=m)
        _, yb = nn.solveODE(f_ref, y0, ts, nn.newODEoptions(dt=1e-3), integrator=m)
        assert float((ya - yb).abs().max()) < 1e-6      # table interpolation error of a quadratic on a 4096-cell grid ~ 6e-8
        assert torch.isfinite(ya).all()


@pytest.mark.gpu
def test_per_component_source_with_a_declared_halo(nn, dev):
    """Rhs.custom(..., per_component=True, halo=(lo, hi)) (nnhip_ode_rhs_set_halo): a body that reads only its neighbours takes them from the
    adjacent lanes (the banded form of LpsOps::rhs, what the compiled-in ring uses) instead of the LDS stage vector — the same expression,
    so the same bits as the undeclared form and as the compiled-in system: fused and streamed, FSAL and non-FSAL methods, both
    layouts, dense output, a non-cyclic stencil (heat equation), and sizes where the banded form does not apply (a padded system, a
    thread-per-IVP one).  A declaration that does not cover the body is refused at registration."""
    import torch
    ring_src = "return -((double)(c + 1) / (double)dim) * y[c] + p[0] * y[(c + 1) % dim];"
    heat_src = "const double l = c > 0 ? y[c - 1] : 0.0; const double r = c + 1 < dim ? y[c + 1] : 0.0; return p[0] * ((l - 2.0 * y[c]) + r);"
    rng = np.random.default_rng(3)
    kw = dict(absTol=1e-8, relTol=1e-8, dtMin=1e-8, dtMax=0.25)
    for dim in (16, 32, 8):
        f_h = nn.Rhs.custom(dim, ring_src, keys=("c",), defaults={"c": 0.1}, name=f"ring{dim}_halo", per_component=True, halo=(0, 1))
        f_p = nn.Rhs.custom(dim, ring_src, keys=("c",), defaults={"c": 0.1}, name=f"ring{dim}_plain", per_component=True)
        assert f_h.kind != f_p.kind
        n = 513
        y0 = rng.uniform(0.5, 1.5, (n, dim))
        for layout, y in ((1, torch.from_numpy(y0).to(dev)), (0, torch.from_numpy(np.ascontiguousarray(y0.T)).to(dev))):
            for integ in ("tsit54", "dopri54", "vern65", "rk4", "bs32"):
                opt = nn.newODEoptions(dt=1e-2, **kw)
                ts = [-0.2, 0.0, 0.3, 1.0]
                a = nn.solveODE(f_h, y, ts, opt, integrator=integ, layout=layout)[1]
                b = nn.solveODE(f_p, y, ts, opt, integrator=integ, layout=layout)[1]
                c = nn.solveODE(nn.Rhs.ring(0.1), y, ts, opt, integrator=integ, layout=layout)[1]
                assert torch.equal(a, b) and torch.equal(a, c), (dim, layout, integ)
            for integ in ("tsit54", "bs32"):
                opt = nn.newODEoptions(**kw)
                yf = nn.solveODE(f_p, y, [0.0, 1.0], opt, integrator=integ, layout=layout)[1][-1]
                ys, launches = nn.adaptiveStream(f_h, y.clone(), 0.0, 1.0, opt, integrator=integ, layout=layout)
                assert torch.equal(ys, yf), (dim, layout, integ)
                t2, yd, ny, l2 = nn.adaptiveStreamSolve(f_h, y, [0.0, 0.4, 1.0], opt, integrator=integ, layout=layout)
                assert torch.equal(yd, nn.solveODE(f_p, y, [0.0, 0.4, 1.0], opt, integrator=integ, layout=layout)[1]), (dim, layout, integ)
    # a non-cyclic stencil; 24 unknowns are padded to 32 (the banded form does not apply: same results through the LDS path); 4: thread-per-IVP
    for dim in (64, 24, 4):
        f_h = nn.Rhs.custom(dim, heat_src, keys=("kappa",), defaults={"kappa": 0.4}, name=f"heat{dim}_halo", per_component=True, halo=(1, 1))
        f_p = nn.Rhs.custom(dim, heat_src, keys=("kappa",), defaults={"kappa": 0.4}, name=f"heat{dim}_plain", per_component=True)
        y = torch.from_numpy(rng.uniform(0.0, 1.0, (300, dim))).to(dev)
        for integ in ("dopri54", "rk4"):
            opt = nn.newODEoptions(dt=1e-2, **kw)
            assert torch.equal(nn.solveODE(f_h, y, [0.0, 0.5, 1.0], opt, integrator=integ, layout=1)[1], nn.solveODE(f_p, y, [0.0, 0.5, 1.0], opt, integrator=integ, layout=1)[1]), (dim, integ)
        ys, launches = nn.adaptiveStream(f_h, y.clone(), 0.0, 1.0, nn.newODEoptions(**kw), integrator="tsit54", layout=1)
        assert torch.equal(ys, nn.solveODE(f_p, y, [0.0, 1.0], nn.newODEoptions(**kw), integrator="tsit54", layout=1)[1][-1]), dim
    # declarations that do not cover the body, or make no sense
    with pytest.raises(ValueError):
        nn.Rhs.custom(16, ring_src, keys=("c",), defaults={"c": 0.1}, name="ring16_bad_halo", per_component=True, halo=(1, 0))
    with pytest.raises(ValueError):
        nn.Rhs.custom(16, "dy[0] = y[1];", name="whole_vector_halo", halo=(1, 1))
    L = nn._lib.lib()
    assert L.nnhip_ode_rhs_set_halo(f_h.kind, 9, 0) != 0 and L.nnhip_ode_rhs_set_halo(12345, 1, 1) != 0 and L.nnhip_ode_rhs_set_halo(nn.Rhs.ring(0.1).kind, 1, 1) != 0
