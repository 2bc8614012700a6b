"""DOPRI54 / Tsit54 (ode.nim:237-374, controller :57-76, driver :471-586) on the GPU vs the live oracle at
batch sizes the oracle finishes in seconds, plus full-size properties for BASELINE configs C3."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
# north_star states 1e-6 for adaptive methods.  Since round 2 the controller's pow is glibc's bit for bit (glibc_pow.hpp), so
# every comparison below demands bit-identity with the oracle instead — far inside the stated tolerance.
TOL_ADAPTIVE = 1e-6


def _same_bits(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))
LOR = [10.0, 28.0, 8.0 / 3.0]


# On the ISA-backed fake node (tests/fake_torch under the preloaded tests/cpp/fake_hip.cpp: scripts/run_gpu_suite_on_isa_node.py) a wavefront-instruction takes
# tens of microseconds: the round-5 tests below shrink their batches there and nowhere else.  On a device _sz(n, small) is n.
_ON_ISA_NODE = bool(os.environ.get("FAKE_HIP_LIB"))


def _sz(n, small):
    return small if _ON_ISA_NODE else n



def _lorenz_y0(n):
    return np.stack([1.0 + (np.arange(n) % 1024) * 2.0 ** -20, np.ones(n), np.ones(n)])  # BASELINE C3


@pytest.mark.parametrize("integrator", ["dopri54", "tsit54"])
@pytest.mark.parametrize("okey", ["default", "tight"])
def test_c3_lorenz_small_batch(nn, oracle, dev, integrator, okey):
    import torch
    O = oracle
    kw = {} if okey == "default" else dict(absTol=1e-10, relTol=1e-10, dtMin=1e-6, dtMax=1e-1)
    n = 2048
    y0 = _lorenz_y0(n)
    t, y, cnt = nn.solveODE(nn.Rhs.lorenz(), torch.from_numpy(y0).to(dev), [0.0, 1.0], nn.newODEoptions(**kw), integrator=integrator,
                            return_counts=True)
    ref = O.solve_ode_batch(O.RHS_LORENZ, LOR, y0, n, 3, [0.0, 1.0], O.new_options(**kw), integrator, n_threads=8)
    got = y.cpu().numpy()
    assert np.abs(got - ref["y"]).max() <= TOL_ADAPTIVE
    assert np.array_equal(cnt["steps"].cpu().numpy(), ref["steps"])
    assert np.array_equal(cnt["rejected"].cpu().numpy(), ref["rejected"])
    assert _same_bits(got, ref["y"])  # every trajectory bit-identical to the reference restatement


@pytest.mark.parametrize("integrator", ["rk4", "dopri54", "tsit54"])
def test_vector3_reference_harness(nn, oracle, dev, integrator):
    """tests/test_ode.nim:139-185 shape: Vector[float] of 3 equal components, f = -0.1*y, dense output both ways."""
    import torch
    O = oracle
    ts = O.linspace(-10.0, 10.0, 100)
    y0 = np.ones((3, 5)) * np.array([1.0, 0.5, 2.0, -1.0, 1.5])
    t, y = nn.solveODE(nn.Rhs.linear(-0.1), torch.from_numpy(y0).to(dev), ts, nn.newODEoptions(relTol=1e-8, dt=1e-2), integrator=integrator)
    assert np.array_equal(t, ts)
    got = y.cpu().numpy()
    ref = O.solve_ode_batch(O.RHS_LINEAR, [-0.1], y0, 5, 3, ts, O.new_options(relTol=1e-8, dt=1e-2), integrator)
    assert _same_bits(got, ref["y"])
    err = np.sqrt(((got[:, :, 0] - np.exp(-0.1 * ts)[:, None]) ** 2).sum(axis=1)) / 3.0   # isClose on Vector (utils.nim:252)
    assert np.all(err <= 1e-8)


@pytest.mark.parametrize("integrator", ["dopri54", "tsit54"])
def test_step_api_matches_oracle_step(nn, oracle, dev, integrator):
    """nnhip_ode_step_batch_f64_dev == one IntegratorProc call (ode.nim:38) per IVP, per-IVP t and dt."""
    import torch
    O = oracle
    n = 257
    rng = np.random.default_rng(5)
    y = rng.uniform(-5, 5, (3, n)) + np.array([[0.0], [0.0], [20.0]])
    t = rng.uniform(0, 2, n)
    dt = 10 ** rng.uniform(-4, -1.3, n)   # some large enough to be rejected and shrunk
    opt_kw = dict(absTol=1e-8, relTol=1e-8, dtMin=1e-7, dtMax=1e-1)
    fs = np.stack([O.rhs(O.RHS_LORENZ, LOR, t[i], list(y[:, i])) for i in range(n)], axis=1)
    yn, fn, dtu, err = nn.integratorStep(nn.Rhs.lorenz(), torch.from_numpy(t).to(dev), torch.from_numpy(y).to(dev),
                                         torch.from_numpy(fs).to(dev), torch.from_numpy(dt).to(dev), nn.newODEoptions(**opt_kw),
                                         integrator=integrator)
    # the same call into preallocated buffers (a loop that reuses them allocates nothing per call): the tensors handed in come back
    bufs = (torch.empty_like(yn), torch.empty_like(fn), torch.empty_like(dtu), None)
    r2 = nn.integratorStep(nn.Rhs.lorenz(), torch.from_numpy(t).to(dev), torch.from_numpy(y).to(dev), torch.from_numpy(fs).to(dev),
                           torch.from_numpy(dt).to(dev), nn.newODEoptions(**opt_kw), integrator=integrator, out=bufs)
    assert r2[0] is bufs[0] and r2[1] is bufs[1] and r2[2] is bufs[2]
    assert all(torch.equal(a, b) for a, b in zip(r2, (yn, fn, dtu, err)))
    yn, fn, dtu, err = (x.cpu().numpy() for x in (yn, fn, dtu, err))
    oo = O.new_options(**opt_kw)
    shrunk = 0
    for i in range(n):
        ryn, rfn, rdt, rerr = O.step(O.RHS_LORENZ, LOR, integrator, oo, t[i], list(y[:, i]), list(fs[:, i]), dt[i])
        assert _same_bits(yn[:, i], ryn) and _same_bits(fn[:, i], rfn)
        assert dtu[i] == rdt and err[i] == rerr  # the shrunk dt (in-step pow) and the error estimate carry the reference's bits
        shrunk += rdt < dt[i]
    assert shrunk > 10  # the in-step retry path (ode.nim:58-76) was exercised


def test_rhs_library_matches_oracle(nn, oracle, dev):
    """Pins the compiled-in RHS definitions (include/nnhip_ode.h) bitwise."""
    import ctypes as C
    import torch
    O = oracle
    L = nn._lib.lib()
    rng = np.random.default_rng(11)
    cases = [(O.RHS_NEG_Y, [], 1), (O.RHS_NEG_Y, [], 4), (O.RHS_LINEAR, [-0.1], 3), (O.RHS_AFFINE_T, [-0.5, 0.25], 2),
             (O.RHS_LORENZ, LOR, 3), (O.RHS_VANDERPOL, [1.5], 2), (O.RHS_RING, [0.1], 4)]
    for kind, params, dim in cases:
        n = 100
        y = rng.normal(size=(dim, n))
        yt = torch.from_numpy(y).to(dev)
        out = torch.empty_like(yt)
        p = np.asarray(params, dtype=np.float64)
        rc = L.nnhip_ode_rhs_batch_f64_dev(kind, p.ctypes.data_as(C.POINTER(C.c_double)) if len(p) else None, len(p), n, dim, 0, 0.75,
                                           yt.data_ptr(), out.data_ptr(), None)
        assert rc == 0
        torch.cuda.synchronize()
        ref = np.stack([np.atleast_1d(O.rhs(kind, params, 0.75, list(y[:, i]))) for i in range(n)], axis=1)
        assert np.array_equal(out.cpu().numpy(), ref), (kind, dim)


def test_hermite_kernel_matches_oracle(nn, oracle, dev):
    import torch
    O = oracle
    L = nn._lib.lib()
    rng = np.random.default_rng(3)
    n = 1000
    a = [torch.from_numpy(rng.normal(size=n)).to(dev) for _ in range(4)]
    out = torch.empty(n, dtype=torch.float64, device=dev)
    assert L.nnhip_hermite_spline_f64_dev(0.3, 0.1, 0.9, a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), a[3].data_ptr(), out.data_ptr(), n, None) == 0
    torch.cuda.synchronize()
    an = [x.cpu().numpy() for x in a]
    ref = np.array([O.hermite_spline(0.3, 0.1, 0.9, an[0][i], an[1][i], an[2][i], an[3][i]) for i in range(n)])
    assert np.array_equal(out.cpu().numpy(), ref)


@pytest.mark.parametrize("okey", ["default", "tight"])
@pytest.mark.parametrize("integrator", ["dopri54", "tsit54"])
def test_c3_full_size_properties(nn, oracle, dev, integrator, okey):
    """BASELINE C3 at full size (1e6 Lorenz IVPs), both option sets of SURVEY 8(d): y0 repeats with period 1024, so (1) the
    result must be exactly periodic in the IVP index (trajectories are independent: idempotence under batch position),
    (2) the first period equals the oracle bit for bit, (3) every IVP ends at tEnd with ny == 2."""
    import torch
    O = oracle
    kw = {} if okey == "default" else dict(absTol=1e-10, relTol=1e-10, dtMin=1e-6, dtMax=1e-1)
    n = 1_000_000
    y0 = torch.from_numpy(_lorenz_y0(n)).to(dev)
    t, y, cnt = nn.solveODE(nn.Rhs.lorenz(), y0, [0.0, 1.0], nn.newODEoptions(**kw), integrator=integrator, return_counts=True)
    yf = y[-1]
    assert torch.equal(y[0], y0)
    m = (n // 1024) * 1024
    assert torch.equal(yf[:, :m].reshape(3, -1, 1024), yf[:, :1024].reshape(3, 1, 1024).expand(3, m // 1024, 1024))
    ref = O.solve_ode_batch(O.RHS_LORENZ, LOR, _lorenz_y0(1024), 1024, 3, [0.0, 1.0], O.new_options(**kw), integrator, n_threads=8)
    assert _same_bits(yf[:, :1024].cpu().numpy(), ref["y"][-1])
    assert bool((cnt["ny"] == 2).all())
    assert np.array_equal(cnt["steps"][:1024].cpu().numpy(), ref["steps"])
    assert np.array_equal(cnt["rejected"][:1024].cpu().numpy(), ref["rejected"])


def test_host_pointer_entry_and_stats(nn, oracle, dev):
    """nnhip_ode_solve_batch_f64 (host buffers in, host buffers out) + aggregate stats."""
    O = oracle
    n = 1000
    y0 = _lorenz_y0(n)
    st = nn.ode.Stats()
    kw = dict(absTol=1e-6, relTol=1e-6, dtMin=1e-9, dtMax=1.0)
    t, y, cnt = nn.solveODE(nn.Rhs.lorenz(), y0, [0.0, 2.0, 5.0], nn.newODEoptions(**kw), integrator="dopri54", stats=st, return_counts=True)
    ref = O.solve_ode_batch(O.RHS_LORENZ, LOR, y0, n, 3, [0.0, 2.0, 5.0], O.new_options(**kw), "dopri54", n_threads=8)
    assert _same_bits(y, ref["y"])  # chaotic or not: same operations, same bits (T = 5 amplifies any ulp by ~e^{4.5})
    assert st.steps_total == int(cnt["steps"].sum()) and st.rejected_total == int(cnt["rejected"].sum())
    assert st.steps_max == int(cnt["steps"].max()) and st.ny_min == 3 and st.n_t_out == 3 and st.nan_aborts == 0
    assert st.rejected_total > 0 and st.kernel_ms > 0
    assert st.steps_total == int(ref["steps"].sum()) and st.rejected_total == int(ref["rejected"].sum())


def test_edge_cases(nn, oracle, dev):
    import torch
    O = oracle
    f = nn.Rhs.neg_y()
    # N = 1; tspan unsorted with tStart inside; duplicate tStart
    for tspan in ([1.0, -1.0, 0.0], [0.0, 0.0, 1.0], [-2.0, -1.0], [2.0], [0.5, 0.25, 0.75, 1.0]):
        for integ in ("rk4", "dopri54", "tsit54"):
            t, y, cnt = nn.solveODE(f, torch.tensor([1.5], dtype=torch.float64, device=dev), tspan, nn.newODEoptions(dt=1e-2),
                                    integrator=integ, return_counts=True)
            rt, ry, st = O.solve_ode(O.RHS_NEG_Y, [], 1.5, tspan, O.new_options(dt=1e-2), integ)
            assert np.array_equal(t, rt)
            g = y[:, 0].cpu().numpy()
            assert int(cnt["ny"][0]) == st.n_y
            assert _same_bits(g[:st.n_y], np.asarray(ry).reshape(-1))
            assert np.isnan(g[st.n_y:]).all()
    # NaN initial state: adaptive trajectory is aborted and flagged instead of spinning forever
    y0 = torch.tensor([1.0, float("nan"), 2.0], dtype=torch.float64, device=dev)
    t, y = nn.solveODE(f, y0, [0.0, 1.0], integrator="dopri54")
    g = y[-1].cpu().numpy()
    assert np.isnan(g[1]) and not np.isnan(g[0]) and not np.isnan(g[2])
    # options that would make the reference loop forever are refused
    with pytest.raises(ValueError):
        nn.solveODE(f, y0, [0.0, 1.0], nn.newODEoptions(dt=0.0), integrator="rk4")
    with pytest.raises(ValueError):
        nn.solveODE(f, y0, [0.0, 1.0], nn.newODEoptions(dtMin=0.0), integrator="dopri54")
    # max_steps truncation
    t, y, cnt = nn.solveODE(f, y0[:1], [0.0, 1.0], nn.newODEoptions(dt=1e-3), integrator="rk4", max_steps=10, return_counts=True)
    assert int(cnt["steps"][0]) == 10
    # unsupported combination is reported, not silently emulated
    with pytest.raises(NotImplementedError):
        nn.solveODE(nn.Rhs.lorenz(), torch.ones(2, 4, dtype=torch.float64, device=dev), [0.0, 1.0], integrator="rk4")


# ---- BASELINE config C4: Tsit54, 16-dim vector state, lanes-per-system kernel (LDS-staged stage vector) -------
def _ring_y0(n, d=16):
    s = np.arange(n)
    return (1.0 + np.arange(d)[None, :] / d + ((s % 1024) * 2.0 ** -20)[:, None])  # [n, d] AoS: y0[s][i] = 1 + i/16 + (s mod 1024)*2^-20


@pytest.mark.parametrize("layout", [1, 0], ids=["aos", "soa"])
@pytest.mark.parametrize("integrator", ["tsit54", "dopri54", "rk4"])
@pytest.mark.parametrize("okey", ["default", "tight"])
def test_c4_ring16_small_batch(nn, oracle, dev, integrator, okey, layout):
    import torch
    O = oracle
    kw = dict(dt=1e-3) if okey == "default" else dict(absTol=1e-10, relTol=1e-10, dtMin=1e-6, dtMax=1e-1, dt=1e-3)
    n = 1000  # not a multiple of 16 systems per workgroup
    y0 = _ring_y0(n)
    y0l = y0 if layout == 1 else np.ascontiguousarray(y0.T)
    t, y, cnt = nn.solveODE(nn.Rhs.ring(0.1), torch.from_numpy(y0l).to(dev), [0.0, 1.0], nn.newODEoptions(**kw), integrator=integrator,
                            layout=layout, return_counts=True)
    ref = O.solve_ode_batch(O.RHS_RING, [0.1], y0l, n, 16, [0.0, 1.0], O.new_options(**kw), integrator, layout=layout, n_threads=8)
    got = y.cpu().numpy()
    assert _same_bits(got, ref["y"])
    assert np.array_equal(cnt["steps"].cpu().numpy(), ref["steps"])
    assert np.array_equal(cnt["rejected"].cpu().numpy(), ref["rejected"])
    assert np.array_equal(cnt["ny"].cpu().numpy(), ref["ny"])


@pytest.mark.parametrize("dim", [8, 32])
def test_lps_other_dims_dense_backward(nn, oracle, dev, dim):
    """Lanes-per-system kernel with 8 and 32 lanes per system, dense output on both sides of tStart."""
    import torch
    O = oracle
    n = 37
    rng = np.random.default_rng(dim)
    y0 = rng.uniform(0.5, 2.0, (n, dim))
    ts = O.linspace(-0.5, 1.0, 13)
    for integ in ("rk4", "tsit54"):
        t, y = nn.solveODE(nn.Rhs.ring(0.1), torch.from_numpy(y0).to(dev), ts, nn.newODEoptions(dt=1e-2), integrator=integ, layout=1)
        ref = O.solve_ode_batch(O.RHS_RING, [0.1], y0, n, dim, ts, O.new_options(dt=1e-2), integ, layout=1)
        assert np.array_equal(t, ref["t"])
        assert _same_bits(y.cpu().numpy(), ref["y"])


def test_lps_step_api(nn, oracle, dev):
    import torch
    O = oracle
    n, d = 100, 16
    rng = np.random.default_rng(2)
    y = rng.uniform(0.5, 2.0, (n, d))
    kw = dict(absTol=1e-9, relTol=1e-9, dtMin=1e-7, dtMax=1e-1)
    fs = np.stack([O.rhs(O.RHS_RING, [0.1], 0.0, list(y[i])) for i in range(n)])
    dt = 10 ** rng.uniform(-3, -0.5, n)
    yn, fn, dtu, err = nn.integratorStep(nn.Rhs.ring(0.1), 0.0, torch.from_numpy(y).to(dev), torch.from_numpy(fs).to(dev),
                                         torch.from_numpy(dt).to(dev), nn.newODEoptions(**kw), integrator="tsit54", layout=1)
    yn, fn, dtu, err = (x.cpu().numpy() for x in (yn, fn, dtu, err))
    oo = O.new_options(**kw)
    for i in range(n):
        ryn, rfn, rdt, rerr = O.step(O.RHS_RING, [0.1], "tsit54", oo, 0.0, list(y[i]), list(fs[i]), dt[i])
        assert _same_bits(yn[i], ryn) and _same_bits(fn[i], rfn) and dtu[i] == rdt and err[i] == rerr


@pytest.mark.parametrize("okey", ["default", "tight"])
def test_c4_full_size_properties(nn, oracle, dev, okey):
    """BASELINE C4 at full size, both option sets of SURVEY 8(d): Tsit54, 1e6 systems x 16 components.  y0 is periodic in the
    system index with period 1024 -> the result must be exactly periodic too; the first period equals the oracle bit for bit."""
    import torch
    O = oracle
    kw = {} if okey == "default" else dict(absTol=1e-10, relTol=1e-10, dtMin=1e-6, dtMax=1e-1)
    n = 1_000_000
    y0 = torch.from_numpy(_ring_y0(n)).to(dev)
    t, y, cnt = nn.solveODE(nn.Rhs.ring(0.1), y0, [0.0, 1.0], nn.newODEoptions(**kw), integrator="tsit54", layout=1, return_counts=True)
    yf = y[-1]
    m = (n // 1024) * 1024
    assert torch.equal(yf[:m].reshape(-1, 1024, 16), yf[:1024].reshape(1, 1024, 16).expand(m // 1024, 1024, 16))
    ref = O.solve_ode_batch(O.RHS_RING, [0.1], _ring_y0(1024), 1024, 16, [0.0, 1.0], O.new_options(**kw), "tsit54", layout=1, n_threads=8)
    assert _same_bits(yf[:1024].cpu().numpy(), ref["y"][-1])
    assert np.array_equal(cnt["steps"][:1024].cpu().numpy(), ref["steps"])
    assert np.array_equal(cnt["rejected"][:1024].cpu().numpy(), ref["rejected"])
    assert bool((cnt["ny"] == 2).all())


@pytest.mark.parametrize("order", [2, 3, 5, 6])
def test_controller_factor_is_libm_exact(nn, oracle, dev, order):
    """The controller factor min(4, max(0.125, 0.9*pow(1/error, 1/order))) (ode.nim:71,537) computed on the device
    (glibc_pow.hpp: glibc's table-driven pow restated operation for operation) vs the oracle's, which calls the C
    library's pow as Nim's std/math pow does: bit-identical on every argument, including the ~0.07 % where glibc's pow
    is not correctly rounded, subnormal 1/error, 0, inf and NaN."""
    import torch
    L = nn._lib.lib()
    rng = np.random.default_rng(order)
    err = np.concatenate([10 ** rng.uniform(-8, 8, 1_000_000), 10 ** rng.uniform(-0.5, 0.5, 1_000_000),
                          1.0 + (rng.uniform(-1, 1, 500_000)) * 2.0 ** -rng.integers(1, 52, 500_000),   # the accept/reject knife edge
                          10 ** rng.uniform(-300, 308, 200_000),                                        # incl. subnormal 1/error
                          (0.9 / 4) ** order * (1 + rng.uniform(-3e-3, 3e-3, 300_000)),        # around the clamp early-out thresholds
                          (0.9 / 0.125) ** order * (1 + rng.uniform(-3e-3, 3e-3, 300_000)),
                          [1.0, 1.0 + 2 ** -52, 1.0 - 2 ** -53, 1e-300, 1e300, 1.7e308, 5e-324, np.inf, 3.0, 0.5, 0.0]])
    e = torch.from_numpy(err).to(dev)
    out = torch.empty_like(e)
    assert L.nnhip_ode_controller_factor_f64_dev(order, e.data_ptr(), out.data_ptr(), e.numel(), None) == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    with np.errstate(over="ignore", divide="ignore"):
        ref = oracle.controller_factor(err, order)
    assert np.array_equal(got, ref), (int((got != ref).sum()), err[got != ref][:5], got[got != ref][:5], ref[got != ref][:5])
    assert ((ref == 4.0) | (ref == 0.125)).sum() > 1000
    # correctly rounded value for comparison: glibc (and therefore the device) misses it on a small fraction of arguments
    with np.errstate(over="ignore", divide="ignore"):
        x = (1.0 / err[:2_000_000]).astype(np.longdouble)
        cr = np.exp(np.log(x) * np.longdouble(np.float64(1.0) / np.float64(order))).astype(np.float64)
        cr = np.minimum(4.0, np.maximum(0.125, 0.9 * cr))
    assert (got[:2_000_000] != cr).mean() < 0.01
    # NaN error propagates as NaN (the reference's min/max let NaN through, ode.nim:71)
    en = torch.tensor([float("nan")], dtype=torch.float64, device=dev)
    on = torch.empty_like(en)
    L.nnhip_ode_controller_factor_f64_dev(order, en.data_ptr(), on.data_ptr(), 1, None)
    torch.cuda.synchronize()
    assert np.isnan(on.cpu().numpy()[0])


@pytest.mark.parametrize("layout", [0, 1], ids=["soa", "aos"])
def test_multi_gpu_c_entry_single_device(nn, oracle, dev, layout):
    """nnhip_ode_solve_batch_multi_gpu_f64 (one process, G devices, host buffers) with the one device this box has:
    exercises the shard pack / unpack for both layouts, dense output included."""
    import ctypes as C
    O = oracle
    L = nn._lib.lib()
    n, dim = 777, 3
    y0 = np.stack([1.0 + (np.arange(n) % 1024) * 2.0 ** -20, np.ones(n), np.ones(n)])
    y0l = np.ascontiguousarray(y0 if layout == 0 else y0.T)
    ts = np.array([-0.1, 0.0, 0.2, 0.3])
    opt = nn.newODEoptions()
    out = np.empty((len(ts),) + y0l.shape)
    t_out = np.empty(len(ts))
    ny = np.empty(n, dtype=np.int32)
    st = nn.ode.Stats()
    p = np.array([10.0, 28.0, 8.0 / 3.0])
    dp = C.POINTER(C.c_double)
    rc = L.nnhip_ode_solve_batch_multi_gpu_f64(C.byref(opt), 1, 2, p.ctypes.data_as(dp), 3, y0l.ctypes.data, n, dim, layout,
                                               ts.ctypes.data_as(dp), len(ts), t_out.ctypes.data_as(dp), out.ctypes.data, ny.ctypes.data, 0,
                                               C.byref(st), 1)
    assert rc == 0, nn._lib.last_error()
    ref = O.solve_ode_batch(O.RHS_LORENZ, list(p), y0l, n, dim, ts, O.new_options(), "dopri54", layout=layout, n_threads=8)
    assert np.array_equal(t_out, ref["t"]) and _same_bits(out, ref["y"])
    assert np.array_equal(ny, ref["ny"]) and st.steps_total == int(ref["steps"].sum())
    assert L.nnhip_ode_solve_batch_multi_gpu_f64(C.byref(opt), 1, 2, p.ctypes.data_as(dp), 3, y0l.ctypes.data, n, dim, layout,
                                                 ts.ctypes.data_as(dp), len(ts), t_out.ctypes.data_as(dp), out.ctypes.data, ny.ctypes.data, 0,
                                                 C.byref(st), 9) != 0  # more GPUs than the box has -> refused
    # a run-time compiled right-hand side through the same entry (its code object is loaded per device by the worker threads)
    f = nn.Rhs.custom(3, "dy[0] = p[0] * (y[1] - y[0]); dy[1] = y[0] * (p[1] - y[2]) - y[1]; dy[2] = y[0] * y[1] - p[2] * y[2];",
                      keys=("sigma", "rho", "beta"), name="lorenz_mg")
    out2 = np.empty_like(out)
    rc = L.nnhip_ode_solve_batch_multi_gpu_f64(C.byref(opt), 1, f.kind, p.ctypes.data_as(dp), 3, y0l.ctypes.data, n, dim, layout,
                                               ts.ctypes.data_as(dp), len(ts), t_out.ctypes.data_as(dp), out2.ctypes.data, ny.ctypes.data, 0,
                                               C.byref(st), 1)
    assert rc == 0, nn._lib.last_error()
    assert np.array_equal(out2, out, equal_nan=True)  # same expressions as the compiled-in Lorenz system: same bits


def test_fp_contract_opt_in_stays_within_north_star_tolerance(nn, oracle, dev):
    """Tuning knob "fp_contract" = 1 selects FMA-contracted instantiations of the fused RK4 / DOPRI54 / Tsit54 / Vern65
    kernels (opt-in: not bit-exact).  They must still meet BASELINE.json's tolerances: 1e-10 fixed-step, 1e-6 adaptive."""
    import torch
    O = oracle
    L = nn._lib.lib()
    n = 2048
    y0 = 1.0 + np.arange(n) * 2.0 ** -11
    y0l = np.stack([1.0 + (np.arange(n) % 1024) * 2.0 ** -20, np.ones(n), np.ones(n)])
    dt = 2.0 ** -10
    try:
        L.nnhip_tune_set(b"fp_contract", 1)
        t, y = nn.solveODE(nn.Rhs.neg_y(), torch.from_numpy(y0).to(dev), [0.0, 1000 * dt], nn.newODEoptions(dt=dt), integrator="rk4")
        ref = O.solve_ode_batch(O.RHS_NEG_Y, [], y0, n, 0, [0.0, 1000 * dt], O.new_options(dt=dt), "rk4")
        err = np.abs(y[-1].cpu().numpy() - ref["y"][-1, 0]).max()
        assert 0 < err <= 1e-10, err   # contracted: differs in the last bits, far inside the tolerance
        for integ in ("dopri54", "tsit54", "vern65"):
            t, yl = nn.solveODE(nn.Rhs.lorenz(), torch.from_numpy(y0l).to(dev), [0.0, 1.0], integrator=integ)
            refl = O.solve_ode_batch(O.RHS_LORENZ, LOR, y0l, n, 3, [0.0, 1.0], O.new_options(), integ, n_threads=8)
            assert np.abs(yl.cpu().numpy() - refl["y"]).max() <= TOL_ADAPTIVE
    finally:
        L.nnhip_tune_set(b"fp_contract", 0)
    t, y2 = nn.solveODE(nn.Rhs.neg_y(), torch.from_numpy(y0).to(dev), [0.0, 1000 * dt], nn.newODEoptions(dt=dt), integrator="rk4")
    assert np.array_equal(y2[-1].cpu().numpy(), ref["y"][-1, 0])  # default build is bit-exact again


def test_host_entry_chunked_pipeline_is_bitwise_identical(nn, dev):
    """Tuning knobs host_chunks / host_register of the host-pointer solve only change how the batch flows over PCIe."""
    L = nn._lib.lib()
    n = 10_007
    y0 = _lorenz_y0(n)
    ts = [-0.1, 0.0, 0.15, 0.3]
    ref = None
    try:
        for chunks, reg in ((1, 0), (3, 0), (7, 1), (64, 1)):
            L.nnhip_tune_set(b"host_chunks", chunks)
            L.nnhip_tune_set(b"host_register", reg)
            for layout in (0, 1):
                st = nn.ode.Stats()
                y0l = y0 if layout == 0 else np.ascontiguousarray(y0.T)
                t, y, cnt = nn.solveODE(nn.Rhs.lorenz(), y0l, ts, integrator="tsit54", layout=layout, stats=st, return_counts=True)
                yy = y if layout == 0 else np.ascontiguousarray(np.transpose(y, (0, 2, 1)))
                if ref is None:
                    ref = (yy.copy(), cnt["steps"].copy(), st.steps_total)
                assert np.array_equal(yy, ref[0]) and np.array_equal(cnt["steps"], ref[1]) and st.steps_total == ref[2]
                assert st.steps_total == int(cnt["steps"].sum()) and st.ny_min == 4
    finally:
        L.nnhip_tune_set(b"host_chunks", 0)
        L.nnhip_tune_set(b"host_register", 0)


@pytest.mark.parametrize("integrator", ["dopri54", "tsit54", "vern65", "bs32", "rk21"])
def test_adaptive_stream_driver_equals_fused(nn, oracle, dev, integrator):
    """nnhip_ode_adaptive_stream_f64_dev: ODESolver's adaptive loop with (y, FSAL, t, dt) resident in HBM between launches
    must give the bits of the fused solve (and so match the oracle) — rejections included."""
    import torch
    O = oracle
    n = 3000
    y0 = _lorenz_y0(n)
    kw = dict(absTol=1e-6, relTol=1e-6, dtMin=1e-9, dtMax=1.0)
    yt = torch.from_numpy(y0).to(dev)
    t, yf = nn.solveODE(nn.Rhs.lorenz(), yt, [0.0, 1.5], nn.newODEoptions(**kw), integrator=integrator)
    ys, launches = nn.adaptiveStream(nn.Rhs.lorenz(), yt.clone(), 0.0, 1.5, nn.newODEoptions(**kw), integrator=integrator, check_every=5)
    assert torch.equal(ys, yf[-1])
    ref = O.solve_ode_batch(O.RHS_LORENZ, LOR, y0, n, 3, [0.0, 1.5], O.new_options(**kw), integrator, n_threads=8)
    assert _same_bits(ys.cpu().numpy(), ref["y"][-1])
    # one loop iteration per launch, polled every 5, with the next group of 5 always enqueued before the host waits
    assert int(ref["steps"].max()) <= launches < int(ref["steps"].max()) + 10


def test_adaptive_stream_with_runtime_compiled_right_hand_sides(nn, dev):
    """The HBM-resident adaptive loop also runs right-hand sides instantiated at run time: a user system from source and a
    built-in kind at a size without an ahead-of-time kernel; bitwise equal to the fused solve.  Lanes-per-system kinds are refused."""
    import torch
    n = 2000
    rng = np.random.default_rng(9)
    kw = dict(absTol=1e-7, relTol=1e-7, dtMin=1e-9, dtMax=0.5)
    duff = nn.Rhs.custom(2, "dy[0] = y[1]; dy[1] = ((-p[0] * y[1] - p[1] * y[0]) - p[2] * (y[0] * y[0] * y[0])) + p[3] * t;",
                         keys=("delta", "alpha", "beta", "gamma"), defaults=dict(delta=0.2, alpha=1.0, beta=0.5, gamma=0.3), name="duffing_stream")
    for f, dim in ((duff, 2), (nn.Rhs.linear(-0.7), 5), (nn.Rhs.ring(0.1), 6)):
        y0 = torch.from_numpy(0.5 + rng.random((dim, n))).to(dev)
        for integ in ("dopri54", "bs32"):
            t, yf = nn.solveODE(f, y0, [0.0, 2.0], nn.newODEoptions(**kw), integrator=integ)
            ys, launches = nn.adaptiveStream(f, y0.clone(), 0.0, 2.0, nn.newODEoptions(**kw), integrator=integ)
            assert torch.equal(ys, yf[-1]) and launches > 0, (dim, integ)


@pytest.mark.parametrize("graph", [True, False], ids=["graph", "eager"])
@pytest.mark.parametrize("integrator", ["tsit54", "dopri54", "vern65", "bs32", "rk21"])
def test_adaptive_stream_lanes_per_system_equals_fused(nn, oracle, dev, integrator, graph):
    """C4's streamed form: the HBM-resident adaptive loop over Vector[float] states (advance_lps_kernel, ode.nim:525-541 over
    Vector[float]) — 16 lanes per system ahead of time, other sizes instantiated at run time — gives the bits of the fused
    solve and of the oracle, with hipGraph replay of the polling groups (side stream) and with eager launches (default stream)."""
    import torch
    O = oracle
    kw = dict(absTol=1e-8, relTol=1e-8, dtMin=1e-7, dtMax=0.25)
    for dim, n, layout in ((16, 1000, 1), (16, 777, 0), (8, 300, 1), (24, 100, 1), (100, 40, 1)):
        y0 = _ring_y0(n, dim)
        y0l = y0 if layout == 1 else np.ascontiguousarray(y0.T)
        yt = torch.from_numpy(y0l).to(dev)
        t, yf = nn.solveODE(nn.Rhs.ring(0.1), yt, [0.0, 1.0], nn.newODEoptions(**kw), integrator=integrator, layout=layout)
        side = torch.cuda.Stream() if graph else torch.cuda.current_stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for rep in range(2):  # the second call replays the cached graph
                ys, launches = nn.adaptiveStream(nn.Rhs.ring(0.1), yt.clone(), 0.0, 1.0, nn.newODEoptions(**kw), integrator=integrator,
                                                 layout=layout, check_every=4)
                side.synchronize()
                assert torch.equal(ys, yf[-1]), (dim, layout, rep)
        if dim == 16:
            ref = O.solve_ode_batch(O.RHS_RING, [0.1], y0l, n, dim, [0.0, 1.0], O.new_options(**kw), integrator, layout=layout, n_threads=8)
            assert _same_bits(ys.cpu().numpy(), ref["y"][-1])
            assert int(ref["steps"].max()) <= launches < int(ref["steps"].max()) + 8


def test_adaptive_stream_graph_replay_thread_per_ivp(nn, oracle, dev):
    """Graph-replayed polling groups (tuning knob "stream_graph" = 1, non-default stream; eager launches are the default since round 3) for
    the thread-per-IVP and the lanes-per-system advance kernels and for the dense streaming driver: bits of the fused solve, call after call
    (the second and third replay the cached graphs)."""
    import torch
    L = nn._lib.lib()
    n = 5000
    yt = torch.from_numpy(_lorenz_y0(n)).to(dev)
    yr = torch.from_numpy(_ring_y0(700, 16)).to(dev)
    ts = [0.0, 0.25, 0.5, 1.0]
    try:
        for knob in (1, 2):
            assert L.nnhip_tune_set(b"stream_graph", knob) == 0
            for integ in ("dopri54", "tsit54"):
                t, yf = nn.solveODE(nn.Rhs.lorenz(), yt, [0.0, 1.0], integrator=integ)
                td, yd = nn.solveODE(nn.Rhs.lorenz(), yt, ts, integrator=integ)
                tr, yrf = nn.solveODE(nn.Rhs.ring(0.1), yr, [0.0, 1.0], integrator=integ, layout=1)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for rep in range(3):
                        ys, launches = nn.adaptiveStream(nn.Rhs.lorenz(), yt.clone(), 0.0, 1.0, integrator=integ)
                        side.synchronize()
                        assert torch.equal(ys, yf[-1]) and launches >= 102
                        t2, y2, ny, launches = nn.adaptiveStreamSolve(nn.Rhs.lorenz(), yt, ts, integrator=integ)
                        side.synchronize()
                        assert torch.equal(y2, yd)
                        ys, launches = nn.adaptiveStream(nn.Rhs.ring(0.1), yr.clone(), 0.0, 1.0, integrator=integ, layout=1)
                        side.synchronize()
                        assert torch.equal(ys, yrf[-1])
    finally:
        assert L.nnhip_tune_set(b"stream_graph", 2) == 0


@pytest.mark.parametrize("mode", [0, 1])
def test_adaptive_stream_fsal_carried_or_reevaluated(nn, dev, mode):
    """Tuning knob "adv_recompute_fsal": the streaming loops of DOPRI54 / Tsit54 either carry FSAL through HBM (0: the IntegratorProc
    signature as the reference passes it, ode.nim:38) or re-evaluate it as f(t, y) at the start of each launch (1: it IS the last
    stage f(t + dt, yNew) of the step before, ode.nim:299-305 / 362-374).  Both must give the bits of the fused solve — thread-per-IVP
    and lanes-per-system kernels (2 and 4 components per lane), a NON-autonomous run-time compiled right-hand side (t + dt*c_7 with
    c_7 = 1 must be the t the next launch reads), K iterations per launch, the dense streaming driver, and a workspace that is only
    8-byte aligned (two columns for t and dt instead of one of pairs).  Vern65 / BS32 / RK21 ignore the knob."""
    import ctypes as C
    import torch
    L = nn._lib.lib()
    assert L.nnhip_tune_set(b"adv_recompute_fsal", mode) == 0
    try:
        n = 2001
        yt = torch.from_numpy(_lorenz_y0(n)).to(dev)
        kw = dict(absTol=1e-6, relTol=1e-6, dtMin=1e-9, dtMax=1.0)
        for integ in ("dopri54", "tsit54", "vern65", "bs32", "rk21"):
            t, yf = nn.solveODE(nn.Rhs.lorenz(), yt, [0.0, 1.5], nn.newODEoptions(**kw), integrator=integ)
            for K in (1, 3):
                ys, launches = nn.adaptiveStream(nn.Rhs.lorenz(), yt.clone(), 0.0, 1.5, nn.newODEoptions(**kw), integrator=integ, check_every=3, steps_per_launch=K)
                assert torch.equal(ys, yf[-1]), (integ, K)
            ts = [0.0, 0.2, 0.20001, 0.9, 1.5]
            t2, yd, ny, launches = nn.adaptiveStreamSolve(nn.Rhs.lorenz(), yt, ts, nn.newODEoptions(**kw), integrator=integ)
            assert torch.equal(yd, nn.solveODE(nn.Rhs.lorenz(), yt, ts, nn.newODEoptions(**kw), integrator=integ)[1]), integ
        assert L.nnhip_tune_set(b"adv_steps_per_launch", 1) == 0
        kw = dict(absTol=1e-8, relTol=1e-8, dtMin=1e-7, dtMax=0.25)
        for dim, n, layout in ((16, 1000, 1), (16, 777, 0), (8, 500, 1), (24, 100, 1)):   # 24: run-time compiled lanes-per-system kernel
            y0 = _ring_y0(n, dim)
            yl = torch.from_numpy(y0 if layout == 1 else np.ascontiguousarray(y0.T)).to(dev)
            for integ in ("tsit54", "dopri54"):
                t, yf = nn.solveODE(nn.Rhs.ring(0.1), yl, [0.0, 1.0], nn.newODEoptions(**kw), integrator=integ, layout=layout)
                ys, launches = nn.adaptiveStream(nn.Rhs.ring(0.1), yl.clone(), 0.0, 1.0, nn.newODEoptions(**kw), integrator=integ, layout=layout, check_every=4)
                assert torch.equal(ys, yf[-1]), (dim, layout, integ)
            ts = [0.0, 0.3, 1.0]
            t2, yd, ny, launches = nn.adaptiveStreamSolve(nn.Rhs.ring(0.1), yl, ts, nn.newODEoptions(**kw), integrator="tsit54", layout=layout)
            assert torch.equal(yd, nn.solveODE(nn.Rhs.ring(0.1), yl, ts, nn.newODEoptions(**kw), integrator="tsit54", layout=layout)[1]), (dim, layout)
        duff = nn.Rhs.custom(2, "dy[0] = y[1]; dy[1] = ((-p[0] * y[1] - p[1] * y[0]) - p[2] * (y[0] * y[0] * y[0])) + p[3] * t;",
                             keys=("delta", "alpha", "beta", "gamma"), defaults=dict(delta=0.2, alpha=1.0, beta=0.5, gamma=0.3), name="duffing_fsal_modes")
        y0 = torch.from_numpy(0.5 + np.random.default_rng(9).random((2, 2000))).to(dev)
        o = nn.newODEoptions(absTol=1e-7, relTol=1e-7, dtMin=1e-9, dtMax=0.5, tStart=0.125)
        for integ in ("dopri54", "tsit54"):
            t, yf = nn.solveODE(duff, y0, [0.125, 2.0], o, integrator=integ)
            ys, launches = nn.adaptiveStream(duff, y0.clone(), 0.125, 2.0, o, integrator=integ)
            assert torch.equal(ys, yf[-1]), integ
            ts = [-0.3, 0.125, 0.5, 2.0]
            t2, yd, ny, launches = nn.adaptiveStreamSolve(duff, y0, ts, o, integrator=integ)
            assert torch.equal(yd, nn.solveODE(duff, y0, ts, o, integrator=integ)[1]), integ
        # a workspace at an odd multiple of 8 bytes: (t, dt) fall back to two columns
        yt = torch.from_numpy(_lorenz_y0(n)).to(dev)
        opt = nn.newODEoptions(absTol=1e-6, relTol=1e-6, dtMin=1e-9, dtMax=1.0)
        yf = nn.solveODE(nn.Rhs.lorenz(), yt, [0.0, 1.0], opt, integrator="dopri54")[1][-1]
        wsb = int(L.nnhip_ode_adaptive_stream_workspace_bytes(n, 3))
        ws = torch.empty(wsb + 16, dtype=torch.uint8, device=dev)
        y = yt.clone()
        nl = C.c_int64(0)
        p = (C.c_double * 3)(10.0, 28.0, 8.0 / 3.0)
        rc = L.nnhip_ode_adaptive_stream_f64_dev(C.byref(opt), nn.ode.integrator_id("dopri54"), nn.Rhs.lorenz().kind, p, 3, n, 3, 0, 0.0, 1.0, y.data_ptr(),
                                                 ws.data_ptr() + 8, wsb, 4, 0, C.byref(nl), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, nn._lib.last_error()
        torch.cuda.synchronize()
        assert torch.equal(y, yf)
    finally:
        assert L.nnhip_tune_set(b"adv_recompute_fsal", -1) == 0
        assert L.nnhip_tune_set(b"adv_steps_per_launch", 1) == 0
    assert L.nnhip_tune_set(b"adv_recompute_fsal", 2) != 0 and L.nnhip_tune_set(b"adv_recompute_fsal", -2) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("split", [2, 4])
def test_adaptive_stream_split_ranges(nn, dev, split):
    """Tuning knob "adv_split": the streaming loop cut into 2 or 4 index ranges on separate streams.  Range r >= 1 must address ITS slice
    of the packed (t, dt) pairs (t_io advanced by two doubles per IVP, dt_io staying null) as well as of the two-column layout an
    8-byte-aligned workspace falls back to — same bits as the fused solve either way (ode.nim:525-541 per IVP)."""
    import ctypes as C
    import torch
    L = nn._lib.lib()
    assert L.nnhip_tune_set(b"adv_split", split) == 0
    try:
        n = 3001
        yt = torch.from_numpy(_lorenz_y0(n)).to(dev)
        opt = nn.newODEoptions(absTol=1e-6, relTol=1e-6, dtMin=1e-9, dtMax=1.0)
        for integ in ("dopri54", "tsit54", "bs32"):
            yf = nn.solveODE(nn.Rhs.lorenz(), yt, [0.0, 1.25], opt, integrator=integ)[1][-1]
            ys, launches = nn.adaptiveStream(nn.Rhs.lorenz(), yt.clone(), 0.0, 1.25, opt, integrator=integ, check_every=3)
            assert torch.equal(ys, yf), integ
        y0 = _ring_y0(777, 16)
        yl = torch.from_numpy(y0).to(dev)  # [n, dim]: AoS
        o2 = nn.newODEoptions(absTol=1e-8, relTol=1e-8, dtMin=1e-7, dtMax=0.25)
        yf = nn.solveODE(nn.Rhs.ring(0.1), yl, [0.0, 1.0], o2, integrator="tsit54", layout=1)[1][-1]
        ys, launches = nn.adaptiveStream(nn.Rhs.ring(0.1), yl.clone(), 0.0, 1.0, o2, integrator="tsit54", layout=1, check_every=4)
        assert torch.equal(ys, yf)
        # two-column (t, dt): a workspace at an odd multiple of 8 bytes
        yf = nn.solveODE(nn.Rhs.lorenz(), yt, [0.0, 1.0], opt, integrator="dopri54")[1][-1]
        wsb = int(L.nnhip_ode_adaptive_stream_workspace_bytes(n, 3))
        ws = torch.empty(wsb + 16, dtype=torch.uint8, device=dev)
        y = yt.clone()
        nl = C.c_int64(0)
        p = (C.c_double * 3)(10.0, 28.0, 8.0 / 3.0)
        rc = L.nnhip_ode_adaptive_stream_f64_dev(C.byref(opt), nn.ode.integrator_id("dopri54"), nn.Rhs.lorenz().kind, p, 3, n, 3, 0, 0.0, 1.0, y.data_ptr(),
                                                 ws.data_ptr() + 8, wsb, 4, 0, C.byref(nl), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, nn._lib.last_error()
        torch.cuda.synchronize()
        assert torch.equal(y, yf)
    finally:
        assert L.nnhip_tune_set(b"adv_split", 0) == 0


def test_adaptive_stream_edge_shapes(nn, dev):
    """The streaming drivers at the edges of their index arithmetic: one IVP, odd batch sizes (the (t, dt) pairs and the 16-byte state accesses
    must not assume an even N), a scalar state, one launch per polling group, an empty integration span, and a launch limit of one —
    thread-per-IVP and lanes-per-system kernels, both FSAL modes (default and carried), bits of the fused solve."""
    import warnings
    import torch
    L = nn._lib.lib()
    kw = dict(absTol=1e-7, relTol=1e-7, dtMin=1e-8, dtMax=0.5)
    try:
        for mode in (-1, 0):
            assert L.nnhip_tune_set(b"adv_recompute_fsal", mode) == 0
            for n in (1, 2, 3, 64, 65, 258):
                yl = torch.from_numpy(_lorenz_y0(n)).to(dev)
                yr = torch.from_numpy(_ring_y0(n, 16)).to(dev)
                ys1 = torch.from_numpy(1.0 + np.arange(n) * 2.0 ** -8).to(dev)
                for f, y0, layout in ((nn.Rhs.lorenz(), yl, 0), (nn.Rhs.ring(0.1), yr, 1), (nn.Rhs.linear(-0.7), ys1, 0)):
                    for integ in ("dopri54", "tsit54", "bs32"):
                        opt = nn.newODEoptions(**kw)
                        yf = nn.solveODE(f, y0, [0.0, 0.8], opt, integrator=integ, layout=layout)[1][-1]
                        ys, launches = nn.adaptiveStream(f, y0.clone(), 0.0, 0.8, opt, integrator=integ, layout=layout, check_every=1)
                        assert torch.equal(ys, yf), (mode, n, integ, layout)
                        ts = [0.0, 0.05, 0.8]
                        t2, yd, ny, l2 = nn.adaptiveStreamSolve(f, y0, ts, opt, integrator=integ, layout=layout, check_every=1)
                        assert torch.equal(yd, nn.solveODE(f, y0, ts, opt, integrator=integ, layout=layout)[1]), (mode, n, integ, layout)
            # nothing to integrate: the state comes back untouched, no launch is issued
            y = yl.clone()
            ys, launches = nn.adaptiveStream(nn.Rhs.lorenz(), y, 1.0, 1.0, nn.newODEoptions(**kw), integrator="dopri54")
            assert torch.equal(ys, yl) and launches == 0
            # a launch limit of one: exactly the fused solve's max_steps = 1
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                t2, yd, ny, l2 = nn.adaptiveStreamSolve(nn.Rhs.lorenz(), yl, [0.0, 0.5, 1.0], nn.newODEoptions(**kw), integrator="tsit54", max_launches=1)
            tf, yf, cf = nn.solveODE(nn.Rhs.lorenz(), yl, [0.0, 0.5, 1.0], nn.newODEoptions(**kw), integrator="tsit54", max_steps=1, return_counts=True)
            assert l2 == 1 and torch.equal(ny, cf["ny"]) and torch.equal(torch.nan_to_num(yd, nan=-7.0), torch.nan_to_num(yf, nan=-7.0)) and len(w) == 1
        # a batch whose members finish at very different launches (finished IVPs next to live ones in every wave)
        assert L.nnhip_tune_set(b"adv_recompute_fsal", -1) == 0
        rng = np.random.default_rng(12)
        n = 4096
        yh = torch.from_numpy(np.stack([rng.uniform(-15.0, 15.0, n), rng.uniform(-15.0, 15.0, n), rng.uniform(5.0, 40.0, n)])).to(dev)
        for integ in ("dopri54", "tsit54", "bs32", "rk21"):
            opt = nn.newODEoptions(absTol=1e-6, relTol=1e-6, dtMin=1e-9, dtMax=0.5)
            yf, cf = nn.solveODE(nn.Rhs.lorenz(), yh, [0.0, 0.6], opt, integrator=integ, return_counts=True)[1:]
            ys, launches = nn.adaptiveStream(nn.Rhs.lorenz(), yh.clone(), 0.0, 0.6, opt, integrator=integ, check_every=2)
            assert torch.equal(ys, yf[-1]), integ
            assert int(cf["steps"].max()) > int(cf["steps"].min())   # neighbours really finish at different launches
    finally:
        assert L.nnhip_tune_set(b"adv_recompute_fsal", -1) == 0


@pytest.mark.parametrize("K", [2, 3, 7, 1000])
def test_adaptive_stream_several_iterations_per_launch(nn, oracle, dev, K):
    """Tuning knob "adv_steps_per_launch": K iterations of ode.nim:525-541 per IVP and launch, state in registers in between.  Same
    operations in the same order as K launches, so the bits of the fused solve — thread-per-IVP and lanes-per-system kernels, a
    run-time compiled right-hand side, rejections, IVPs that finish in the middle of a launch — with about 1/K of the launches."""
    import torch
    O = oracle
    L = nn._lib.lib()
    try:
        n = 3001
        y0 = _lorenz_y0(n)
        kw = dict(absTol=1e-6, relTol=1e-6, dtMin=1e-9, dtMax=1.0)
        yt = torch.from_numpy(y0).to(dev)
        for integ in ("dopri54", "tsit54", "vern65", "bs32", "rk21"):
            t, yf, cnt = nn.solveODE(nn.Rhs.lorenz(), yt, [0.0, 1.5], nn.newODEoptions(**kw), integrator=integ, return_counts=True)
            ys, launches = nn.adaptiveStream(nn.Rhs.lorenz(), yt.clone(), 0.0, 1.5, nn.newODEoptions(**kw), integrator=integ, check_every=3,
                                             steps_per_launch=K)
            assert torch.equal(ys, yf[-1]), integ
            need = -(-int(cnt["steps"].max()) // K)   # launches until the slowest IVP is done
            assert need <= launches < need + 6, (integ, launches, need)
        ref = O.solve_ode_batch(O.RHS_LORENZ, LOR, y0, n, 3, [0.0, 1.5], O.new_options(**kw), "rk21", n_threads=8)
        assert _same_bits(ys.cpu().numpy(), ref["y"][-1])
        kw = dict(absTol=1e-8, relTol=1e-8, dtMin=1e-7, dtMax=0.25)
        for dim, n, layout in ((16, 1000, 1), (16, 777, 0), (24, 100, 1)):   # 24: run-time compiled lanes-per-system kernel
            y0 = _ring_y0(n, dim)
            y0l = y0 if layout == 1 else np.ascontiguousarray(y0.T)
            yt = torch.from_numpy(y0l).to(dev)
            t, yf = nn.solveODE(nn.Rhs.ring(0.1), yt, [0.0, 1.0], nn.newODEoptions(**kw), integrator="tsit54", layout=layout)
            ys, launches = nn.adaptiveStream(nn.Rhs.ring(0.1), yt.clone(), 0.0, 1.0, nn.newODEoptions(**kw), integrator="tsit54", layout=layout,
                                             check_every=4, steps_per_launch=K)
            assert torch.equal(ys, yf[-1]), (dim, layout)
        duff = nn.Rhs.custom(2, "dy[0] = y[1]; dy[1] = ((-p[0] * y[1] - p[1] * y[0]) - p[2] * (y[0] * y[0] * y[0])) + p[3] * t;",
                             keys=("delta", "alpha", "beta", "gamma"), defaults=dict(delta=0.2, alpha=1.0, beta=0.5, gamma=0.3), name="duffing_stream_k")
        y0 = torch.from_numpy(0.5 + np.random.default_rng(9).random((2, 2000))).to(dev)
        t, yf = nn.solveODE(duff, y0, [0.0, 2.0], nn.newODEoptions(absTol=1e-7, relTol=1e-7, dtMin=1e-9, dtMax=0.5), integrator="dopri54")
        ys, launches = nn.adaptiveStream(duff, y0.clone(), 0.0, 2.0, nn.newODEoptions(absTol=1e-7, relTol=1e-7, dtMin=1e-9, dtMax=0.5),
                                         integrator="dopri54", steps_per_launch=K)
        assert torch.equal(ys, yf[-1])
    finally:
        assert L.nnhip_tune_set(b"adv_steps_per_launch", 1) == 0
    assert L.nnhip_tune_set(b"adv_steps_per_launch", 0) != 0 and L.nnhip_tune_set(b"adv_steps_per_launch", 1025) != 0


@pytest.mark.parametrize("layout,dim", [(0, 1), (0, 3), (1, 3)])
def test_rccl_allgather_states_single_device(nn, dev, layout, dim):
    """nnhip_allgather_states_f64_dev (one process, G devices, RCCL): with the one device of this box the gather must
    reproduce the shard as the full tensor (plane-by-plane for SoA)."""
    import ctypes as C
    import torch
    L = nn._lib.lib()
    n = 12345
    shard = torch.randn((dim, n) if layout == 0 else (n, dim), dtype=torch.float64, device=dev)
    if dim == 1 and layout == 0:
        shard = shard.reshape(n)
    full = torch.zeros_like(shard)
    sp = (C.c_void_p * 1)(shard.data_ptr())
    fp = (C.c_void_p * 1)(full.data_ptr())
    cnt = (C.c_int64 * 1)(n)
    torch.cuda.synchronize()
    rc = L.nnhip_allgather_states_f64_dev(1, sp, cnt, dim, layout, fp, None)
    assert rc == 0, L.nnhip_multigpu_last_error().decode()
    torch.cuda.synchronize()
    assert torch.equal(full, shard)
    assert L.nnhip_allgather_states_f64_dev(5, sp, cnt, dim, layout, fp, None) != 0  # more devices than the box has


@pytest.mark.parametrize("integrator", ["rk4", "dopri54", "tsit54"])
def test_parameter_sweep_per_ivp_params(nn, oracle, dev, integrator):
    """nnhip_ode_solve_batch_sweep_f64_dev: every IVP carries its own RHS parameters (its own ctx in the reference's terms).
    Thread-per-IVP (Lorenz: per-IVP rho and sigma) and lanes-per-system (16-dim ring: per-IVP coupling) kernels vs one
    oracle solve per IVP with that IVP's parameters."""
    import torch
    O = oracle
    n = 257
    rng = np.random.default_rng(21)
    y0 = _lorenz_y0(n)
    sigma, rho = rng.uniform(8, 12, n), rng.uniform(20, 35, n)
    sweep = torch.from_numpy(np.stack([sigma, rho])).to(dev)           # first two parameters per IVP; beta stays batch-wide
    ts = [-0.1, 0.0, 0.2, 0.4]
    kw = dict(dt=1e-3, absTol=1e-8, relTol=1e-8, dtMin=1e-7, dtMax=5e-2)
    t, y, cnt = nn.solveODE(nn.Rhs.lorenz(), torch.from_numpy(y0).to(dev), ts, nn.newODEoptions(**kw), integrator=integrator, sweep=sweep,
                            return_counts=True)
    got = y.cpu().numpy()
    for i in range(0, n, 9):
        rt, ry, st = O.solve_ode(O.RHS_LORENZ, [sigma[i], rho[i], 8.0 / 3.0], list(y0[:, i]), ts, O.new_options(**kw), integrator)
        assert _same_bits(got[:, :, i], ry)
        assert int(cnt["steps"][i]) == st.steps
    # lanes-per-system kernel
    y16 = _ring_y0(n)
    csw = rng.uniform(-0.3, 0.3, n)
    t, y = nn.solveODE(nn.Rhs.ring(0.1), torch.from_numpy(y16).to(dev), [0.0, 0.5], nn.newODEoptions(**kw), integrator=integrator, layout=1,
                       sweep=torch.from_numpy(csw[None, :].copy()).to(dev))
    got = y.cpu().numpy()
    for i in range(0, n, 31):
        rt, ry, st = O.solve_ode(O.RHS_RING, [csw[i]], list(y16[i]), [0.0, 0.5], O.new_options(**kw), integrator)
        assert _same_bits(got[:, i, :], ry)


@pytest.fixture(params=[0, 1], ids=["order_array_in_kernel", "physical_reorder"])
def sort_copy(request, nn):
    """Tuning knob "sort_copy": the binned solve follows the order array inside the kernel (0, default) or gathers the batch into integration
    order, solves it with coalesced accesses and brings the results back through the inverse order (1)."""
    L = nn._lib.lib()
    assert L.nnhip_tune_set(b"sort_copy", request.param) == 0
    yield request.param
    assert L.nnhip_tune_set(b"sort_copy", 0) == 0


def test_sort_by_returns_identical_results_in_caller_order(nn, dev, sort_copy):
    """solveODE(sort_by=...) integrates in sorted order (less wavefront divergence) and un-permutes: bit-identical output."""
    import torch
    n = 5000
    rng = np.random.default_rng(8)
    mu = torch.from_numpy(rng.uniform(0.1, 10.0, n)).to(dev)
    y0 = torch.from_numpy(np.stack([rng.uniform(1.5, 2.5, n), np.zeros(n)])).to(dev)
    opt = nn.newODEoptions(absTol=1e-6, relTol=1e-6, dtMin=1e-9, dtMax=1.0)
    ts = [0.0, 1.0, 3.0]
    a = nn.solveODE(nn.Rhs.vanderpol(), y0, ts, opt, integrator="tsit54", sweep=mu[None, :], return_counts=True)
    b = nn.solveODE(nn.Rhs.vanderpol(), y0, ts, opt, integrator="tsit54", sweep=mu[None, :], return_counts=True, sort_by=mu)
    assert torch.equal(a[1], b[1]) and all(torch.equal(a[2][k], b[2][k]) for k in a[2])
    y0a = y0.t().contiguous()
    c = nn.solveODE(nn.Rhs.vanderpol(), y0a, ts, opt, integrator="tsit54", sweep=mu[None, :], layout=1, sort_by=mu)
    assert torch.equal(c[1].permute(0, 2, 1), a[1])
    # automatic two-pass mode (probe solve -> device argsort -> solve in that order), every adaptive integrator, with counters
    # (ranked by the steps still to take — knob "sort_auto_key" 1, the default — or by the probe's progress, 0; a two-sided tspan always uses the latter)
    Lk = nn._lib.lib()
    for integ in ("dopri54", "vern65", "bs32", "rk21"):
        a = nn.solveODE(nn.Rhs.vanderpol(), y0, ts, opt, integrator=integ, sweep=mu[None, :], return_counts=True)
        a2s = nn.solveODE(nn.Rhs.vanderpol(), y0, [-1.0, 0.0, 2.0], opt, integrator=integ, sweep=mu[None, :], return_counts=True)
        for auto_key in (1, 0):
            assert Lk.nnhip_tune_set(b"sort_auto_key", auto_key) == 0
            try:
                d = nn.solveODE(nn.Rhs.vanderpol(), y0, ts, opt, integrator=integ, sweep=mu[None, :], return_counts=True, sort_by="auto")
                d2s = nn.solveODE(nn.Rhs.vanderpol(), y0, [-1.0, 0.0, 2.0], opt, integrator=integ, sweep=mu[None, :], return_counts=True, sort_by="auto")
            finally:
                assert Lk.nnhip_tune_set(b"sort_auto_key", 1) == 0
            assert torch.equal(a[1], d[1]) and all(torch.equal(a[2][k], d[2][k]) for k in a[2]), (integ, auto_key)
            assert torch.equal(torch.nan_to_num(a2s[1], nan=-1.0), torch.nan_to_num(d2s[1], nan=-1.0)), (integ, auto_key)   # (backwards, Van der Pol blows up for the larger mu: NaN rows on both sides)
            assert all(torch.equal(a2s[2][k], d2s[2][k]) for k in a2s[2]), (integ, auto_key)
    assert Lk.nnhip_tune_set(b"sort_auto_key", 2) != 0
    # max_steps smaller than the probe, a fixed-step method (runs unsorted), N = 1 and N = 0
    e0 = nn.solveODE(nn.Rhs.vanderpol(), y0, ts, opt, integrator="tsit54", sweep=mu[None, :], max_steps=5, return_counts=True)
    e1 = nn.solveODE(nn.Rhs.vanderpol(), y0, ts, opt, integrator="tsit54", sweep=mu[None, :], max_steps=5, return_counts=True, sort_by="auto")
    assert torch.equal(e0[2]["steps"], e1[2]["steps"]) and bool((e1[2]["steps"] == 5).all())
    assert torch.equal(torch.nan_to_num(e0[1], nan=-1.0), torch.nan_to_num(e1[1], nan=-1.0))
    f0 = nn.solveODE(nn.Rhs.vanderpol(), y0, ts, nn.newODEoptions(dt=1e-2), integrator="rk4", sweep=mu[None, :])
    f1 = nn.solveODE(nn.Rhs.vanderpol(), y0, ts, nn.newODEoptions(dt=1e-2), integrator="rk4", sweep=mu[None, :], sort_by=mu)
    assert torch.equal(f0[1], f1[1])
    for m in (1, 0):
        g0 = nn.solveODE(nn.Rhs.vanderpol(), y0[:, :m].contiguous(), ts, opt, integrator="tsit54", sweep=mu[None, :m].contiguous())
        g1 = nn.solveODE(nn.Rhs.vanderpol(), y0[:, :m].contiguous(), ts, opt, integrator="tsit54", sweep=mu[None, :m].contiguous(), sort_by="auto")
        assert torch.equal(g0[1], g1[1])
    # keys within 5 % of each other (and constant / all-NaN / negative ones): the batch runs in the caller's order — same bits either way, and
    # with the check switched off (knob 0: always sort) too
    L = nn._lib.lib()
    a = nn.solveODE(nn.Rhs.vanderpol(), y0, ts, opt, integrator="tsit54", sweep=mu[None, :], return_counts=True)
    wild = torch.from_numpy(np.concatenate([10.0 ** rng.uniform(-300, 300, n - 8) * rng.choice([-1.0, 1.0], n - 8),   # 600 decades, both signs, and every special value
                                            [np.inf, -np.inf, np.nan, 0.0, -0.0, 5e-324, -5e-324, 1.7976931348623157e308]])).to(dev)
    for key in (wild, 10.0 ** (6.0 * torch.rand(n, dtype=torch.float64, device=dev)),
                100.0 + torch.rand(n, dtype=torch.float64, device=dev), torch.full((n,), 3.0, dtype=torch.float64, device=dev),
                torch.full((n,), float("nan"), dtype=torch.float64, device=dev), -100.0 - 2.0 * torch.rand(n, dtype=torch.float64, device=dev),
                torch.zeros(n, dtype=torch.float64, device=dev)):
        for permille in (50, 0):
            assert L.nnhip_tune_set(b"sort_min_spread_permille", permille) == 0
            try:
                h = nn.solveODE(nn.Rhs.vanderpol(), y0, ts, opt, integrator="tsit54", sweep=mu[None, :], return_counts=True, sort_by=key)
            finally:
                assert L.nnhip_tune_set(b"sort_min_spread_permille", 50) == 0
            assert torch.equal(a[1], h[1]) and all(torch.equal(a[2][k], h[2][k]) for k in a[2])
    assert L.nnhip_tune_set(b"sort_min_spread_permille", 1001) != 0
    # automatic mode on a forward 2-point tspan: the sorted pass RESUMES from the probe's state (t, dt, y) instead of restarting — rows, row count and both
    # counters must be those of the plain solve: every adaptive method (Vern65 restarts: its FSAL is not f(t, y)), IVPs that arrive within the probe, a
    # max_steps cap above and below the probe, per-IVP parameters, both layouts, a probe of another length, and the knob switched off
    tsf = [0.0, 2.0]
    y0a = y0.t().contiguous()
    assert L.nnhip_tune_set(b"sort_resume", 1) == 0   # (not the default: within 1 % either way, see ode_capi.hip)
    for integ in ("dopri54", "tsit54", "bs32", "rk21", "vern65"):
        for kw in (dict(), dict(max_steps=30), dict(max_steps=5), dict(probe_steps=3)):
            kw0 = {k: v for k, v in kw.items() if k != "probe_steps"}
            a2 = nn.solveODE(nn.Rhs.vanderpol(), y0, tsf, opt, integrator=integ, sweep=mu[None, :], return_counts=True, **kw0)
            d2 = nn.solveODE(nn.Rhs.vanderpol(), y0, tsf, opt, integrator=integ, sweep=mu[None, :], return_counts=True, sort_by="auto", **kw)
            assert np.array_equal(a2[0], d2[0]) and torch.equal(torch.nan_to_num(a2[1], nan=-1.0), torch.nan_to_num(d2[1], nan=-1.0)), (integ, kw)
            assert all(torch.equal(a2[2][k], d2[2][k]) for k in a2[2]), (integ, kw)
    short = nn.solveODE(nn.Rhs.vanderpol(), y0, [0.0, 1e-3], opt, integrator="tsit54", sweep=mu[None, :], return_counts=True)           # done within 8 steps
    short2 = nn.solveODE(nn.Rhs.vanderpol(), y0, [0.0, 1e-3], opt, integrator="tsit54", sweep=mu[None, :], return_counts=True, sort_by="auto")
    assert torch.equal(short[1], short2[1]) and all(torch.equal(short[2][k], short2[2][k]) for k in short[2])
    e = nn.solveODE(nn.Rhs.vanderpol(), y0a, tsf, opt, integrator="dopri54", sweep=mu[None, :], layout=1, return_counts=True)
    e2 = nn.solveODE(nn.Rhs.vanderpol(), y0a, tsf, opt, integrator="dopri54", sweep=mu[None, :], layout=1, return_counts=True, sort_by="auto")
    assert torch.equal(e[1], e2[1]) and all(torch.equal(e[2][k], e2[2][k]) for k in e[2])
    o_shift = nn.newODEoptions(absTol=1e-6, relTol=1e-6, dtMin=1e-9, dtMax=1.0, tStart=0.5)
    g = nn.solveODE(nn.Rhs.vanderpol(), y0, [0.5, 2.5], o_shift, integrator="tsit54", sweep=mu[None, :], return_counts=True)
    g2 = nn.solveODE(nn.Rhs.vanderpol(), y0, [2.5, 0.5], o_shift, integrator="tsit54", sweep=mu[None, :], return_counts=True, sort_by="auto")
    assert torch.equal(g[1], g2[1]) and all(torch.equal(g[2][k], g2[2][k]) for k in g[2])
    # ... and re-binned mid-solve (knob "sort_rebin_steps" = S: probe -> S more steps in the probe's order -> binned again by the steps still to take -> the rest):
    # S below and above what the IVPs need, caps on either side of probe + S, the second layout, a span that ends inside the probe, dtMin = 0 (refused either way)
    for S in (1, 10, 60, 100000):
        assert L.nnhip_tune_set(b"sort_rebin_steps", S) == 0
        try:
            for integ in ("dopri54", "tsit54", "bs32", "rk21", "vern65"):
                for kw in (dict(), dict(max_steps=8 + S), dict(max_steps=9 + S), dict(max_steps=30)):
                    a2 = nn.solveODE(nn.Rhs.vanderpol(), y0, tsf, opt, integrator=integ, sweep=mu[None, :], return_counts=True, **kw)
                    d2 = nn.solveODE(nn.Rhs.vanderpol(), y0, tsf, opt, integrator=integ, sweep=mu[None, :], return_counts=True, sort_by="auto", **kw)
                    assert np.array_equal(a2[0], d2[0]) and torch.equal(torch.nan_to_num(a2[1], nan=-1.0), torch.nan_to_num(d2[1], nan=-1.0)), (S, integ, kw)
                    assert all(torch.equal(a2[2][k], d2[2][k]) for k in a2[2]), (S, integ, kw)
            short2 = nn.solveODE(nn.Rhs.vanderpol(), y0, [0.0, 1e-3], opt, integrator="tsit54", sweep=mu[None, :], return_counts=True, sort_by="auto")
            assert torch.equal(short[1], short2[1]) and all(torch.equal(short[2][k], short2[2][k]) for k in short[2])
            e2 = nn.solveODE(nn.Rhs.vanderpol(), y0a, tsf, opt, integrator="dopri54", sweep=mu[None, :], layout=1, return_counts=True, sort_by="auto")
            assert torch.equal(e[1], e2[1]) and all(torch.equal(e[2][k], e2[2][k]) for k in e[2])
            o_zero = nn.newODEoptions(absTol=1e-6, relTol=1e-6, dtMin=0.0, dtMax=1.0)
            z = nn.solveODE(nn.Rhs.vanderpol(), y0, tsf, o_zero, integrator="tsit54", sweep=mu[None, :], return_counts=True, max_steps=50)
            z2 = nn.solveODE(nn.Rhs.vanderpol(), y0, tsf, o_zero, integrator="tsit54", sweep=mu[None, :], return_counts=True, max_steps=50, sort_by="auto")
            assert torch.equal(torch.nan_to_num(z[1], nan=-1.0), torch.nan_to_num(z2[1], nan=-1.0)) and all(torch.equal(z[2][k], z2[2][k]) for k in z[2])
        finally:
            assert L.nnhip_tune_set(b"sort_rebin_steps", 0) == 0
    assert L.nnhip_tune_set(b"sort_rebin_steps", -1) != 0
    assert L.nnhip_tune_set(b"sort_resume", 0) == 0
    d3 = nn.solveODE(nn.Rhs.vanderpol(), y0, tsf, opt, integrator="tsit54", sweep=mu[None, :], return_counts=True, sort_by="auto")
    a3 = nn.solveODE(nn.Rhs.vanderpol(), y0, tsf, opt, integrator="tsit54", sweep=mu[None, :], return_counts=True)
    assert torch.equal(a3[1], d3[1]) and all(torch.equal(a3[2][k], d3[2][k]) for k in a3[2])
    # lanes-per-system kernels: 16-component ring systems with per-system coupling, sorted by the coupling and automatically
    rngc = np.random.default_rng(4)
    y16 = torch.from_numpy(_ring_y0(700)).to(dev)
    csw = torch.from_numpy(rngc.uniform(-3.0, 3.0, 700)).to(dev)
    kw = dict(absTol=1e-8, relTol=1e-8, dtMin=1e-8, dtMax=0.5)
    h0 = nn.solveODE(nn.Rhs.ring(0.1), y16, [0.0, 0.4, 1.0], nn.newODEoptions(**kw), integrator="tsit54", layout=1, sweep=csw[None, :], return_counts=True)
    for key in (csw.abs(), "auto"):
        h1 = nn.solveODE(nn.Rhs.ring(0.1), y16, [0.0, 0.4, 1.0], nn.newODEoptions(**kw), integrator="tsit54", layout=1, sweep=csw[None, :], return_counts=True,
                         sort_by=key)
        assert torch.equal(h0[1], h1[1]) and all(torch.equal(h0[2][k], h1[2][k]) for k in h0[2])
    assert int(h0[2]["steps"].max()) > 2 * int(h0[2]["steps"].min())   # the batch really is heterogeneous
    # host-pointer form (numpy batch, numpy key / "auto"): nnhip_ode_solve_batch_sorted_f64, what a Nim host holding seqs calls
    a = nn.solveODE(nn.Rhs.vanderpol(), y0, ts, opt, integrator="tsit54", sweep=mu[None, :], return_counts=True)
    for key in (mu.cpu().numpy(), "auto"):
        hh = nn.solveODE(nn.Rhs.vanderpol(), y0.cpu().numpy(), ts, opt, integrator="tsit54", sweep=mu[None, :].cpu().numpy(), return_counts=True, sort_by=key)
        assert np.array_equal(hh[1], a[1].cpu().numpy()) and all(np.array_equal(hh[2][k], a[2][k].cpu().numpy()) for k in a[2])


def test_parameter_sweep_through_the_host_pointer_entry(nn, dev):
    """solveODE(numpy y0, sweep=numpy [k, N]) (nnhip_ode_solve_batch_sweep_f64) == the device-pointer sweep bit for bit."""
    import torch
    n = 5000
    rng = np.random.default_rng(21)
    y0 = np.stack([1.0 + rng.random(n), np.ones(n), np.ones(n)])
    sw = np.stack([np.full(n, 10.0), rng.uniform(20, 30, n)])  # per-IVP sigma, rho; beta from the Rhs defaults
    ts = [0.0, 0.3, 0.6]
    th, yh, ch = nn.solveODE(nn.Rhs.lorenz(), y0, ts, nn.newODEoptions(), integrator="tsit54", sweep=sw, return_counts=True)
    td, yd, cd = nn.solveODE(nn.Rhs.lorenz(), torch.from_numpy(y0).to(dev), ts, nn.newODEoptions(), integrator="tsit54", sweep=torch.from_numpy(sw).to(dev),
                             return_counts=True)
    assert np.array_equal(yh, yd.cpu().numpy()) and np.array_equal(ch["steps"], cd["steps"].cpu().numpy())
    plain = nn.solveODE(nn.Rhs.lorenz(), y0, ts, nn.newODEoptions(), integrator="tsit54")[1]
    assert not np.array_equal(plain, yh)  # the sweep really changes the trajectories


@pytest.mark.parametrize("integrator", ["dopri54", "tsit54", "vern65", "bs32", "rk21"])
def test_adaptive_dense_output_through_the_step_streaming_seam(nn, oracle, dev, integrator):
    """nnhip_ode_adaptive_stream_dense_f64_dev: the whole ODESolver driver (ode.nim:471-586) over the HBM-resident advance kernel
    with per-IVP Hermite history — the reference's own harness (tests/test_ode.nim:15: linspace(-10, 10, 100), both directions)
    and the awkward grids (tStart inside / outside / duplicated, unsorted, one point, two points on one side, requested times
    closer than the steps so that the reference drops rows, per IVP differently) — bitwise equal to the fused solve and to the oracle."""
    import torch
    O = oracle
    rng = np.random.default_rng(44)
    ts_h = O.linspace(-10.0, 10.0, 100)
    y0s = torch.from_numpy(1.0 + np.arange(40) * 2.0 ** -6).to(dev)
    opt = nn.newODEoptions(relTol=1e-8, dt=1e-2)
    t, y, ny, launches = nn.adaptiveStreamSolve(nn.Rhs.linear(-0.1), y0s, ts_h, opt, integrator=integrator)
    tf, yf, cf = nn.solveODE(nn.Rhs.linear(-0.1), y0s, ts_h, opt, integrator=integrator, return_counts=True)
    assert np.array_equal(t, ts_h) and torch.equal(y, yf) and torch.equal(ny, cf["ny"]) and bool((ny == 100).all())
    rt, ry, st = O.solve_ode(O.RHS_LINEAR, [-0.1], float(y0s[5]), ts_h, O.new_options(relTol=1e-8, dt=1e-2), integrator)
    assert np.array_equal(y[:, 5].cpu().numpy(), np.asarray(ry))
    # (Lorenz is unstable backwards in time: long backward spans with a tiny dtMin take 1e7+ steps in the reference as well)
    grids = [[0.6, -0.3, 0.0], [0.0, 0.0, 1.0], [-0.3, -0.1], [1.2], [0.5, 1.0], [0.5, 0.25, 0.75, 1.0], [0.3, 0.30001, 0.30002, 0.9],
             [-0.2, -0.20001, -0.1, 0.7, 0.70001, 0.70002], []]
    for f, dim, layout in ((nn.Rhs.lorenz(), 3, 0), (nn.Rhs.lorenz(), 3, 1), (nn.Rhs.vanderpol(2.0), 2, 0), (nn.Rhs.neg_y(), 1, 0)):
        n = 203
        y0 = rng.uniform(0.5, 1.5, (n, dim)) + (np.array([0.0, 0.0, 20.0]) if dim == 3 else 0.0)
        y0l = torch.from_numpy(np.ascontiguousarray(y0 if layout == 1 else y0.T) if dim > 1 else y0[:, 0].copy()).to(dev)
        for ts in grids:
            for tstart, kw in ((0.0, {}), (0.25, dict(absTol=1e-7, relTol=1e-7, dtMin=1e-6, dtMax=0.5))):
                o2 = nn.newODEoptions(tStart=tstart, **kw)
                t, y, ny, launches = nn.adaptiveStreamSolve(f, y0l, ts, o2, integrator=integrator, layout=layout, check_every=3)
                tf, yf, cf = nn.solveODE(f, y0l, ts, o2, integrator=integrator, layout=layout, return_counts=True)
                assert np.array_equal(t, tf), (dim, ts)
                assert torch.equal(ny, cf["ny"]), (dim, layout, ts, tstart)
                assert torch.equal(torch.nan_to_num(y, nan=-7.0), torch.nan_to_num(yf, nan=-7.0)), (dim, layout, ts, tstart)


@pytest.mark.parametrize("integrator", ["dopri54", "vern65", "bs32"])
def test_adaptive_dense_stream_cut_by_max_launches_is_the_fused_max_steps(nn, dev, integrator):
    """max_launches bounds each direction's loop where max_steps bounds the fused solve's: the launch that is the last one permitted
    leaves its emission to the iteration that never comes (StepArgs::emitAfter), so rows, row counts and the NaN fill are the fused
    solve's with max_steps = max_launches — also when the cut is not a multiple of check_every, and on both sides of tStart."""
    import warnings
    import torch
    rng = np.random.default_rng(5)
    n = 301
    y0 = rng.uniform(0.5, 1.5, (n, 3)) + np.array([0.0, 0.0, 20.0])
    y0l = torch.from_numpy(np.ascontiguousarray(y0.T)).to(dev)
    for ts, tstart in (([0.1, 0.2, 0.4, 0.8, 1.6], 0.0), ([-0.2, -0.1, 0.3, 0.30001, 0.9], 0.0), ([0.5, 1.0], 0.25)):
        opt = nn.newODEoptions(tStart=tstart, absTol=1e-7, relTol=1e-7, dtMin=1e-6, dtMax=0.5)
        full = nn.solveODE(nn.Rhs.lorenz(), y0l, ts, opt, integrator=integrator, return_counts=True)[2]["steps"]
        for cut, ce in ((7, 3), (16, 8), (1, 8), (int(full.max()) + 40, 8)):
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                t, y, ny, launches = nn.adaptiveStreamSolve(nn.Rhs.lorenz(), y0l, ts, opt, integrator=integrator, check_every=ce, max_launches=cut)
            tf, yf, cf = nn.solveODE(nn.Rhs.lorenz(), y0l, ts, opt, integrator=integrator, max_steps=cut, return_counts=True)
            assert torch.equal(ny, cf["ny"]), (ts, cut)
            assert torch.equal(torch.nan_to_num(y, nan=-7.0), torch.nan_to_num(yf, nan=-7.0)), (ts, cut)
            assert bool(w) == (cut < int(full.max())), (ts, cut, len(w))


@pytest.mark.parametrize("integrator", ["dopri54", "tsit54", "rk4", "bs32", "heun3"])
def test_every_ivp_its_own_tspan_end(nn, oracle, dev, integrator):
    """nnhip_ode_solve_batch_tend_f64_dev: IVP i is solveODE(f, y0_i, [tStart, t_end[i]]) (each reference call owns its tspan,
    ode.nim:589-591, 476-480): forward, backward and zero-length spans in one batch, thread-per-IVP and lanes-per-system kernels,
    bitwise equal to one oracle call per IVP — rows in the reference's sorted order, ny = 1 where tEnd == tStart."""
    import torch
    O = oracle
    rng = np.random.default_rng(7)
    for f, okind, params, dim, layout in ((nn.Rhs.lorenz(), O.RHS_LORENZ, LOR, 3, 0), (nn.Rhs.ring(0.1), O.RHS_RING, [0.1], 16, 1),
                                          (nn.Rhs.linear(-0.4), O.RHS_LINEAR, [-0.4], 1, 0)):
        n = 333
        y0 = rng.uniform(0.5, 1.5, (n, dim)) + (np.array([0.0, 0.0, 20.0]) if dim == 3 else 0.0)
        te = rng.uniform(-0.3, 1.0, n)
        te[::17] = 0.25          # tEnd == tStart
        te[5] = np.nextafter(0.25, 1.0)
        te[6] = np.nextafter(0.25, -1.0)
        kw = dict(tStart=0.25, dt=1e-2, absTol=1e-7, relTol=1e-7, dtMin=1e-6, dtMax=0.25)
        y0l = np.ascontiguousarray(y0 if layout == 1 else y0.T) if dim > 1 else y0[:, 0].copy()
        y, cnt = nn.solveODEPerIvpEnd(f, torch.from_numpy(y0l).to(dev), torch.from_numpy(te).to(dev), nn.newODEoptions(**kw), integrator=integrator, layout=layout)
        got = y.cpu().numpy()
        ny, steps = cnt["ny"].cpu().numpy(), cnt["steps"].cpu().numpy()
        for i in list(range(0, n, 7)) + [5, 6]:
            yi = list(y0[i]) if dim > 1 else float(y0[i, 0])
            rt, ry, st = O.solve_ode(okind, params, yi, [0.25, te[i]], O.new_options(**kw), integrator)
            gi = got[:, i] if dim == 1 else (got[:, :, i] if layout == 0 else got[:, i, :])
            assert ny[i] == st.n_y and steps[i] == st.steps, (i, te[i])
            assert _same_bits(gi[:st.n_y].reshape(st.n_y, -1), np.asarray(ry).reshape(st.n_y, -1)), (integrator, dim, i, te[i])
            assert np.isnan(gi[st.n_y:]).all()
            if te[i] == 0.25:
                assert st.n_y == 1


@pytest.mark.parametrize("integrator", ["dopri54", "tsit54", "rk4", "vern65"])
def test_every_ivp_its_own_options_and_tspan(nn, oracle, dev, integrator):
    """nnhip_ode_solve_batch_calls_f64_dev: IVP i is solveODE(f, y0_i, [tStart_i, tEnd_i], newODEoptions(dt_i, absTol_i, relTol_i,
    dtMax_i, dtMin_i, tStart = tStart_i)) — each reference call owns its ODEoptions and its tspan (ode.nim:589-591, 26-34, 78-102).
    Bitwise equal to one oracle call per IVP; negative option values go through abs() like newODEoptions; calls the reference
    refuses (dtMax < dtMin) are flagged ny = -1."""
    import torch
    O = oracle
    rng = np.random.default_rng(11)
    for f, okind, params, dim, layout in ((nn.Rhs.lorenz(), O.RHS_LORENZ, LOR, 3, 0), (nn.Rhs.ring(0.1), O.RHS_RING, [0.1], 16, 1)):
        n = 200
        y0 = rng.uniform(0.5, 1.5, (n, dim)) + (np.array([0.0, 0.0, 20.0]) if dim == 3 else 0.0)
        ts = rng.uniform(-0.2, 0.2, n)
        te = ts + rng.uniform(-0.3, 0.8, n)
        at = 10 ** rng.uniform(-9, -4, n) * rng.choice([1.0, -1.0], n)      # sign is dropped by newODEoptions' abs()
        rt_ = 10 ** rng.uniform(-9, -4, n)
        dmax = 10 ** rng.uniform(-2, -0.5, n)
        dmin = 10 ** rng.uniform(-6, -4, n)
        dts = 10 ** rng.uniform(-2.5, -1.5, n)
        dmin[3] = 1.0        # dtMax < dtMin: newODEoptions raises ValueError for this call
        dev_t = lambda a: torch.from_numpy(a).to(dev)
        y0l = np.ascontiguousarray(y0 if layout == 1 else y0.T)
        y, cnt = nn.solveODEPerIvpEnd(f, dev_t(y0l), dev_t(te), nn.newODEoptions(), integrator=integrator, layout=layout, t_start=dev_t(ts), absTol=dev_t(at),
                                      relTol=dev_t(rt_), dtMax=dev_t(dmax), dtMin=dev_t(dmin), dt=dev_t(dts))
        got = y.cpu().numpy()
        ny, steps, rej = (cnt[k].cpu().numpy() for k in ("ny", "steps", "rejected"))
        fixed = integrator in nn.fixedODE
        assert (ny[3] == -1) == (not fixed)       # fixed-step methods never look at dtMax / dtMin
        for i in list(range(0, n, 9)) + [3]:
            if i == 3 and not fixed:
                gi = got[:, :, i] if layout == 0 else got[:, i, :]
                assert np.isnan(gi).all()
                with pytest.raises(ValueError):
                    O.new_options(dt=dts[i], absTol=at[i], relTol=rt_[i], dtMax=dmax[i], dtMin=dmin[i], tStart=ts[i])
                continue
            oi = O.new_options(dt=dts[i], absTol=at[i], relTol=rt_[i], dtMax=dmax[i], dtMin=min(dmin[i], dmax[i]) if fixed else dmin[i], tStart=ts[i])
            rt, ry, st = O.solve_ode(okind, params, list(y0[i]), [ts[i], te[i]], oi, integrator)
            gi = got[:, :, i] if layout == 0 else got[:, i, :]
            assert ny[i] == st.n_y and steps[i] == st.steps and rej[i] == st.rejected, (i, integrator)
            assert _same_bits(gi[:st.n_y].reshape(st.n_y, -1), np.asarray(ry).reshape(st.n_y, -1)), (integrator, dim, i)


@pytest.mark.parametrize("integrator", ["tsit54", "rk4"])
def test_host_form_of_the_per_call_solve(nn, dev, integrator):
    """nnhip_ode_solve_batch_calls_f64 (host arrays, one ODEoptions OBJECT per call) == the device-table entry on the same calls,
    bit for bit; and each call equals the ordinary 1-IVP solveODE with its own tspan and options (ode.nim:589-591)."""
    import torch
    rng = np.random.default_rng(5)
    n = 300
    y0 = np.ascontiguousarray((rng.uniform(0.5, 1.5, (n, 3)) + np.array([0.0, 0.0, 20.0])).T)   # SoA [3][n]
    ts = rng.uniform(-0.2, 0.2, n)
    te = ts + rng.uniform(-0.3, 0.6, n)
    te[17] = ts[17]
    opts = [nn.newODEoptions(dt=10 ** rng.uniform(-2.5, -1.5), absTol=10 ** rng.uniform(-9, -4), relTol=10 ** rng.uniform(-9, -4),
                             dtMax=10 ** rng.uniform(-2, -0.5), dtMin=10 ** rng.uniform(-6, -4), tStart=ts[i]) for i in range(n)]
    f = nn.Rhs.lorenz()
    y, cnt = nn.solveODECalls(f, y0, te, opts, integrator=integrator)
    col = lambda name: torch.tensor([getattr(o, name) for o in opts], dtype=torch.float64, device=dev)
    yd, cd = nn.solveODEPerIvpEnd(f, torch.from_numpy(y0).to(dev), torch.from_numpy(te).to(dev), nn.newODEoptions(), integrator=integrator,
                                  t_start=col("tStart"), absTol=col("absTol"), relTol=col("relTol"), dtMax=col("dtMax"), dtMin=col("dtMin"), dt=col("dt"))
    assert _same_bits(y, yd.cpu().numpy())
    for k in ("ny", "steps", "rejected"):
        assert (cnt[k] == cd[k].cpu().numpy()).all(), k
    assert cnt["ny"][17] == 1 and np.isnan(y[1, :, 17]).all()
    for i in (0, 17, 101, 299):
        t1, y1 = nn.solveODE(f, torch.from_numpy(np.ascontiguousarray(y0[:, i:i + 1])).to(dev), [ts[i], te[i]], opts[i], integrator=integrator)
        assert _same_bits(y1.cpu().numpy()[:, :, 0][:cnt["ny"][i]], y[:cnt["ny"][i], :, i]), i
    # one object for every call; and the argument checks
    y2, c2 = nn.solveODECalls(f, y0, te, opts[3], integrator=integrator)
    t1, y1 = nn.solveODE(f, torch.from_numpy(np.ascontiguousarray(y0[:, 8:9])).to(dev), [opts[3].tStart, te[8]], opts[3], integrator=integrator)
    assert _same_bits(y1.cpu().numpy()[:, :, 0], y2[:, :, 8])
    with pytest.raises(ValueError):
        nn.solveODECalls(f, y0, te[:5], opts, integrator=integrator)
    with pytest.raises(ValueError):
        nn.solveODECalls(f, y0, te, opts[:5], integrator=integrator)
    ye, ce = nn.solveODECalls(f, np.zeros((3, 0)), np.zeros(0), [], integrator=integrator)
    assert ye.shape == (2, 3, 0)


@pytest.mark.parametrize("integrator", ["tsit54", "dopri54", "bs32", "vern65"])
def test_adaptive_dense_stream_every_rhs_kind(nn, dev, integrator):
    """nnhip_ode_adaptive_stream_dense_f64_dev beyond the compiled-in thread-per-IVP kinds: lanes-per-system systems (Vector[float]
    states of 8 / 16 / 32 components, C4's shape), run-time compiled right-hand sides (whole-vector and per-component bodies), and
    a built-in kind at a size that is instantiated at run time — both directions, dense rows, bitwise equal to the fused solve."""
    import torch
    rng = np.random.default_rng(99)
    heat = "const int l = (c + dim - 1) % dim, r = (c + 1) % dim; return p[0] * ((y[l] - 2.0 * y[c]) + y[r]);"
    kinds = [(nn.Rhs.ring(0.1), 16, 1), (nn.Rhs.ring(0.1), 16, 0), (nn.Rhs.ring(0.2), 8, 1), (nn.Rhs.ring(0.05), 32, 1), (nn.Rhs.affine_t(-0.3, 0.2), 16, 1),
             (nn.Rhs.ring(0.1), 12, 1),                                                                      # built-in kind, run-time size
             (nn.Rhs.custom(2, "dy[0] = y[1]; dy[1] = p[0] * (1.0 - y[0] * y[0]) * y[1] - y[0];", keys=("mu",), defaults={"mu": 1.5}, name="vdp_src"), 2, 0),
             (nn.Rhs.custom(24, heat, keys=("kappa",), defaults={"kappa": 0.4}, name="heat24_dense", per_component=True), 24, 1)]
    grids = [[0.6, -0.3, 0.0, 0.2, 0.45], [0.5, 1.0], [-0.4, -0.2], [0.3, 0.30001, 0.30002, 0.9], [-0.2, -0.20001, -0.1, 0.7, 0.70001]]
    if integrator == "vern65":
        kinds, grids = kinds[:3] + kinds[6:], grids[:2]   # keeps the run-time compilations of the slowest method few
    for f, dim, layout in kinds:
        n = 131
        y0 = rng.uniform(0.5, 1.5, (n, dim))
        y0l = torch.from_numpy(np.ascontiguousarray(y0 if layout == 1 else y0.T)).to(dev)
        for ts in grids:
            for tstart, kw in ((0.0, {}), (0.25, dict(absTol=1e-7, relTol=1e-7, dtMin=1e-6, dtMax=0.5))):
                o2 = nn.newODEoptions(tStart=tstart, **kw)
                t, y, ny, launches = nn.adaptiveStreamSolve(f, y0l, ts, o2, integrator=integrator, layout=layout, check_every=4)
                tf, yf, cf = nn.solveODE(f, y0l, ts, o2, integrator=integrator, layout=layout, return_counts=True)
                assert np.array_equal(t, tf), (dim, ts)
                assert torch.equal(ny, cf["ny"]), (dim, layout, ts, tstart)
                assert torch.equal(torch.nan_to_num(y, nan=-7.0), torch.nan_to_num(yf, nan=-7.0)), (f.kind, dim, layout, ts, tstart)


def test_adaptive_stream_state_that_is_not_16_byte_aligned(nn, dev):
    """The lanes-per-system advance kernel moves AoS states with 16-byte accesses when the arrays allow it; a state tensor that starts
    8 bytes off (a slice of a larger buffer) must take the 8-byte path and give the same bits."""
    import torch
    n, dim = 500, 16
    y0 = torch.from_numpy(_ring_y0(n, dim)).to(dev)
    kw = dict(absTol=1e-8, relTol=1e-8, dtMin=1e-7, dtMax=0.25)
    t, yf = nn.solveODE(nn.Rhs.ring(0.1), y0, [0.0, 1.0], nn.newODEoptions(**kw), integrator="tsit54", layout=1)
    buf = torch.zeros(n * dim + 1, dtype=torch.float64, device=dev)
    view = buf[1:].view(n, dim)
    view.copy_(y0)
    assert view.data_ptr() % 16 == 8 and view.is_contiguous()
    ys, launches = nn.adaptiveStream(nn.Rhs.ring(0.1), view, 0.0, 1.0, nn.newODEoptions(**kw), integrator="tsit54", layout=1, check_every=4)
    torch.cuda.synchronize()
    assert torch.equal(ys, yf[-1])
    ya, _ = nn.adaptiveStream(nn.Rhs.ring(0.1), y0.clone(), 0.0, 1.0, nn.newODEoptions(**kw), integrator="tsit54", layout=1, check_every=4)
    assert torch.equal(ya, yf[-1])


@pytest.mark.parametrize("integrator", ["tsit54", "rk4", "bs32"])
def test_per_call_solve_with_runtime_compiled_right_hand_sides(nn, dev, integrator):
    """nnhip_ode_solve_batch_calls_f64_dev for right-hand sides that are compiled at run time (user source; built-in kinds at sizes
    without an ahead-of-time kernel; thread-per-IVP and lanes-per-system): every IVP's own tspan end, bits of the 2-point fused solves.
    (Found by tests/tools/soak_paths.py: the run-time compiled solve kernel used to ignore the per-call data.)"""
    import torch
    rng = np.random.default_rng(3)
    heat = "const int l = (c + dim - 1) % dim, r = (c + 1) % dim; return p[0] * ((y[l] - 2.0 * y[c]) + y[r]);"
    kinds = [(nn.Rhs.linear(-0.4), 5, 0), (nn.Rhs.ring(0.1), 20, 1), (nn.Rhs.ring(0.1), 40, 0),
             (nn.Rhs.custom(2, "dy[0] = y[1]; dy[1] = p[0] * (1.0 - y[0] * y[0]) * y[1] - y[0];", keys=("mu",), defaults={"mu": 1.5}, name="vdp_calls"), 2, 0),
             (nn.Rhs.custom(24, heat, keys=("kappa",), defaults={"kappa": 0.4}, name="heat24_calls", per_component=True), 24, 1)]
    o = nn.newODEoptions(dt=2.0 ** -7, absTol=1e-7, relTol=1e-7, dtMin=1e-6, dtMax=0.05, tStart=0.25)
    for f, dim, layout in kinds:
        n = 97
        y0 = rng.uniform(0.5, 1.5, (n, dim))
        yt = torch.from_numpy(np.ascontiguousarray(y0 if layout == 1 else y0.T)).to(dev)
        ends = np.array([0.25, 0.6, -0.1, 0.31])
        te = ends[rng.integers(0, 4, n)]
        yc, cc = nn.solveODEPerIvpEnd(f, yt, torch.from_numpy(te).to(dev), o, integrator=integrator, layout=layout)
        for e in ends:
            tt, ye, ce = nn.solveODE(f, yt, [0.25, float(e)], o, integrator=integrator, layout=layout, return_counts=True)
            m = torch.from_numpy(te == e).to(dev)
            a, b = (yc[:, :, m], ye[:, :, m]) if layout == 0 else (yc[:, m, :], ye[:, m, :])
            assert torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)), (dim, layout, e)
            assert torch.equal(cc["ny"][m], ce["ny"][m]) and torch.equal(cc["steps"][m], ce["steps"][m])


@pytest.mark.parametrize("integrator", ["dopri54", "tsit54", "rk4", "vern65", "heun3"])
def test_every_ivp_its_own_dense_tspan(nn, oracle, dev, integrator):
    """nnhip_ode_solve_batch_tspans_f64_dev: IVP i is solveODE(f, y0_i, tspans[i], options) with its OWN n_t-point tspan — unsorted,
    on both sides of tStart, with duplicates, tStart inside or not, requested times closer than the steps (the reference then returns
    fewer rows) — thread-per-IVP, lanes-per-system and run-time compiled kernels: times, rows, row counts and step counters of one
    oracle call per IVP, bit for bit; a non-finite tspan fails its own call only."""
    import torch
    O = oracle
    rng = np.random.default_rng(17)
    kw = dict(dt=2.0 ** -7, absTol=1e-7, relTol=1e-7, dtMin=1e-6, dtMax=0.05, tStart=0.25)
    o, oo = nn.newODEoptions(**kw), O.new_options(**kw)
    for f, okind, params, dim, layout in ((nn.Rhs.lorenz(), O.RHS_LORENZ, LOR, 3, 0), (nn.Rhs.ring(0.1), O.RHS_RING, [0.1], 16, 1),
                                          (nn.Rhs.linear(-0.4), O.RHS_LINEAR, [-0.4], 5, 0), (nn.Rhs.linear(-0.4), O.RHS_LINEAR, [-0.4], 1, 0)):
        for n_t in (1, 2, 6):
            n = 150
            y0 = rng.uniform(0.5, 1.5, (n, dim)) + (np.array([0.0, 0.0, 20.0]) if dim == 3 else 0.0)
            ts = np.round(rng.uniform(0.25 - 0.4, 0.25 + 0.5, (n, n_t)), 3)
            ts[::7, 0] = 0.25                                   # tStart inside
            if n_t > 2:
                ts[::5, 1] = ts[::5, 2]                         # duplicates
                ts[3::11] = np.abs(ts[3::11] - 0.25) + 0.25     # forward side only
                ts[4::13, 1] = ts[4::13, 0] + 1e-5              # closer than a step: the reference drops rows
            ts[9, 0] = np.nan                                   # a call the reference would never return from
            y0l = np.ascontiguousarray(y0 if layout == 1 else y0.T) if dim > 1 else y0[:, 0].copy()
            lay = layout if dim > 1 else 0
            t, y, cnt = nn.solveODEPerIvpTspan(f, torch.from_numpy(y0l).to(dev), torch.from_numpy(ts).to(dev), o, integrator=integrator, layout=lay)
            t, y = t.cpu().numpy(), y.cpu().numpy()
            ny, steps, rej = (cnt[k].cpu().numpy() for k in ("ny", "steps", "rejected"))
            assert ny[9] == -1 and np.isnan(t[9]).all()
            for i in list(range(0, n, 4)) + [9]:
                gi = y[:, i] if dim == 1 else (y[:, :, i] if lay == 0 else y[:, i, :])
                if i == 9:
                    assert np.isnan(gi).all()
                    continue
                yi = list(y0[i]) if dim > 1 else float(y0[i, 0])
                rt, ry, st = O.solve_ode(okind, params, yi, list(ts[i]), oo, integrator)
                assert len(rt) <= n_t and _same_bits(t[i, :len(rt)], np.asarray(rt)) and np.isnan(t[i, len(rt):]).all(), (dim, n_t, i)
                assert ny[i] == st.n_y and steps[i] == st.steps and rej[i] == st.rejected, (dim, n_t, i, integrator)
                assert _same_bits(gi[:st.n_y].reshape(st.n_y, -1), np.asarray(ry).reshape(st.n_y, -1)), (integrator, dim, n_t, i)
                assert np.isnan(gi[st.n_y:]).all()
    # all rows equal -> the plain batched solve, bit for bit; per-IVP options ride along
    n = 64
    y0 = torch.from_numpy(np.ascontiguousarray((rng.uniform(0.5, 1.5, (n, 3)) + np.array([0.0, 0.0, 20.0])).T)).to(dev)
    row = np.array([0.7, -0.1, 0.25, 0.4, 0.55])
    tp, yp, cp = nn.solveODE(nn.Rhs.lorenz(), y0, row, o, integrator=integrator, return_counts=True)
    tq, yq, cq = nn.solveODEPerIvpTspan(nn.Rhs.lorenz(), y0, torch.from_numpy(np.tile(row, (n, 1))).to(dev), o, integrator=integrator)
    assert np.array_equal(tq.cpu().numpy(), np.tile(tp, (n, 1))) and torch.equal(torch.nan_to_num(yq, nan=-7.0), torch.nan_to_num(yp, nan=-7.0))
    assert torch.equal(cq["ny"], cp["ny"]) and torch.equal(cq["steps"], cp["steps"])
    ye = nn.solveODEPerIvpTspan(nn.Rhs.lorenz(), y0, torch.empty((n, 0), dtype=torch.float64, device=dev), o, integrator=integrator)
    assert ye[1].shape[0] == 0 and int(ye[2]["ny"].abs().sum()) == 0


def test_host_form_of_the_per_ivp_tspan_solve(nn, dev):
    """nnhip_ode_solve_batch_tspans_f64 (host arrays, option OBJECTS per call) == the device entry with the same rows and option
    columns, and every call == the ordinary 1-IVP solveODE with its tspan and options."""
    import torch
    rng = np.random.default_rng(23)
    n, n_t = 120, 5
    y0 = np.ascontiguousarray((rng.uniform(0.5, 1.5, (n, 3)) + np.array([0.0, 0.0, 20.0])).T)
    tstart = rng.uniform(-0.2, 0.2, n)
    ts = np.round(tstart[:, None] + rng.uniform(-0.3, 0.5, (n, n_t)), 3)
    ts[::6, 2] = tstart[::6]
    opts = [nn.newODEoptions(dt=10 ** rng.uniform(-2.5, -1.5), absTol=10 ** rng.uniform(-9, -5), relTol=10 ** rng.uniform(-9, -5),
                             dtMax=10 ** rng.uniform(-2, -0.5), dtMin=10 ** rng.uniform(-6, -4), tStart=tstart[i]) for i in range(n)]
    f = nn.Rhs.lorenz()
    for integrator in ("tsit54", "ralston4"):
        th, yh, ch = nn.solveODECallsTspan(f, y0, ts, opts, integrator=integrator)
        col = lambda name: torch.tensor([getattr(o, name) for o in opts], dtype=torch.float64, device=dev)
        td, yd, cd = nn.solveODEPerIvpTspan(f, torch.from_numpy(y0).to(dev), torch.from_numpy(ts).to(dev), nn.newODEoptions(), integrator=integrator,
                                            t_start=col("tStart"), absTol=col("absTol"), relTol=col("relTol"), dtMax=col("dtMax"), dtMin=col("dtMin"), dt=col("dt"))
        assert _same_bits(th, td.cpu().numpy()) and _same_bits(yh, yd.cpu().numpy())
        for k in ("ny", "steps", "rejected"):
            assert (ch[k] == cd[k].cpu().numpy()).all(), k
        for i in (0, 6, 55, 119):
            t1, y1, c1 = nn.solveODE(f, torch.from_numpy(np.ascontiguousarray(y0[:, i:i + 1])).to(dev), ts[i], opts[i], integrator=integrator, return_counts=True)
            k = int(ch["ny"][i])
            assert k == int(c1["ny"][0]) and np.array_equal(th[i, :len(t1)], t1) and np.isnan(th[i, len(t1):]).all()
            assert _same_bits(y1.cpu().numpy()[:k, :, 0], yh[:k, :, i]), (integrator, i)
    with pytest.raises(ValueError):
        nn.solveODECallsTspan(f, y0, ts[:5], opts)


@pytest.mark.parametrize("integrator", ["dopri54", "tsit54", "rk4", "bs32"])
def test_per_call_solves_binned_by_span_are_the_same_calls(nn, dev, integrator):
    """32768 calls or more are integrated longest span first (knob "calls_bin", default on): the same rows, row counts and step counters at the caller's
    indices as in the caller's order — every IVP its own tEnd / tStart / tolerances, both directions, refused calls, non-finite ends, both layouts, a
    16-component system; equal spans (nothing to bin: the order stays the caller's) and a batch below the threshold."""
    import torch
    L = nn._lib.lib()
    rng = np.random.default_rng(17)
    dev_t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def both(fn):
        out = []
        for knob in (1, 0):
            assert L.nnhip_tune_set(b"calls_bin", knob) == 0
            try:
                out.append(fn())
            finally:
                assert L.nnhip_tune_set(b"calls_bin", 1) == 0
        (ya, ca), (yb, cb) = out
        assert torch.equal(torch.nan_to_num(ya, nan=-1.0), torch.nan_to_num(yb, nan=-1.0)) and torch.equal(torch.isnan(ya), torch.isnan(yb))
        assert all(torch.equal(ca[k], cb[k]) for k in ca)
        return ya, ca

    for f, dim, layout, n in ((nn.Rhs.vanderpol(1.5), 2, 0, 70000), (nn.Rhs.lorenz(), 3, 1, 40000), (nn.Rhs.ring(0.1), 16, 1, 33000), (nn.Rhs.lorenz(), 3, 0, 7000)):
        y0 = rng.uniform(0.5, 1.5, (n, dim)) + (np.array([0.0, 0.0, 20.0]) if dim == 3 else 0.0)
        ts = rng.uniform(-0.2, 0.2, n)
        te = ts + rng.uniform(-0.5, 1.5, n) * rng.choice([1.0, 0.05], n)
        te[5], te[6], te[7] = ts[5], np.nan, np.inf                       # an empty span and two ends no call can reach
        tol = 10 ** rng.uniform(-8, -4, n)
        dmin = np.full(n, 1e-7); dmin[11] = 10.0                           # dtMax < dtMin: refused
        y0l = y0 if layout == 1 else y0.T
        kw = dict(integrator=integrator, layout=layout, t_start=dev_t(ts), absTol=dev_t(tol), relTol=dev_t(tol), dtMax=dev_t(np.full(n, 0.3)), dtMin=dev_t(dmin),
                  dt=dev_t(np.full(n, 1e-2)))
        y, cnt = both(lambda: nn.solveODEPerIvpEnd(f, dev_t(y0l), dev_t(te), nn.newODEoptions(), **kw))
        assert int(cnt["steps"].max()) >= 2 * max(int(cnt["steps"].float().median()), 1) or n < 32768   # the calls really differ in length
        both(lambda: nn.solveODEPerIvpEnd(f, dev_t(y0l), dev_t(np.full(n, 0.7)), nn.newODEoptions(absTol=1e-6, relTol=1e-6, dtMax=0.3, dtMin=1e-7, dt=1e-2),
                                          integrator=integrator, layout=layout))                                             # equal spans
    assert L.nnhip_tune_set(b"calls_bin", 2) != 0


@pytest.mark.parametrize("integrator", ["tsit54", "rk4"])
def test_per_call_tspans_binned_by_span_are_the_same_calls(nn, dev, integrator):
    """the n_t-point form (nnhip_ode_solve_batch_tspans_f64_dev) from 32768 calls on: longest integration time first — both directions counted, refused rows last"""
    import torch
    L = nn._lib.lib()
    rng = np.random.default_rng(23)
    n, n_t = 40000, 5
    y0 = torch.from_numpy(np.stack([rng.uniform(1.5, 2.5, n), np.zeros(n)])).to(dev)
    scale = rng.choice([0.05, 0.3, 1.0], n)
    tspans = rng.uniform(-0.5, 1.5, (n, n_t)) * scale[:, None]
    tspans[::7, 2] = 0.0                                  # tStart inside some rows
    tspans[9, 1] = np.nan                                  # a refused call
    ts = torch.from_numpy(tspans).to(dev)
    opt = nn.newODEoptions(absTol=1e-6, relTol=1e-6, dtMax=0.3, dtMin=1e-8, dt=1e-2)
    out = []
    for knob in (1, 0):
        assert L.nnhip_tune_set(b"calls_bin", knob) == 0
        try:
            out.append(nn.solveODEPerIvpTspan(nn.Rhs.vanderpol(1.5), y0, ts, opt, integrator=integrator))
        finally:
            assert L.nnhip_tune_set(b"calls_bin", 1) == 0
    (ta, ya, ca), (tb, yb, cb) = out
    same = lambda a, b: torch.equal(torch.nan_to_num(a, nan=-1.0), torch.nan_to_num(b, nan=-1.0)) and torch.equal(torch.isnan(a), torch.isnan(b))
    assert same(ta, tb) and same(ya, yb) and all(torch.equal(ca[k], cb[k]) for k in ca)
    assert int(ca["ny"][9]) == -1 and int(ca["ny"].max()) == n_t


def test_automatic_polling_schedule_wastes_at_most_two_launches_beyond_the_speculative_pair(nn, dev):
    """check_every <= 0 under knob "adv_auto_poll" = 1 (opt-in: never run on an MI355X before this test does): no step is longer than dtMax, so the
    first ceil((tEnd - t0) / dtMax) launches go out unpolled, then the host polls every 2 launches (2, 2, 4, 8 ...).  BASELINE's C3 / C4 options
    (defaults, tspan [0, 1], dtMax 1e-2): 102 loop iterations, 104 launches (the default, uniform groups of 8: 112) — and the bits of the fused solve,
    lanes-per-system form included."""
    import torch
    with nn.tuning(adv_auto_poll=1):
        _automatic_polling_schedule(nn, dev, torch)


def _automatic_polling_schedule(nn, dev, torch):
    n = _sz(5000, 130)
    for f, y0, layout, integ in ((nn.Rhs.lorenz(), _lorenz_y0(n), 0, "dopri54"), (nn.Rhs.ring(0.1), _ring_y0(n, 16), 1, "tsit54")):
        yt = torch.from_numpy(y0).to(dev)
        t, yf, cnt = nn.solveODE(f, yt, [0.0, 1.0], integrator=integ, layout=layout, return_counts=True)
        ys, launches = nn.adaptiveStream(f, yt.clone(), 0.0, 1.0, integrator=integ, layout=layout)
        need = int(cnt["steps"].max())
        assert torch.equal(ys, yf[-1]) and need == 102 and launches == 104, (integ, need, launches)
        with nn.tuning(adv_auto_poll=0):   # the default: groups of 8, the last one speculative
            ys, launches = nn.adaptiveStream(f, yt.clone(), 0.0, 1.0, integrator=integ, layout=layout)
        assert torch.equal(ys, yf[-1]) and launches == 112, (integ, launches)
    # a span that is not a multiple of dtMax, a batch that finishes at different launches, a bound on the launches
    opt = nn.newODEoptions(absTol=1e-7, relTol=1e-7, dtMin=1e-9, dtMax=0.03)
    yh = _lorenz_y0(_sz(4000, 130))
    yh[0] *= np.linspace(1.0, 30.0, _sz(4000, 130))
    yt = torch.from_numpy(yh).to(dev)
    t, yf, cnt = nn.solveODE(nn.Rhs.lorenz(), yt, [0.0, 0.5], opt, integrator="tsit54", return_counts=True)
    ys, launches = nn.adaptiveStream(nn.Rhs.lorenz(), yt.clone(), 0.0, 0.5, opt, integrator="tsit54")
    need = int(cnt["steps"].max())
    assert torch.equal(ys, yf[-1]) and need <= launches <= need + 16, (need, launches)
    assert need > 17  # = ceil(0.5 / 0.03): the unpolled part was really shorter than the loop
    # the dense driver polls on the same schedule, per direction: 11 requested times on both sides of tStart, rows and launches
    ts = np.concatenate([np.linspace(-0.3, -0.05, 4), np.linspace(0.0, 1.0, 7)])
    yl = torch.from_numpy(_lorenz_y0(_sz(3000, 70))).to(dev)
    tf, yf, cf = nn.solveODE(nn.Rhs.lorenz(), yl, ts, integrator="dopri54", return_counts=True)
    t2, y2, ny2, launches = nn.adaptiveStreamSolve(nn.Rhs.lorenz(), yl, ts, integrator="dopri54")
    assert np.array_equal(t2, tf) and np.array_equal(y2.cpu().numpy(), yf.cpu().numpy(), equal_nan=True)
    assert launches <= 102 + 4 + 32 + 4      # forward: 100 unpolled + 2 + 2; backward: ceil(0.3 / 0.01) = 30 unpolled + the two short first steps + 2 + 2 (uniform groups of 8: 112 + 40)


def test_lean_advance_kernels_give_the_general_kernels_bits(nn, oracle, dev):
    """advance_lps_lean_kernel / advance_tpi_lean_kernel (the streaming driver's own layout as the kernel's contract: one uniform base per
    block + a 32-bit lane offset, seven scalar arguments) against the general kernels (knob "adv_lean" = 0) and the oracle: DOPRI54 and Tsit54,
    thread-per-IVP (SoA planes) and lanes-per-system (AoS) systems, batches that do not fill their last block, rejections, tight and loose
    tolerances, heterogeneous finishing times."""
    import torch
    O = oracle
    L = nn._lib.lib()
    cases = []
    for n in _sz((1, 63, 64, 257, 3001), (1, 63, 65)):
        cases.append(("lorenz", nn.Rhs.lorenz(), _lorenz_y0(n), 0, 3, (O.RHS_LORENZ, LOR)))
        cases.append(("ring16", nn.Rhs.ring(0.1), _ring_y0(n, 16), 1, 16, (O.RHS_RING, [0.1])))
    cases.append(("ring8", nn.Rhs.ring(0.1), _ring_y0(_sz(500, 40), 8), 1, 8, (O.RHS_RING, [0.1])))
    cases.append(("ring32", nn.Rhs.ring(0.1), _ring_y0(_sz(130, 10), 32), 1, 32, (O.RHS_RING, [0.1])))
    nv = _sz(999, 70)
    cases.append(("vdp", nn.Rhs.vanderpol(3.0), np.stack([np.linspace(0.5, 2.5, nv), np.zeros(nv)]), 0, 2, None))
    for name, f, y0, layout, dim, orc in cases:
        for integ in ("dopri54", "tsit54"):
            for kw in ({}, dict(absTol=1e-9, relTol=1e-9, dtMin=1e-8, dtMax=0.2)):
                y0 = np.ascontiguousarray(y0)
                yt = torch.from_numpy(y0).to(dev)
                opt = nn.newODEoptions(**kw)
                got = {}
                for lean in (1, 0):
                    try:
                        assert L.nnhip_tune_set(b"adv_lean", lean) == 0
                        got[lean], launches = nn.adaptiveStream(f, yt.clone(), 0.0, 0.7, opt, integrator=integ, layout=layout)
                    finally:
                        assert L.nnhip_tune_set(b"adv_lean", 0) == 0
                assert torch.equal(got[1], got[0]), (name, integ, kw)
                t, yf = nn.solveODE(f, yt, [0.0, 0.7], opt, integrator=integ, layout=layout)
                assert torch.equal(got[1], yf[-1]), (name, integ, kw)
                if orc is not None and y0.shape[1 - layout] <= 300:
                    n = y0.shape[1 - layout]
                    ref = O.solve_ode_batch(orc[0], orc[1], y0, n, dim, [0.0, 0.7], O.new_options(**kw), integ, layout=layout, n_threads=4)
                    assert _same_bits(got[1].cpu().numpy(), ref["y"][-1]), (name, integ, kw)


def test_streamed_fp_contract_opt_in_stays_within_north_star_tolerance(nn, oracle, dev):
    """Knob "fp_contract" = 1 on the adaptive streaming loop (round 6): where a launch has the lean kernels' layout — the driver's default set-up — the
    FMA-contracted lean kernel of DOPRI54 / Tsit54 advances it (ode_tu_lean_fast.hip): BASELINE's C3 / C4 shapes, default and tight options, against the
    oracle at north_star's tolerance for adaptive methods (1e-6 absolute per component).  Step counts can differ from the bit-exact loop on knife-edge
    steps: the launch counts are reported in the assertion message, not asserted equal.  A launch WITHOUT that layout (FSAL carried: knob
    adv_recompute_fsal = 0) keeps the bit-exact general kernel under the same knob, and knob off is bit-exact again."""
    import torch
    O = oracle
    n = _sz(3001, 70)
    cases = (("lorenz", nn.Rhs.lorenz(), _lorenz_y0(n), 0, 3, (O.RHS_LORENZ, LOR), "dopri54"), ("ring16", nn.Rhs.ring(0.1), _ring_y0(n, 16), 1, 16, (O.RHS_RING, [0.1]), "tsit54"),
             ("lorenz", nn.Rhs.lorenz(), _lorenz_y0(n), 0, 3, (O.RHS_LORENZ, LOR), "tsit54"), ("ring16", nn.Rhs.ring(0.1), _ring_y0(n, 16), 1, 16, (O.RHS_RING, [0.1]), "dopri54"))
    for name, f, y0, layout, dim, orc, integ in cases:
        for kw in ({}, dict(absTol=1e-10, relTol=1e-10, dtMin=1e-6, dtMax=1e-1)):
            y0 = np.ascontiguousarray(y0)
            yt = torch.from_numpy(y0).to(dev)
            opt = nn.newODEoptions(**kw)
            exact, l_exact = nn.adaptiveStream(f, yt.clone(), 0.0, 1.0, opt, integrator=integ, layout=layout)
            with nn.tuning(fp_contract=1):
                fast, l_fast = nn.adaptiveStream(f, yt.clone(), 0.0, 1.0, opt, integrator=integ, layout=layout)
                with nn.tuning(adv_recompute_fsal=0):   # FSAL carried through HBM: not the lean layout -> the general kernel, the reference's bits
                    carried, _ = nn.adaptiveStream(f, yt.clone(), 0.0, 1.0, opt, integrator=integ, layout=layout)
            again, _ = nn.adaptiveStream(f, yt.clone(), 0.0, 1.0, opt, integrator=integ, layout=layout)
            assert torch.equal(carried, exact) and torch.equal(again, exact), (name, integ, kw)
            ref = O.solve_ode_batch(orc[0], orc[1], y0, n, dim, [0.0, 1.0], O.new_options(**kw), integ, layout=layout, n_threads=8)
            dev_abs = float(np.abs(fast.cpu().numpy() - ref["y"][-1]).max())
            assert _same_bits(exact.cpu().numpy(), ref["y"][-1]), (name, integ, kw)
            assert 0.0 < dev_abs <= TOL_ADAPTIVE, (name, integ, kw, dev_abs, "launches exact / contracted", l_exact, l_fast)


def test_bin_order_spends_its_bins_on_the_keys_it_gets(nn, dev):
    """Round-4 advice: the order of integration bins the keys by their bit image — logarithmic all the way down — so ONE key equal to 0 (a finished
    IVP's "0 steps left", a zero-length span) or keys of both signs used to leave a uniform sweep with four bins for half its IVPs: the speed-up gone,
    the results unchanged, no test the wiser.  The bins are linear in value when the range touches or straddles zero; same-signed keys keep the
    logarithmic image.  Checked on the order itself (nnhip_ode_bin_order_f64_dev): sorted from slice to slice, every index once, and wavefronts
    (64 consecutive positions) whose keys lie close together."""
    import torch
    L = nn._lib.lib()
    n = 1 << 18
    rng = np.random.default_rng(3)

    def order_of(keys):
        kt = torch.from_numpy(keys).to(dev)
        out = torch.empty(n, dtype=torch.int32, device=dev)
        assert L.nnhip_ode_bin_order_f64_dev(kt.data_ptr(), n, out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0, nn._lib.last_error()
        torch.cuda.synchronize()
        o = out.cpu().numpy().astype(np.int64) & 0xffffffff
        assert np.array_equal(np.sort(o), np.arange(n))        # a permutation
        return o

    def wave_spread(keys, o):
        """median over wavefronts of (max - min of the keys it holds) / (range of all finite keys)"""
        k = keys[o]
        fin = np.isfinite(keys)
        span = keys[fin].max() - keys[fin].min()
        w = k[: (n // 64) * 64].reshape(-1, 64)
        ok = np.isfinite(w).all(axis=1)
        return float(np.median((w[ok].max(axis=1) - w[ok].min(axis=1)) / span))

    cases = {
        "uniform_with_a_zero": np.concatenate([[0.0], rng.uniform(0.0, 10.0, n - 1)]),          # linspace(0, ...)-like sweep
        "steps_left_with_finished_ivps": -np.concatenate([np.zeros(1000), rng.uniform(1.0, 500.0, n - 1000)]),
        "both_signs": rng.uniform(-5.0, 10.0, n),                                                  # a centred parameter
        "one_sign_narrow": rng.uniform(100.0, 101.0, n),
        "one_sign_six_decades": -10.0 ** rng.uniform(-6.0, 0.0, n),                               # probe progress
    }
    for name, keys in cases.items():
        rng.shuffle(keys)
        o = order_of(keys)
        k = keys[o]
        # ascending up to the width of a slice: a key may precede a smaller one only inside its own slice (1 / 4094 of the range, or of the image's)
        spread = wave_spread(keys, o)
        if name == "one_sign_six_decades":
            w = np.log10(-k[: (n // 64) * 64]).reshape(-1, 64)
            assert float(np.median(w.max(axis=1) - w.min(axis=1))) < 0.02, name      # a wavefront spans < 5 % in value: logarithmic slices
        else:
            wide = 2.0 if name == "one_sign_narrow" else 1.0   # (the logarithmic image is cut by a power-of-two shift: slices up to twice the ideal width)
            assert spread < wide * 3.0 / 4094, (name, spread)                           # ~64 of 2^18 uniform keys per slice: a wavefront = one or two slices
            viol = np.maximum.accumulate(k) - k
            assert float(viol.max()) <= wide * 1.01 * (keys.max() - keys.min()) / 4094, name
    # non-finite keys go last, whatever the others are
    keys = rng.uniform(-1.0, 1.0, n)
    keys[::97] = np.nan
    keys[5::1013] = np.inf
    o = order_of(keys)
    tail = ~np.isfinite(keys[o])
    assert tail.sum() == (~np.isfinite(keys)).sum() and tail[-tail.sum():].all()


@pytest.mark.parametrize("family", ["tpi_scalar", "tpi_lorenz", "lps_ring16"])
def test_polled_launches_see_a_lone_straggler_in_every_row_of_a_wave(nn, dev, family):
    """The polling loop's "anyone still integrating?" is a workgroup OR over `stillActive` (__syncthreads_or: row scans, one wave-wide shift, a row mirror, LDS across waves).
    A batch in which ONE member needs at least twice the steps of all the others leaves, for most of the loop, a single active lane — here placed in every row of its wave
    (lanes 0-15 / 16-31 / 32-47 / 48-63), in the first and in a later workgroup.  If the reduction lost a row the loop would end while the straggler is still integrating: every
    driver (general and lean advance kernels, the dense driver; polling after every launch and in groups of 3) must return the fused solve's bits.  (Written after the ISA
    interpreter of the CPU suite turned out to drop rows 1 and 3 — tools/gfx950_isa_interp.py, wave_shl:1 — which this repository's homogeneous BASELINE batches never exercised.)"""
    import torch
    opt = nn.newODEoptions(absTol=1e-9, relTol=1e-9, dtMin=1e-9, dtMax=0.5)
    if family == "tpi_scalar":
        f, layout, n = nn.Rhs.linear(-3.0), 0, 150
        base = np.full(n, 1e-13)                      # far below absTol / relTol: the controller lets these run at dtMax
        big = lambda a, k: a.__setitem__(k, 1.0)     # noqa: E731
        positions = (0, 17, 38, 63, 64 + 20, 128 + 21)
    elif family == "tpi_lorenz":
        f, layout, n = nn.Rhs.lorenz(), 0, 90
        base = np.full((3, n), 1e-13)
        big = lambda a, k: a.__setitem__((slice(None), k), [1.0, 1.0, 25.0])   # noqa: E731
        positions = (5, 16, 47, 60, 64 + 18)
    else:   # 4 lanes per system: a 256-lane workgroup holds 64 systems, a wave 16 of them; system k sits in row (k % 16) // 4 of wave (k % 64) // 16
        f, layout, n = nn.Rhs.ring(0.9), 1, 80
        base = np.full((n, 16), 1e-13)
        big = lambda a, k: a.__setitem__(k, 1.0 + np.arange(16) / 4.0)         # noqa: E731
        positions = (1, 6, 10, 15, 16 + 5, 48 + 14, 64 + 7)
    for k in positions:
        y0h = base.copy()
        big(y0h, k)
        y0 = torch.from_numpy(y0h).to(dev)
        tf, yf, cnt = nn.solveODE(f, y0, [0.0, 2.0], opt, integrator="tsit54", layout=layout, return_counts=True)
        steps = cnt["steps"].cpu().numpy()
        assert steps[k] >= 2 * np.delete(steps, k).max(), (family, k, int(steps[k]), int(np.delete(steps, k).max()))   # the batch has the shape this test is about: the straggler integrates alone for half the loop or more
        for ce in (1, 3):
            ys, launches = nn.adaptiveStream(f, y0.clone(), 0.0, 2.0, opt, integrator="tsit54", layout=layout, check_every=ce)
            assert torch.equal(ys, yf[-1]) and launches >= int(steps[k]), (family, k, ce, "general", launches, int(steps[k]))
            with nn.tuning(adv_lean=1):
                yl, ll = nn.adaptiveStream(f, y0.clone(), 0.0, 2.0, opt, integrator="tsit54", layout=layout, check_every=ce)
            assert torch.equal(yl, yf[-1]) and ll >= int(steps[k]), (family, k, ce, "lean", ll, int(steps[k]))
            ts = [0.0, 0.7, 2.0]
            t2, yd, ny, ld = nn.adaptiveStreamSolve(f, y0, ts, opt, integrator="tsit54", layout=layout, check_every=ce)
            assert torch.equal(yd, nn.solveODE(f, y0, ts, opt, integrator="tsit54", layout=layout)[1]), (family, k, ce, "dense")
