"""GPU parity of the RK4 hot path (ode.nim:180-189 + ODESolver loop :511-542) against the CPU oracle.

Tolerance stated by BASELINE.json's north_star: 1e-10 abs for fixed-step RK4.  Because the device code is
built -ffp-contract=off with the reference's operation order we additionally assert BIT equality.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_FIXED = 1e-10


def _c1(n=1024):
    return 1.0 + np.arange(n, dtype=np.float64) * 2.0 ** -10


def test_c1_fused_solve_matches_oracle(nn, oracle, dev):
    """BASELINE config C1: dy/dt=-y, 1024 scalar IVPs, dt=2^-10, 1000 steps."""
    import torch
    O = oracle
    y0 = _c1()
    dt = 2.0 ** -10
    tspan = [0.0, 1000 * dt]
    t, y, cnt = nn.solveODE(nn.Rhs.neg_y(), torch.from_numpy(y0).to(dev), tspan, nn.newODEoptions(dt=dt), integrator="rk4",
                            return_counts=True)
    ref = O.solve_ode_batch(O.RHS_NEG_Y, [], y0, len(y0), 0, tspan, O.new_options(dt=dt), "rk4")
    got = y.cpu().numpy()
    assert np.array_equal(t, ref["t"])
    assert np.abs(got[-1] - ref["y"][-1, 0]).max() <= TOL_FIXED
    assert np.array_equal(got, ref["y"][:, 0, :]), "expected bit-exact agreement"
    assert np.array_equal(cnt["steps"].cpu().numpy(), ref["steps"])
    assert (cnt["steps"].cpu().numpy() == 1000).all()
    assert np.array_equal(cnt["ny"].cpu().numpy(), ref["ny"])
    # SURVEY Appendix B known-answer (independent restatement): y0=1.0 -> 0x1.81a455c174b97p-2
    assert float(got[-1][0]).hex() == "0x1.81a455c174b97p-2"


@pytest.mark.parametrize("pingpong", [False, True])
def test_c1_step_streaming_matches_oracle(nn, oracle, dev, pingpong):
    import torch
    O = oracle
    y0 = _c1()
    dt = 2.0 ** -10
    y = torch.from_numpy(y0).to(dev)
    scratch = torch.empty_like(y) if pingpong else None
    yf, nsteps = nn.fixedStream(nn.Rhs.neg_y(), y, 0.0, 1000 * dt, nn.newODEoptions(dt=dt), integrator="rk4", scratch=scratch)
    ref = O.solve_ode_batch(O.RHS_NEG_Y, [], y0, len(y0), 0, [0.0, 1000 * dt], O.new_options(dt=dt), "rk4")
    assert nsteps == 1000
    assert np.array_equal(yf.cpu().numpy(), ref["y"][-1, 0])


@pytest.mark.parametrize("n", [1, 2, 63, 255, 256, 2047, 2048, 2049, 4097, 100_003])
def test_stream_ragged_sizes(nn, oracle, dev, n):
    """Ragged batch sizes exercise the vector kernel's scalar tail tile."""
    import torch
    O = oracle
    rng = np.random.default_rng(n)
    y0 = rng.uniform(-2.0, 2.0, n)
    dt = 1e-3  # non-dyadic: exercises the drifting t += dt loop count as well
    opt, oo = nn.newODEoptions(dt=dt), O.new_options(dt=dt)
    yf, nsteps = nn.fixedStream(nn.Rhs.linear(-0.7), torch.from_numpy(y0).to(dev), 0.0, 0.05, opt, integrator="rk4")
    ref = O.solve_ode_batch(O.RHS_LINEAR, [-0.7], y0, n, 0, [0.0, 0.05], oo, "rk4")
    assert nsteps == ref["steps"][0]
    assert np.array_equal(yf.cpu().numpy(), ref["y"][-1, 0])


def test_empty_batch(nn, dev):
    import torch
    y0 = torch.empty(0, dtype=torch.float64, device=dev)
    t, y = nn.solveODE(nn.Rhs.neg_y(), y0, [0.0, 1.0], nn.newODEoptions(dt=0.25), integrator="rk4")
    assert y.shape == (2, 0) and list(t) == [0.0, 1.0]


def test_reference_harness_dense_forward_backward(nn, oracle, dev):
    """tests/test_ode.nim:5-46 harness: f=-0.1*y, tspan=linspace(-10,10,100), tStart=0 not in tspan ->
    backward AND forward branches with dense Hermite output; `t == tspan` and |y-exp(-0.1t)| <= 1e-4."""
    import torch
    O = oracle
    ts = O.linspace(-10.0, 10.0, 100)
    y0 = np.array([1.0, 0.5, 2.0, -1.25])
    opt, oo = nn.newODEoptions(dt=1e-2), O.new_options(dt=1e-2)
    t, y = nn.solveODE(nn.Rhs.linear(-0.1), torch.from_numpy(y0).to(dev), ts, opt, integrator="rk4")
    assert np.array_equal(t, ts)  # check t == tspan
    got = y.cpu().numpy()
    assert np.abs(got[:, 0] - np.exp(-0.1 * ts)).max() <= 1e-4
    ref = O.solve_ode_batch(O.RHS_LINEAR, [-0.1], y0, len(y0), 0, ts, oo, "rk4")
    assert np.abs(got - ref["y"][:, 0, :]).max() <= TOL_FIXED
    assert np.array_equal(got, ref["y"][:, 0, :])


def test_c2_full_size_properties(nn, oracle, dev):
    """BASELINE config C2 at full size (1e7 IVPs, 1000 RK4 steps) through size-independent properties:
    (1) oracle parity on a fixed 4096-index subsample, (2) exact linearity in y0 under power-of-two
    scaling (RK4 on a linear RHS commutes with exact scalings), (3) fused == streamed bitwise,
    (4) monotone in y0 (the RK4 amplification factor is a positive constant)."""
    import torch
    from numericalnim_amd import distributed as nd
    O = oracle
    n, dt, nsteps = 10_000_000, 2.0 ** -10, 1000   # C2 itself: 1e7 IVPs x 1000 steps
    y0 = nd.c2_y0_torch(0, n, dev)
    opt = nn.newODEoptions(dt=dt)
    y = y0.clone()
    yf, k = nn.fixedStream(nn.Rhs.neg_y(), y, 0.0, nsteps * dt, opt, integrator="rk4")
    assert k == nsteps
    idx = (np.arange(4096, dtype=np.int64) * 2441) % n
    ref = O.solve_ode_batch(O.RHS_NEG_Y, [], nd.c2_y0_numpy(0, n)[idx], len(idx), 0, [0.0, nsteps * dt], O.new_options(dt=dt), "rk4")
    assert np.array_equal(yf[torch.from_numpy(idx).to(dev)].cpu().numpy(), ref["y"][-1, 0])
    y2 = (y0 * 4.0).clone()
    y2f, _ = nn.fixedStream(nn.Rhs.neg_y(), y2, 0.0, nsteps * dt, opt, integrator="rk4")
    assert torch.equal(y2f, yf * 4.0)
    _, yfused = nn.solveODE(nn.Rhs.neg_y(), y0, [0.0, nsteps * dt], opt, integrator="rk4")
    assert torch.equal(yfused[-1], yf)
    assert torch.equal(yfused[0], y0)
    order = torch.argsort(y0[:1 << 20])
    assert bool((torch.diff(yf[:1 << 20][order]) >= 0).all())


def test_all_integrators_reference_harness(nn, oracle, dev):
    """Every integrator name of the reference (allODE, ode.nim:40-42) on the tests/test_ode.nim harness, scalar and
    Vector states, against the oracle and against the reference's own analytic tolerance for that integrator."""
    import torch
    O = oracle
    ts = O.linspace(-10.0, 10.0, 100)
    tol_ref = {"dopri54": 1e-4, "rk4": 1e-4, "heun2": 1e-10, "ralston2": 1e-10, "kutta3": 1e-10, "heun3": 1e-10, "ralston3": 1e-10,
               "ssprk3": 1e-10, "ralston4": 1e-10, "kutta4": 1e-10, "rk21": 1e-6, "bs32": 1e-6, "tsit54": 1e-4, "vern65": 1e-4}
    assert sorted(tol_ref) == sorted(nn.allODE)
    y0s = torch.tensor([1.0], dtype=torch.float64, device=dev)
    y0v = torch.ones(3, 1, dtype=torch.float64, device=dev)
    for m in nn.allODE:
        t, y = nn.solveODE(nn.Rhs.linear(-0.1), y0s, ts, integrator=m)           # default options, as test_ode.nim
        assert np.array_equal(t, ts)
        got = y[:, 0].cpu().numpy()
        assert np.abs(got - np.exp(-0.1 * ts)).max() <= tol_ref[m], m
        rt, ry, st = O.solve_ode(O.RHS_LINEAR, [-0.1], 1.0, ts, O.new_options(), m)
        assert np.array_equal(got, ry), m   # all 14 integrators bit-exact, adaptive ones included
        tv, yv = nn.solveODE(nn.Rhs.linear(-0.1), y0v, ts, integrator=m)
        gv = yv[:, :, 0].cpu().numpy()
        assert np.array_equal(gv[:, 0], got) and np.array_equal(gv[:, 1], got) and np.array_equal(gv[:, 2], got), m


def test_stream_loop_hipgraph_replay(nn, oracle, dev):
    """Tuning knob "stream_graph": the fixed-step streaming loop captured in a hipGraph and replayed gives the same bits
    as eager launches (2.5x faster for launch-bound batch sizes, profiles/r01_stream_graph.txt)."""
    import torch
    O = oracle
    L = nn._lib.lib()
    n, dt, nsteps = 5000, 2.0 ** -10, 200
    y0 = 1.0 + np.arange(n) * 2.0 ** -13
    ref = O.solve_ode_batch(O.RHS_LINEAR, [-0.5], y0, n, 0, [0.0, nsteps * dt], O.new_options(dt=dt), "rk4")["y"][-1, 0]
    side = torch.cuda.Stream()
    try:
        L.nnhip_tune_set(b"stream_graph", 1)
        with torch.cuda.stream(side):
            y = torch.from_numpy(y0).to(dev)
            sc = torch.empty_like(y)
            for rep in range(3):  # first call captures + instantiates, later calls replay the cached graph
                y.copy_(torch.from_numpy(y0).to(dev))
                yf, k = nn.fixedStream(nn.Rhs.linear(-0.5), y, 0.0, nsteps * dt, nn.newODEoptions(dt=dt), integrator="rk4", scratch=sc)
                side.synchronize()
                assert k == nsteps and np.array_equal(yf.cpu().numpy(), ref)
    finally:
        L.nnhip_tune_set(b"stream_graph", 2)
    # default (automatic) mode: the first identical call runs eagerly, later ones replay a graph; other methods and a
    # run-time compiled right-hand side included; always the same bits as the eager loop
    L.nnhip_tune_set(b"stream_graph", 0)
    eager = {}
    cases = [("rk4", nn.Rhs.linear(-0.5), 1), ("heun3", nn.Rhs.lorenz(), 3), ("kutta4", nn.Rhs.linear(-0.3), 5)]
    with torch.cuda.stream(side):
        for integ, f, dim in cases:
            yy = torch.from_numpy(np.tile(y0[:1000], (dim, 1)) if dim > 1 else y0[:1000].copy()).to(dev)
            eager[integ] = nn.fixedStream(f, yy, 0.0, 64 * dt, nn.newODEoptions(dt=dt), integrator=integ, scratch=torch.empty_like(yy))[0].clone()
        side.synchronize()
        L.nnhip_tune_set(b"stream_graph", 2)
        for integ, f, dim in cases:
            base = torch.from_numpy(np.tile(y0[:1000], (dim, 1)) if dim > 1 else y0[:1000].copy()).to(dev)
            yy, sc2 = base.clone(), torch.empty_like(base)
            for rep in range(4):
                yy.copy_(base)
                got = nn.fixedStream(f, yy, 0.0, 64 * dt, nn.newODEoptions(dt=dt), integrator=integ, scratch=sc2)[0]
                side.synchronize()
                assert torch.equal(got, eager[integ]), (integ, rep)


def test_host_path_result_array_reuse_and_page_locked_buffers(nn, oracle):
    """solveODE(numpy, out=...) writes into the caller's array; page-locked y0 / out take the chunked, overlapped transfer path
    (automatic) — same bits as the plain call."""
    import torch
    n = 300000
    y0 = 1.0 + (np.arange(n) % 4096) * 2.0 ** -12
    opt = nn.newODEoptions(dt=2.0 ** -7)
    ts = [0.0, 0.5, 1.0]
    t, ref = nn.solveODE(nn.Rhs.neg_y(), y0, ts, opt, integrator="rk4")
    out = np.zeros((3, n))
    t2, y2 = nn.solveODE(nn.Rhs.neg_y(), y0, ts, opt, integrator="rk4", out=out)
    assert y2 is out and np.array_equal(out, ref)
    y0p = torch.empty(n, dtype=torch.float64).pin_memory()
    y0p.numpy()[:] = y0
    outp = torch.zeros(3, n, dtype=torch.float64).pin_memory()
    nn.solveODE(nn.Rhs.neg_y(), y0p.numpy(), ts, opt, integrator="rk4", out=outp.numpy())
    assert np.array_equal(outp.numpy(), ref)
    # large enough for the automatic chunked pipeline (>= 32 MB moved)
    big = 3_000_000
    yb = torch.empty(big, dtype=torch.float64).pin_memory()
    yb.numpy()[:] = 1.0 + (np.arange(big) % 4096) * 2.0 ** -12
    ob = torch.zeros(2, big, dtype=torch.float64).pin_memory()
    nn.solveODE(nn.Rhs.neg_y(), yb.numpy(), [0.0, 1.0], opt, integrator="rk4", out=ob.numpy())
    tb, refb = nn.solveODE(nn.Rhs.neg_y(), yb.numpy().copy(), [0.0, 1.0], opt, integrator="rk4")
    assert np.array_equal(ob.numpy(), refb)
    with pytest.raises(ValueError):
        nn.solveODE(nn.Rhs.neg_y(), y0, ts, opt, integrator="rk4", out=np.zeros((2, n)))


@pytest.mark.parametrize("ipl", [2, -2])
@pytest.mark.parametrize("integrator", ["rk4", "heun2", "ralston2", "kutta3", "heun3", "ralston3", "ssprk3", "ralston4", "kutta4"])
def test_fixed_step_vectorised_streaming_kernel(nn, oracle, dev, integrator, ipl):
    """fixed_stream_vec_kernel (any fixed-step IntegratorProc over a thread-per-IVP system, 16-byte lane accesses, 2 IVPs per
    lane, plain (ipl = 2) and non-temporal (ipl = -2 here: knob adv_nontemporal forced on) instantiations; ode.nim:107-189) vs the
    one-IVP-per-lane step kernel (tuning knob fixed_vec_ipl = 0) and vs the oracle's
    stepper: same bits for both layouts, uniform and per-IVP (t, dt), with and without the FSAL slot, full tiles + ragged
    tail, negated time, and in-place (y_out == y_in)."""
    import torch
    O = oracle
    L = nn._lib.lib()
    rng = np.random.default_rng(17)
    opt = nn.newODEoptions(dt=2.0 ** -6)
    cases = [(nn.Rhs.lorenz(), O.RHS_LORENZ, [10.0, 28.0, 8.0 / 3.0], 3), (nn.Rhs.vanderpol(1.5), O.RHS_VANDERPOL, [1.5], 2),
             (nn.Rhs.linear(-0.4), O.RHS_LINEAR, [-0.4], 1), (nn.Rhs.ring(0.1), O.RHS_RING, [0.1], 4)]
    try:
        for f, okind, params, dim in cases:
            for n in (5000, 4097 if dim == 1 else 4098):   # several full tiles + a ragged tail
                for layout in ((0, 1) if dim > 1 else (0,)):
                    y = rng.uniform(-2, 2, (n, dim)) + (np.array([0.0, 0.0, 20.0]) if dim == 3 else 0.0)
                    yl = np.ascontiguousarray(y if layout == 1 else y.T) if dim > 1 else y[:, 0].copy()
                    yt = torch.from_numpy(yl).to(dev)
                    tarr = torch.from_numpy(rng.uniform(0, 1, n)).to(dev)
                    dtarr = torch.from_numpy(10 ** rng.uniform(-3, -1.5, n)).to(dev)
                    for tt, dd, fsal, neg in ((0.25, 2.0 ** -6, None, False), (tarr, dtarr, yt, False), (0.25, dtarr, None, True), (tarr, 2.0 ** -6, yt, True)):
                        L.nnhip_tune_set(b"fixed_vec_ipl", 0)
                        r0 = nn.integratorStep(f, tt, yt, fsal, dd, opt, integrator=integrator, layout=layout, negate_time=neg)
                        L.nnhip_tune_set(b"fixed_vec_ipl", abs(ipl))
                        L.nnhip_tune_set(b"adv_nontemporal", 1 if ipl < 0 else 0)
                        r1 = nn.integratorStep(f, tt, yt, fsal, dd, opt, integrator=integrator, layout=layout, negate_time=neg)
                        assert torch.equal(r0[0], r1[0]), (integrator, dim, n, layout)
                        if fsal is not None:
                            assert torch.equal(r1[1], r1[0])          # yNew in the FSAL slot (ode.nim:189)
                        inplace = yt.clone()
                        nn.integratorStep(f, tt, inplace, None, dd, opt, integrator=integrator, layout=layout, negate_time=neg, out=inplace)
                        assert torch.equal(inplace, r1[0])
                    # against the oracle's stepper on a sample (uniform t, dt)
                    got = nn.integratorStep(f, 0.25, yt, None, 2.0 ** -6, opt, integrator=integrator, layout=layout)[0].cpu().numpy()
                    got = got.reshape(n, dim) if (layout == 1 or dim == 1) else got.T
                    for i in (0, 1, 511, 512, 1023, 1024, n - 2, n - 1):
                        yi = list(y[i]) if dim > 1 else float(y[i, 0])
                        ryn = O.step(okind, params, integrator, O.new_options(dt=2.0 ** -6), 0.25, yi, yi, 2.0 ** -6)[0]
                        assert np.array_equal(got[i], np.atleast_1d(ryn)), (integrator, dim, i)
    finally:
        L.nnhip_tune_set(b"fixed_vec_ipl", 2)
        L.nnhip_tune_set(b"adv_nontemporal", -1)


@pytest.mark.parametrize("integrator", ["rk4", "heun2", "kutta3", "ssprk3", "ralston4"])
def test_fixed_step_dense_output_through_the_step_streaming_seam(nn, oracle, dev, integrator):
    """nnhip_ode_fixed_stream_dense_f64_dev: the whole ODESolver driver (ode.nim:471-586) over one IntegratorProc launch per time
    step — the reference's own harness (tests/test_ode.nim:15: linspace(-10, 10, 100), both directions, dense Hermite rows) and
    the awkward grids (tStart inside / outside / duplicated, unsorted, single point, rows the reference drops) — bitwise equal to
    the fused solve and to the oracle, for scalar, small-vector, lanes-per-system and run-time compiled right-hand sides."""
    import torch
    O = oracle
    rng = np.random.default_rng(33)
    ts_h = O.linspace(-10.0, 10.0, 100)
    y0s = torch.from_numpy(1.0 + np.arange(50) * 2.0 ** -7).to(dev)
    t, y, ny, ns = nn.fixedStreamSolve(nn.Rhs.linear(-0.1), y0s, ts_h, nn.newODEoptions(dt=2.0 ** -6), integrator=integrator)
    tf, yf, cf = nn.solveODE(nn.Rhs.linear(-0.1), y0s, ts_h, nn.newODEoptions(dt=2.0 ** -6), integrator=integrator, return_counts=True)
    assert np.array_equal(t, ts_h) and np.array_equal(t, tf) and torch.equal(y, yf) and ny == 100
    assert ns == int(cf["steps"][0])
    rt, ry, st = O.solve_ode(O.RHS_LINEAR, [-0.1], float(y0s[3]), ts_h, O.new_options(dt=2.0 ** -6), integrator)
    assert np.array_equal(y[:, 3].cpu().numpy(), np.asarray(ry))
    cases = [(nn.Rhs.lorenz(), 3, 0), (nn.Rhs.lorenz(), 3, 1), (nn.Rhs.ring(0.1), 16, 1), (nn.Rhs.ring(0.1), 6, 0), (nn.Rhs.neg_y(), 1, 0)]
    grids = [[1.0, -1.0, 0.0], [0.0, 0.0, 1.0], [-2.0, -1.0], [2.0], [0.5, 0.25, 0.75, 1.0], [0.3, 0.301, 0.302, 0.9], [-0.5, -0.501, -0.2, 0.7, 0.701], []]
    for f, dim, layout in cases:
        n = 77
        y0 = rng.uniform(0.5, 1.5, (n, dim)) + (np.array([0.0, 0.0, 20.0]) if dim == 3 else 0.0)
        y0l = torch.from_numpy(np.ascontiguousarray(y0 if layout == 1 else y0.T) if dim > 1 else y0[:, 0].copy()).to(dev)
        for ts in grids:
            for tstart in (0.0, 0.25):
                opt = nn.newODEoptions(dt=0.0625 if len(ts) != 4 else 0.01, tStart=tstart)   # dt > spacing: the reference drops rows
                t, y, ny, ns = nn.fixedStreamSolve(f, y0l, ts, opt, integrator=integrator, layout=layout)
                tf, yf, cf = nn.solveODE(f, y0l, ts, opt, integrator=integrator, layout=layout, return_counts=True)
                assert np.array_equal(t, tf), (dim, ts)
                assert torch.equal(torch.nan_to_num(y, nan=-7.0), torch.nan_to_num(yf, nan=-7.0)), (dim, layout, ts, tstart)
                assert bool((cf["ny"] == ny).all()) and bool((cf["steps"] == ns).all()), (dim, ts, tstart)


def test_fixed_stream_dense_reports_truncation(nn, dev):
    """when max_steps cuts a direction short, nnhip_ode_fixed_stream_dense_f64_dev says so (NNHIP_TRUNCATED, a warning in the
    Python mirror) instead of returning OK with a last row that is not y(tEnd); the rows equal the fused solve's, which flags `truncated`."""
    import ctypes as C
    import warnings
    import torch
    y0 = torch.from_numpy(1.0 + np.arange(40) * 2.0 ** -7).to(dev)
    opt = nn.newODEoptions(dt=2.0 ** -6)
    ts = [0.0, 0.5, 1.0]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        t, y, ny, ns = nn.fixedStreamSolve(nn.Rhs.linear(-0.1), y0, ts, opt, integrator="rk4", max_steps=40)
    assert ns == 40 and any("max_steps" in str(x.message) for x in w)
    tf, yf = nn.solveODE(nn.Rhs.linear(-0.1), y0, ts, opt, integrator="rk4", max_steps=40)
    assert torch.equal(torch.nan_to_num(y, nan=-7.0), torch.nan_to_num(yf, nan=-7.0))
    with warnings.catch_warnings(record=True) as w:   # enough steps: no warning
        warnings.simplefilter("always")
        t, y, ny, ns = nn.fixedStreamSolve(nn.Rhs.linear(-0.1), y0, ts, opt, integrator="rk4", max_steps=64)
    assert ns == 64 and not w


def test_per_step_seams_refuse_companions_they_would_misread(nn, dev):
    """On the device: a float32 state, an FSAL / scratch buffer of another shape, dtype or device, a 0-d or wrong-length (t, dt) tensor — each would make a kernel read or
    write outside what it was given; the Python mirror refuses them (ValueError) and the well-formed call next to each still works."""
    import torch
    f = nn.Rhs.lorenz()
    y = torch.ones((3, 64), dtype=torch.float64, device=dev)
    opt = nn.newODEoptions(dt=2.0 ** -6)
    for bad in (y.float(), y.cpu()):
        with pytest.raises(ValueError, match="float64 tensor on a CUDA/HIP device"):
            nn.integratorStep(f, 0.0, bad, None, 0.01, opt, integrator="rk4")
        with pytest.raises(ValueError, match="float64 tensor on a CUDA/HIP device"):
            nn.adaptiveStream(f, bad, 0.0, 0.1)
    with pytest.raises(ValueError, match="FSAL must"):
        nn.integratorStep(f, 0.0, y, y[:, :32].contiguous(), 0.01, opt, integrator="dopri54")
    with pytest.raises(ValueError, match="FSAL must"):
        nn.integratorStep(f, 0.0, y, y.float(), 0.01, opt, integrator="dopri54")
    for bad_t in (torch.zeros((), dtype=torch.float64, device=dev), torch.zeros(63, dtype=torch.float64, device=dev), torch.zeros(64, dtype=torch.float32, device=dev)):
        with pytest.raises(ValueError, match="Python float or a float64 tensor"):
            nn.integratorStep(f, bad_t, y, None, 0.01, opt, integrator="rk4")
        with pytest.raises(ValueError, match="Python float or a float64 tensor"):
            nn.integratorStep(f, 0.0, y, None, bad_t, opt, integrator="rk4")
    yy = torch.ones(1000, dtype=torch.float64, device=dev)
    with pytest.raises(ValueError, match="scratch must"):
        nn.fixedStream(nn.Rhs.neg_y(), yy, 0.0, 0.25, opt, scratch=torch.empty(999, dtype=torch.float64, device=dev))
    with pytest.raises(ValueError, match="scratch must"):
        nn.fixedStream(nn.Rhs.neg_y(), yy, 0.0, 0.25, opt, scratch=yy)
    # the well-formed neighbours
    r = nn.integratorStep(f, torch.zeros(64, dtype=torch.float64, device=dev), y, y.clone(), torch.full((64,), 0.01, dtype=torch.float64, device=dev), opt, integrator="dopri54")
    assert r[0].shape == y.shape and bool(torch.isfinite(r[0]).all())
    yf, ns = nn.fixedStream(nn.Rhs.neg_y(), yy, 0.0, 0.25, opt, scratch=torch.empty_like(yy))
    assert ns == 16 and bool(torch.isfinite(yf).all())
