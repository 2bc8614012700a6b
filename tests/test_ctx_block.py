"""NumContext in full for run-time compiled right-hand sides (commonTypes.nim:4-27; ode.nim:36 ODEProc's `ctx`, :599 "IT IS MUTABLE"):
any number of fValues, tValues entries shared by the batch or one per IVP (each member of the batch its own ctx), and per-IVP mutable
slots — nnhip_ode_rhs_compile_ctx / nnhip_ode_rhs_bind_ctx_f64_dev.  Parity: bit for bit against the oracle's closures with the same ctx."""
import os

import numpy as np
import pytest

# dy = s * (A y) + g: A is THIS IVP's 16 x 16 matrix (a per-IVP tValues entry), g a forcing vector every IVP shares, s an fValue.
# Association as the oracle's closure (oracle/ode_oracle.cpp RHS_MATVEC): ((A_r0 y_0 + A_r1 y_1) + ...), then s * acc + g_r.
MATVEC_SRC = ("for (int r = 0; r < dim; ++r) { double acc = A(r * dim) * y[0]; for (int k = 1; k < dim; ++k) acc = acc + A(r * dim + k) * y[k];"
              " dy[r] = p[0] * acc + g[r]; }")
# Lorenz whose closure mutates its ctx: aux(0) counts crossings of z = 25 between consecutive calls, aux(1) = z of the last call, aux(2) = calls
ZCROSS_SRC = ("const double z = y[2]; if (aux(2) > 0.0 && (aux(1) - 25.0) * (z - 25.0) < 0.0) aux(0) = aux(0) + 1.0; aux(1) = z; aux(2) = aux(2) + 1.0;"
              " dy[0] = p[0] * (y[1] - y[0]); dy[1] = y[0] * (p[1] - y[2]) - y[1]; dy[2] = y[0] * y[1] - p[2] * y[2];")
DUFF12_SRC = "const double x = y[0], v = y[1]; dy[0] = v; dy[1] = ((-p[8]*v - p[9]*x) - p[10]*(x*x*x)) + p[11]*t;"


def _matvec(nn):
    return nn.Rhs.custom(16, MATVEC_SRC, keys=("s",), tvalues={"g": 16, "A": 256}, per_ivp=("A",), name="matvec16")


def test_layout_declarations_are_checked_without_gpu(nn):
    import ctypes as C
    L = nn._lib.lib()
    f = _matvec(nn)
    assert f.kind >= 1000 and f.ctx_layout["per_ivp"] == ("A",)
    with pytest.raises(ValueError, match="reserved"):
        nn.Rhs.custom(2, "dy[0] = y[1]; dy[1] = 0;", tvalues={"dy": 3})
    with pytest.raises(ValueError, match="identifier"):
        nn.Rhs.custom(2, "dy[0] = y[1]; dy[1] = 0;", tvalues={"2x": 3})
    with pytest.raises(ValueError, match="whole-vector"):
        nn.Rhs.custom(16, "return -y[c];", per_component=True, n_aux=1)
    with pytest.raises(ValueError, match="per_ivp names"):
        nn.Rhs.custom(2, "dy[0] = y[1]; dy[1] = 0;", tvalues={"a": 3}, per_ivp=("b",))
    # a body that uses an undeclared vector does not compile
    with pytest.raises(ValueError, match="undeclared identifier"):
        nn.Rhs.custom(2, "dy[0] = B[0]; dy[1] = 0;", tvalues={"A": 3})
    # binding what the layout does not declare is refused; so is a right-hand side compiled without a layout
    assert L.nnhip_ode_rhs_bind_ctx_f64_dev(f.kind, None, 16, None, 256, None, 0, 10) != 0   # NULL parts
    assert L.nnhip_ode_rhs_bind_ctx_f64_dev(f.kind, C.c_void_p(16), 15, C.c_void_p(16), 256, None, 0, 10) != 0  # wrong shared length
    assert b"layout" in L.nnhip_last_error()
    plain = nn.Rhs.custom(2, "dy[0] = y[1]; dy[1] = -y[0];")
    assert L.nnhip_ode_rhs_bind_ctx_f64_dev(plain.kind, None, 0, None, 0, None, 0, 0) != 0
    # more than eight scalars need the layout entry (the plain one still refuses them)
    k = C.c_int(0)
    assert L.nnhip_ode_rhs_compile(b"x", 2, 12, DUFF12_SRC.encode(), C.byref(k)) != 0
    f12 = nn.Rhs.custom(2, DUFF12_SRC, keys=tuple("k%02d" % i for i in range(12)))
    assert f12.kind >= 1000 and f12.params(None) == []   # the scalars lead the shared block instead of travelling as kernel arguments


@pytest.mark.gpu
def test_unbound_or_too_small_context_is_refused(nn, dev):
    import torch
    f = nn.Rhs.custom(3, ZCROSS_SRC, keys=("sigma", "rho", "beta"), defaults=dict(sigma=10.0, rho=28.0, beta=8.0 / 3.0), n_aux=3, name="zcross_unbound_probe")
    y0 = torch.ones((3, 10), dtype=torch.float64, device=dev)
    with pytest.raises(ValueError, match="pass ctx|must be a contiguous float64 CUDA tensor"):
        nn.solveODE(f, y0, [0.0, 0.1])
    ctx = nn.newNumContext(tValues={"aux": torch.zeros((3, 4), dtype=torch.float64, device=dev)})   # bound for 4 IVPs, asked for 10
    with pytest.raises(ValueError, match="exceeds the batch"):
        nn.solveODE(f, y0, [0.0, 0.1], ctx=ctx)


@pytest.mark.gpu
def test_each_system_its_own_16x16_matrix_1e5(nn, oracle, dev):
    """1e5 systems y' = s A_i y + g, every one with its own 16 x 16 A_i (ctx.tValues), through Rhs.custom — fused solve
    (2-point and dense tspan, both layouts), the IntegratorProc seam and the HBM-resident adaptive loop — equal to the oracle's N closures."""
    import torch
    O = oracle
    n, d = 100_000, 16
    rng = np.random.default_rng(11)
    A = (rng.standard_normal((n, d, d)) * 0.35 - 0.6 * np.eye(d)[None])            # [N, 16, 16]
    g = rng.standard_normal(d) * 0.2
    s = 0.75
    y0 = 0.5 + rng.random((d, n))
    per = np.ascontiguousarray(A.reshape(n, d * d).T)                                # [256, N]: row r*16+c of IVP i
    f = _matvec(nn)
    ctx = nn.newNumContext(fValues={"s": s}, tValues={"g": g, "A": torch.from_numpy(per).to(dev)})
    kw = dict(absTol=1e-7, relTol=1e-7, dtMin=1e-8, dtMax=0.25)
    threads = min(64, os.cpu_count() or 1)
    for integ, ts in (("tsit54", [0.0, 1.0]), ("dopri54", [0.0, 0.2, 0.4, 0.7, 1.0])):
        ref = O.solve_ode_batch_ctx(O.RHS_MATVEC, [s] + list(g), per, None, y0, n, d, ts, O.new_options(**kw), integ, n_threads=threads)
        t, y, cnt = nn.solveODE(f, torch.from_numpy(y0).to(dev), ts, nn.newODEoptions(**kw), ctx=ctx, integrator=integ, return_counts=True)
        assert np.array_equal(t, ref["t"])
        assert np.array_equal(y.cpu().numpy(), ref["y"]), integ
        assert np.array_equal(cnt["steps"].cpu().numpy(), ref["steps"]) and np.array_equal(cnt["rejected"].cpu().numpy(), ref["rejected"])
    # AoS layout, a sub-batch (N < the bound stride: IVP i reads column i), and the streaming seam
    m = 4096
    y0a = torch.from_numpy(np.ascontiguousarray(y0[:, :m].T)).to(dev)
    ta, ya = nn.solveODE(f, y0a, [0.0, 1.0], nn.newODEoptions(**kw), ctx=ctx, integrator="tsit54", layout=1)
    refm = O.solve_ode_batch_ctx(O.RHS_MATVEC, [s] + list(g), per[:, :m], None, y0[:, :m], m, d, [0.0, 1.0], O.new_options(**kw), "tsit54", n_threads=threads)
    assert np.array_equal(ya.cpu().numpy()[-1], refm["y"][-1].T)
    ys, launches = nn.adaptiveStream(f, torch.from_numpy(np.ascontiguousarray(y0[:, :m])).to(dev), 0.0, 1.0, nn.newODEoptions(**kw), ctx=ctx, integrator="tsit54")
    assert np.array_equal(ys.cpu().numpy(), refm["y"][-1]) and launches >= int(refm["steps"].max())


@pytest.mark.gpu
def test_twelve_scalars_live_in_the_shared_block(nn, oracle, dev):
    """ctx.fValues of any size: twelve keys, the right-hand side reads the last four (a Duffing oscillator the oracle has as a closure)."""
    import torch
    O = oracle
    keys = tuple("k%02d" % i for i in range(12))
    vals = dict(zip(keys, [9.0, 8.0, 7.0, 6.0, 5.0, 4.0, 3.0, 2.0, 0.3, -1.0, 1.0, 0.37]))
    f = nn.Rhs.custom(2, DUFF12_SRC, keys=keys, name="duffing12")
    ctx = nn.newNumContext(fValues=vals)
    rng = np.random.default_rng(4)
    n = 3000
    y0 = rng.uniform(-1.5, 1.5, (2, n))
    ts = O.linspace(-1.0, 2.0, 31)
    kw = dict(dt=1e-2, absTol=1e-8, relTol=1e-8, dtMin=1e-6, dtMax=5e-2)
    for integ in ("rk4", "tsit54", "vern65"):
        t, y = nn.solveODE(f, torch.from_numpy(y0).to(dev), ts, nn.newODEoptions(**kw), ctx=ctx, integrator=integ)
        ref = O.solve_ode_batch(O.RHS_DUFFING, [0.3, -1.0, 1.0, 0.37], y0, n, 2, ts, O.new_options(**kw), integ, n_threads=8)
        assert np.array_equal(t, ref["t"]) and np.array_equal(y.cpu().numpy(), ref["y"]), integ


@pytest.mark.gpu
@pytest.mark.parametrize("integrator", ["dopri54", "tsit54", "vern65", "bs32", "rk21", "rk4", "heun2"])
def test_mutable_ctx_counts_z_crossings(nn, oracle, dev, integrator):
    """the closure mutates its ctx (ode.nim:599).  A Lorenz right-hand side that counts, across its own calls, how often z
    crossed 25 — per IVP, in aux — must end with the oracle's counts, last z and number of calls: the device evaluates f where ODESolver does."""
    import torch
    O = oracle
    n = 4000
    y0 = np.stack([1.0 + (np.arange(n) % 1024) * 2.0 ** -10, np.ones(n), np.ones(n)])
    kw = dict(dt=2e-3, absTol=1e-6, relTol=1e-6, dtMin=1e-8, dtMax=0.05)
    f = nn.Rhs.custom(3, ZCROSS_SRC, keys=("sigma", "rho", "beta"), defaults=dict(sigma=10.0, rho=28.0, beta=8.0 / 3.0), n_aux=3, name="lorenz_zcross")
    aux = torch.zeros((3, n), dtype=torch.float64, device=dev)
    ctx = nn.newNumContext(tValues={"aux": aux})
    t, y = nn.solveODE(f, torch.from_numpy(y0).to(dev), [0.0, 3.0], nn.newODEoptions(**kw), ctx=ctx, integrator=integrator)
    ref = O.solve_ode_batch_ctx(O.RHS_LORENZ_ZCROSS, [10.0, 28.0, 8.0 / 3.0], None, np.zeros((3, n)), y0, n, 3, [0.0, 3.0], O.new_options(**kw), integrator, n_threads=8)
    assert np.array_equal(y.cpu().numpy(), ref["y"])
    got = aux.cpu().numpy()
    assert np.array_equal(got, ref["aux"]), (got[:, :3], ref["aux"][:, :3])
    assert got[0].max() >= 2 and got[2].min() > 100          # it did count something
    if integrator in ("dopri54", "tsit54"):                   # the same through the HBM-resident loop: FSAL from the RHS batch kernel, then one launch per iteration
        aux2 = torch.zeros((3, n), dtype=torch.float64, device=dev)
        ys, launches = nn.adaptiveStream(f, torch.from_numpy(y0).to(dev), 0.0, 3.0, nn.newODEoptions(**kw), ctx=nn.newNumContext(tValues={"aux": aux2}), integrator=integrator)
        assert np.array_equal(ys.cpu().numpy(), ref["y"][-1]) and np.array_equal(aux2.cpu().numpy(), ref["aux"])


@pytest.mark.gpu
@pytest.mark.parametrize("integrator", ["dopri54", "tsit54", "vern65"])
def test_mutable_ctx_sees_the_reference_order_of_both_directions(nn, oracle, dev, integrator):
    """tspan straddling tStart: the reference integrates the forward branch, THEN the backward one (ode.nim:508-542, 544-584), on the same
    mutable ctx.  The fused kernel and the dense streaming driver run them in that order too (round 3), so a ctx-mutating closure ends with
    the oracle's slots — here with requested rows on both sides (the FSAL methods' dense output evaluates f nowhere else)."""
    import torch
    O = oracle
    n = 1500
    y0 = np.stack([1.0 + (np.arange(n) % 512) * 2.0 ** -9, np.ones(n), np.full(n, 20.0)])
    kw = dict(absTol=1e-6, relTol=1e-6, dtMin=1e-8, dtMax=0.05)
    f = nn.Rhs.custom(3, ZCROSS_SRC, keys=("sigma", "rho", "beta"), defaults=dict(sigma=10.0, rho=28.0, beta=8.0 / 3.0), n_aux=3, name="lorenz_zcross")
    ts = [-0.4, -0.1, 0.0, 0.5, 1.5]
    aux = torch.zeros((3, n), dtype=torch.float64, device=dev)
    t, y = nn.solveODE(f, torch.from_numpy(y0).to(dev), ts, nn.newODEoptions(**kw), ctx=nn.newNumContext(tValues={"aux": aux}), integrator=integrator)
    ref = O.solve_ode_batch_ctx(O.RHS_LORENZ_ZCROSS, [10.0, 28.0, 8.0 / 3.0], None, np.zeros((3, n)), y0, n, 3, ts, O.new_options(**kw), integrator, n_threads=8)
    assert np.array_equal(t, ref["t"]) and np.array_equal(y.cpu().numpy(), ref["y"], equal_nan=True)
    assert np.array_equal(aux.cpu().numpy(), ref["aux"])
    # the same through the IntegratorProc seam (dense streaming driver): forward direction first there too, two evaluations at t0 before it
    aux2 = torch.zeros((3, n), dtype=torch.float64, device=dev)
    t2, y2, ny, launches = nn.adaptiveStreamSolve(f, torch.from_numpy(y0).to(dev), ts, nn.newODEoptions(**kw), ctx=nn.newNumContext(tValues={"aux": aux2}), integrator=integrator)
    assert np.array_equal(t2, ref["t"]) and np.array_equal(y2.cpu().numpy(), ref["y"], equal_nan=True)
    assert np.array_equal(aux2.cpu().numpy(), ref["aux"])


@pytest.mark.gpu
@pytest.mark.parametrize("integrator", ["rk4", "heun2", "kutta3", "ralston4", "rk21"])
def test_mutable_ctx_call_sequence_of_methods_without_fsal_with_dense_output(nn, oracle, dev, integrator):
    """Methods that do not use FSAL for their dense output make the reference evaluate f once more per step (lastIter.dy, ode.nim:530) and
    once per emitted point (:521).  For a pure f the device makes those evaluations lazily (same values); a right-hand side that mutates
    its ctx gets every one of them where the reference makes it (RhsMutates): slots and rows equal to the oracle's closures, on both
    sides of tStart, with requested times closer than the steps (several points per step) — fused solve, and for the adaptive RK21 also
    the dense streaming driver."""
    import torch
    O = oracle
    n = 700
    y0 = np.stack([1.0 + (np.arange(n) % 256) * 2.0 ** -8, np.ones(n), np.full(n, 20.0)])
    kw = dict(dt=1e-2, absTol=1e-5, relTol=1e-5, dtMin=1e-6, dtMax=0.05)
    f = nn.Rhs.custom(3, ZCROSS_SRC, keys=("sigma", "rho", "beta"), defaults=dict(sigma=10.0, rho=28.0, beta=8.0 / 3.0), n_aux=3, name="lorenz_zcross")
    for ts in ([0.0, 0.3, 0.301, 0.302, 0.9], [-0.2, -0.1, 0.0, 0.25, 0.2501, 0.6]):
        aux = torch.zeros((3, n), dtype=torch.float64, device=dev)
        t, y = nn.solveODE(f, torch.from_numpy(y0).to(dev), ts, nn.newODEoptions(**kw), ctx=nn.newNumContext(tValues={"aux": aux}), integrator=integrator)
        ref = O.solve_ode_batch_ctx(O.RHS_LORENZ_ZCROSS, [10.0, 28.0, 8.0 / 3.0], None, np.zeros((3, n)), y0, n, 3, ts, O.new_options(**kw), integrator, n_threads=8)
        assert np.array_equal(t, ref["t"]) and np.array_equal(y.cpu().numpy(), ref["y"], equal_nan=True), (integrator, ts)
        assert np.array_equal(aux.cpu().numpy(), ref["aux"]), (integrator, ts, aux[2, :4].tolist(), ref["aux"][2, :4].tolist())
        if integrator != "rk21":  # the fixed-step methods also through the host-driven dense streaming entry (forward direction first there too)
            aux3 = torch.zeros((3, n), dtype=torch.float64, device=dev)
            t3, y3, ny3, ns3 = nn.fixedStreamSolve(f, torch.from_numpy(y0).to(dev), ts, nn.newODEoptions(**kw), ctx=nn.newNumContext(tValues={"aux": aux3}), integrator=integrator)
            assert np.array_equal(y3.cpu().numpy()[:ny3], ref["y"][:ny3], equal_nan=True) and np.array_equal(aux3.cpu().numpy(), ref["aux"]), (integrator, ts, aux3[2, :4].tolist(), ref["aux"][2, :4].tolist())
        if integrator == "rk21":
            aux2 = torch.zeros((3, n), dtype=torch.float64, device=dev)
            t2, y2, ny, launches = nn.adaptiveStreamSolve(f, torch.from_numpy(y0).to(dev), ts, nn.newODEoptions(**kw), ctx=nn.newNumContext(tValues={"aux": aux2}),
                                                          integrator=integrator)
            assert np.array_equal(y2.cpu().numpy(), ref["y"], equal_nan=True) and np.array_equal(aux2.cpu().numpy(), ref["aux"]), (ts, aux2[2, :4].tolist(), ref["aux"][2, :4].tolist())


def _host_bind(nn, f, shared, per, aux, stride, device=0):
    import ctypes as C
    dp = C.POINTER(C.c_double)
    L = nn._lib.lib()
    ptr = lambda a: None if a is None else a.ctypes.data_as(dp)  # noqa: E731
    rc = L.nnhip_ode_rhs_bind_ctx_f64(f.kind, ptr(shared), 0 if shared is None else shared.size, ptr(per), 0 if per is None else per.shape[0],
                                      ptr(aux), 0 if aux is None else aux.shape[0], stride, device)
    assert rc == 0, nn._lib.last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("n_shards", [2, 3, 8])
def test_per_ivp_matrices_travel_with_their_shard(nn, oracle, dev, n_shards):
    """the one-call multi-GPU entry on a batch whose members each carry their own 16 x 16 matrix (a per-IVP ctx.tValues entry,
    commonTypes.nim:4-6) and share a forcing vector: the context block bound from host arrays is cut into the shards' column ranges — shard r's
    device gets columns [lo_r, hi_r) — and the result equals the oracle's N closures and the single-device solve, bit for bit.  More shards than
    devices on a one-GPU box (knob multi_gpu_oversubscribe); one shard per device where there are enough."""
    import ctypes as C
    import torch
    O = oracle
    L = nn._lib.lib()
    dp = C.POINTER(C.c_double)
    n, d = 1003, 16   # ragged shards
    rng = np.random.default_rng(23)
    A = (rng.standard_normal((n, d, d)) * 0.35 - 0.6 * np.eye(d)[None])
    g = rng.standard_normal(d) * 0.2
    s = 0.75
    y0 = np.ascontiguousarray(0.5 + rng.random((d, n)))
    per = np.ascontiguousarray(A.reshape(n, d * d).T)
    f = _matvec(nn)
    kw = dict(absTol=1e-7, relTol=1e-7, dtMin=1e-8, dtMax=0.25)
    ts = np.array([0.0, 0.3, 1.0])
    ref = O.solve_ode_batch_ctx(O.RHS_MATVEC, [s] + list(g), per, None, y0, n, d, ts, O.new_options(**kw), "tsit54", n_threads=min(32, os.cpu_count() or 1))
    _host_bind(nn, f, np.ascontiguousarray(g), per, None, n)
    opt = nn.newODEoptions(**kw)
    p = np.array([s])
    out = np.full((len(ts), d, n), -7.0)
    t_out = np.empty(len(ts))
    ny = np.empty(n, dtype=np.int32)
    st = nn.ode.Stats()
    try:
        L.nnhip_tune_set(b"multi_gpu_oversubscribe", 1 if n_shards > torch.cuda.device_count() else 0)
        rc = L.nnhip_ode_solve_batch_multi_gpu_f64(C.byref(opt), nn.ode.integrator_id("tsit54"), f.kind, p.ctypes.data_as(dp), 1, y0.ctypes.data, n, d, 0,
                                                   ts.ctypes.data_as(dp), len(ts), t_out.ctypes.data_as(dp), out.ctypes.data, ny.ctypes.data, 0, C.byref(st), n_shards)
        assert rc == 0, nn._lib.last_error()
    finally:
        L.nnhip_tune_set(b"multi_gpu_oversubscribe", 0)
    assert np.array_equal(t_out, ref["t"]) and np.array_equal(out, ref["y"]) and np.array_equal(ny, ref["ny"])
    assert st.steps_total == int(ref["steps"].sum())
    # the same binding still serves a single-device call of the calling thread (host-pointer entry, whole batch)
    out1 = np.full_like(out, -9.0)
    rc = L.nnhip_ode_solve_batch_f64(C.byref(opt), nn.ode.integrator_id("tsit54"), f.kind, p.ctypes.data_as(dp), 1, y0.ctypes.data, n, d, 0, ts.ctypes.data_as(dp), len(ts),
                                     t_out.ctypes.data_as(dp), out1.ctypes.data, None, None, None, 0, None, 0)
    assert rc == 0 and np.array_equal(out1, out)
    # a block bound as device pointers of one device cannot be cut: refused with the way out
    ctx = nn.newNumContext(fValues={"s": s}, tValues={"g": g, "A": torch.from_numpy(per).to(dev)})
    f.bind(ctx, dev)
    try:
        L.nnhip_tune_set(b"multi_gpu_oversubscribe", 1)
        rc = L.nnhip_ode_solve_batch_multi_gpu_f64(C.byref(opt), nn.ode.integrator_id("tsit54"), f.kind, p.ctypes.data_as(dp), 1, y0.ctypes.data, n, d, 0,
                                                   ts.ctypes.data_as(dp), len(ts), t_out.ctypes.data_as(dp), out.ctypes.data, ny.ctypes.data, 0, C.byref(st), 2)
        assert rc == nn._lib.NNHIP_EUNSUPPORTED and "host arrays" in nn._lib.last_error()
    finally:
        L.nnhip_tune_set(b"multi_gpu_oversubscribe", 0)


@pytest.mark.gpu
def test_mutable_slots_come_back_from_their_shards(nn, oracle, dev):
    """A closure that mutates its ctx (ode.nim:599), sharded: every shard's device updates its own columns of the mutable slots, and
    nnhip_ode_rhs_read_aux_f64 returns them in the caller's order — the oracle's closures' final environments."""
    import ctypes as C
    O = oracle
    L = nn._lib.lib()
    dp = C.POINTER(C.c_double)
    n = 501
    f = nn.Rhs.custom(3, ZCROSS_SRC, keys=("sigma", "rho", "beta"), defaults=dict(sigma=10.0, rho=28.0, beta=8.0 / 3.0), n_aux=3, name="zcross_sharded")
    y0 = np.ascontiguousarray(np.stack([1.0 + np.arange(n) * 1e-3, np.ones(n), np.ones(n) * 20.0]))
    kw = dict(absTol=1e-6, relTol=1e-6, dtMin=1e-9, dtMax=0.1)
    ts = np.array([0.0, 3.0])
    aux0 = np.zeros((3, n))
    ref = O.solve_ode_batch_ctx(O.RHS_LORENZ_ZCROSS, [10.0, 28.0, 8.0 / 3.0], None, aux0.copy(), y0, n, 3, ts, O.new_options(**kw), "dopri54", n_threads=8)
    ref_aux = ref["aux"]
    assert ref_aux[0].max() >= 1.0                        # crossings were counted
    _host_bind(nn, f, None, None, aux0, n)
    opt = nn.newODEoptions(**kw)
    p = np.array([10.0, 28.0, 8.0 / 3.0])
    out = np.empty((2, 3, n))
    t_out = np.empty(2)
    try:
        L.nnhip_tune_set(b"multi_gpu_oversubscribe", 1)
        rc = L.nnhip_ode_solve_batch_multi_gpu_f64(C.byref(opt), nn.ode.integrator_id("dopri54"), f.kind, p.ctypes.data_as(dp), 3, y0.ctypes.data, n, 3, 0,
                                                   ts.ctypes.data_as(dp), 2, t_out.ctypes.data_as(dp), out.ctypes.data, None, 0, None, 3)
        assert rc == 0, nn._lib.last_error()
    finally:
        L.nnhip_tune_set(b"multi_gpu_oversubscribe", 0)
    got = np.empty((3, n))
    assert L.nnhip_ode_rhs_read_aux_f64(f.kind, got.ctypes.data_as(dp)) == 0
    assert np.array_equal(out, ref["y"]) and np.array_equal(got, ref_aux)


@pytest.mark.gpu
def test_two_threads_bind_different_contexts_to_one_source(nn, oracle, dev):
    """a binding belongs to the thread that made it.  Two host threads solve the SAME compiled source with DIFFERENT
    contexts at the same time, many times over: each must get its own context's result every time (was: process-wide per rhs_kind — a bind
    of the other thread between this thread's bind and its launch changed what the launch read)."""
    import threading
    import torch
    O = oracle
    on_isa_node = bool(os.environ.get("FAKE_HIP_LIB"))   # (scripts/run_gpu_suite_on_isa_node.py: the interpreter wants a smaller batch and fewer repeats; a device gets the full ones)
    n, d = (40 if on_isa_node else 300), 16
    reps = 6 if on_isa_node else 40
    f = _matvec(nn)
    kw = dict(absTol=1e-7, relTol=1e-7, dtMin=1e-8, dtMax=0.25)
    rng = np.random.default_rng(5)
    y0 = 0.5 + rng.random((d, n))
    jobs = []
    for k in range(2):
        A = (rng.standard_normal((n, d, d)) * 0.35 - 0.6 * np.eye(d)[None])
        g = rng.standard_normal(d) * 0.2
        per = np.ascontiguousarray(A.reshape(n, d * d).T)
        ref = O.solve_ode_batch_ctx(O.RHS_MATVEC, [0.75] + list(g), per, None, y0, n, d, [0.0, 1.0], O.new_options(**kw), "tsit54", n_threads=8)["y"][-1]
        jobs.append((g, per, ref))
    bad, errors = [0, 0], []

    def worker(k):
        try:
            g, per, ref = jobs[k]
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                ctx = nn.newNumContext(fValues={"s": 0.75}, tValues={"g": g, "A": torch.from_numpy(per).to(dev)})
                yt = torch.from_numpy(y0).to(dev)
                for _ in range(reps):
                    t, y = nn.solveODE(f, yt, [0.0, 1.0], nn.newODEoptions(**kw), ctx=ctx, integrator="tsit54")
                    if not np.array_equal(y[-1].cpu().numpy(), ref):
                        bad[k] += 1
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errors, errors
    assert bad == [0, 0], bad


@pytest.mark.gpu
def test_device_resident_shards_read_their_columns_of_the_context(nn, oracle, dev):
    """The two one-call entries whose shards live in device memory (nnhip_ode_solve_batch_multi_gpu_f64_dev, nnhip_ode_fixed_stream_multi_gpu_f64_dev),
    on a batch with per-IVP matrices: shard r reads columns [lo_r, hi_r) of the host-bound context block.  Three ragged shards (one of them empty)
    over-subscribing this box's devices, without the RCCL reassembly (RCCL wants one device per rank): every shard equals the oracle's closures."""
    import ctypes as C
    import torch
    O = oracle
    L = nn._lib.lib()
    dp = C.POINTER(C.c_double)
    counts = [257, 0, 300, 64]
    G, n, d = len(counts), sum(counts), 16
    ndev = torch.cuda.device_count()
    rng = np.random.default_rng(29)
    A = (rng.standard_normal((n, d, d)) * 0.35 - 0.6 * np.eye(d)[None])
    g = rng.standard_normal(d) * 0.2
    s = 0.75
    y0 = np.ascontiguousarray(0.5 + rng.random((d, n)))
    per = np.ascontiguousarray(A.reshape(n, d * d).T)
    f = _matvec(nn)
    _host_bind(nn, f, np.ascontiguousarray(g), per, None, n)
    kw = dict(absTol=1e-7, relTol=1e-7, dtMin=1e-8, dtMax=0.25)
    opt = nn.newODEoptions(**kw)
    p = np.array([s])
    lo = np.concatenate([[0], np.cumsum(counts)])
    arr = lambda xs: (C.c_void_p * G)(*xs)  # noqa: E731
    devs = [torch.device("cuda", r % ndev) for r in range(G)]
    streams = [torch.cuda.Stream(device=devs[r]) for r in range(G)]
    try:
        L.nnhip_tune_set(b"multi_gpu_oversubscribe", 1)
        # fused adaptive solve, dense tspan
        ts = np.array([0.0, 0.4, 1.0])
        ref = O.solve_ode_batch_ctx(O.RHS_MATVEC, [s] + list(g), per, None, y0, n, d, ts, O.new_options(**kw), "dopri54", n_threads=8)
        y0s = [torch.from_numpy(np.ascontiguousarray(y0[:, lo[r]:lo[r + 1]])).to(devs[r]) for r in range(G)]
        outs = [torch.full((len(ts), d, counts[r]), -3.0, dtype=torch.float64, device=devs[r]) for r in range(G)]
        wsb = int(L.nnhip_ode_solve_workspace_bytes(len(ts)))
        wss = [torch.empty(max(wsb, 8), dtype=torch.uint8, device=devs[r]) for r in range(G)]
        t_out = np.empty(len(ts))
        torch.cuda.synchronize()
        rc = L.nnhip_ode_solve_batch_multi_gpu_f64_dev(C.byref(opt), nn.ode.integrator_id("dopri54"), f.kind, p.ctypes.data_as(dp), 1, G, (C.c_int64 * G)(*counts), d, 0,
                                                       ts.ctypes.data_as(dp), len(ts), t_out.ctypes.data_as(dp), arr([y.data_ptr() for y in y0s]),
                                                       arr([o.data_ptr() for o in outs]), None, 0, arr([w.data_ptr() for w in wss]), wsb, None,
                                                       arr([q.cuda_stream for q in streams]), None)
        assert rc == 0, nn._lib.last_error()
        for q in streams: q.synchronize()
        for r in range(G):
            assert np.array_equal(outs[r].cpu().numpy(), ref["y"][:, :, lo[r]:lo[r + 1]]), r
        # with a gather the over-subscription is refused (one device per RCCL rank)
        fulls = [torch.empty((len(ts), d, n), dtype=torch.float64, device=devs[r]) for r in range(G)]
        rc = L.nnhip_ode_solve_batch_multi_gpu_f64_dev(C.byref(opt), nn.ode.integrator_id("dopri54"), f.kind, p.ctypes.data_as(dp), 1, G, (C.c_int64 * G)(*counts), d, 0,
                                                       ts.ctypes.data_as(dp), len(ts), t_out.ctypes.data_as(dp), arr([y.data_ptr() for y in y0s]),
                                                       arr([o.data_ptr() for o in outs]), None, 0, arr([w.data_ptr() for w in wss]), wsb, arr([x.data_ptr() for x in fulls]),
                                                       arr([q.cuda_stream for q in streams]), None)
        if G > ndev:
            assert rc == nn._lib.NNHIP_EVALUE
        # the step-streaming loop (fixed step) over the same shards
        optf = nn.newODEoptions(dt=2.0 ** -7)
        reff = O.solve_ode_batch_ctx(O.RHS_MATVEC, [s] + list(g), per, None, y0, n, d, [0.0, 0.5], O.new_options(dt=2.0 ** -7), "rk4", n_threads=8)
        ys = [torch.from_numpy(np.ascontiguousarray(y0[:, lo[r]:lo[r + 1]])).to(devs[r]) for r in range(G)]
        scr = [torch.empty_like(y) for y in ys]
        fin = (C.c_void_p * G)()
        nst = C.c_int64(0)
        torch.cuda.synchronize()
        rc = L.nnhip_ode_fixed_stream_multi_gpu_f64_dev(C.byref(optf), nn.ode.integrator_id("rk4"), f.kind, p.ctypes.data_as(dp), 1, G, (C.c_int64 * G)(*counts), d, 0, 0.0, 0.5,
                                                        arr([y.data_ptr() for y in ys]), arr([x.data_ptr() for x in scr]), None, arr([q.cuda_stream for q in streams]), None,
                                                        C.byref(nst), fin)
        assert rc == 0, nn._lib.last_error()
        for q in streams: q.synchronize()
        assert nst.value == 64
        for r in range(G):
            if counts[r] == 0: continue
            got = ys[r] if fin[r] == ys[r].data_ptr() else scr[r]
            assert np.array_equal(got.cpu().numpy(), reff["y"][-1][:, lo[r]:lo[r + 1]]), r
    finally:
        L.nnhip_tune_set(b"multi_gpu_oversubscribe", 0)
