"""nnhip_ode_solve_batch_multi_gpu_f64 / nnhip_allgather_states_f64_dev: the batch shards as contiguous index ranges, one per
GPU, every range being its own set of solveODE calls (ode.nim:589-591).  On a one-GPU box the sharded code path (index ranges of
the caller's arrays, strided copies, empty shards, error propagation from the worker threads) is exercised by over-subscribing
the device (tuning knob multi_gpu_oversubscribe); with >= 2 devices the same tests also run one shard per device, and the RCCL
all-gather runs at n_gpus = device_count."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LOR = [10.0, 28.0, 8.0 / 3.0]
dp = C.POINTER(C.c_double)


def _call(nn, L, y0l, n, dim, layout, ts, n_gpus, integrator=1, rhs=2, params=LOR, opt=None):
    opt = opt or nn.newODEoptions()
    out = np.full((len(ts),) + y0l.shape, -777.0)
    t_out = np.full(len(ts), -1.0)
    ny = np.full(max(n, 1), -5, dtype=np.int32)
    st = nn.ode.Stats()
    p = np.asarray(params, dtype=np.float64)
    rc = L.nnhip_ode_solve_batch_multi_gpu_f64(C.byref(opt), integrator, rhs, p.ctypes.data_as(dp), len(p), y0l.ctypes.data, n, dim, layout,
                                               ts.ctypes.data_as(dp), len(ts), t_out.ctypes.data_as(dp), out.ctypes.data, ny.ctypes.data, 0,
                                               C.byref(st), n_gpus)
    return rc, t_out, out, ny[:n], st


@pytest.mark.parametrize("layout", [0, 1], ids=["soa", "aos"])
@pytest.mark.parametrize("n_shards", [1, 2, 3, 8])
def test_sharded_solve_matches_oracle(nn, oracle, dev, layout, n_shards):
    import torch
    O = oracle
    L = nn._lib.lib()
    ndev = torch.cuda.device_count()
    n, dim = 1001, 3   # not divisible by 2, 3 or 8: ragged shards
    y0 = np.stack([1.0 + (np.arange(n) % 1024) * 2.0 ** -20, np.ones(n), np.ones(n) + np.arange(n) * 1e-4])
    y0l = np.ascontiguousarray(y0 if layout == 0 else y0.T)
    ts = np.array([-0.1, 0.0, 0.2, 0.3])
    ref = O.solve_ode_batch(O.RHS_LORENZ, LOR, y0l, n, dim, ts, O.new_options(), "dopri54", layout=layout, n_threads=8)
    try:
        L.nnhip_tune_set(b"multi_gpu_oversubscribe", 1 if n_shards > ndev else 0)
        rc, t_out, out, ny, st = _call(nn, L, y0l, n, dim, layout, ts, n_shards)
        assert rc == 0, nn._lib.last_error()
        assert np.array_equal(t_out, ref["t"]) and st.n_t_out == len(ref["t"])
        assert np.array_equal(out, ref["y"])                    # bit-exact, every shard in its place
        assert np.array_equal(ny, ref["ny"])
        assert st.steps_total == int(ref["steps"].sum()) and st.rejected_total == int(ref["rejected"].sum())
        assert st.steps_max == int(ref["steps"].max()) and st.ny_min == int(ref["ny"].min())
        # two-point tspan (no dense output) takes the same path
        ts2 = np.array([0.0, 0.25])
        ref2 = O.solve_ode_batch(O.RHS_LORENZ, LOR, y0l, n, dim, ts2, O.new_options(), "tsit54", layout=layout, n_threads=8)
        rc, t_out, out, ny, st = _call(nn, L, y0l, n, dim, layout, ts2, n_shards, integrator=2)
        assert rc == 0 and np.array_equal(out, ref2["y"]) and np.array_equal(t_out, ref2["t"])
    finally:
        L.nnhip_tune_set(b"multi_gpu_oversubscribe", 0)


def test_fewer_ivps_than_shards_and_empty_batch(nn, oracle, dev):
    """N < n_gpus leaves shards empty (the first ones: [r*N/G, (r+1)*N/G)): t_out / n_t_out are still written (they depend on
    options and tspan only) and the non-empty shards land in place.  N = 0 returns the time grid and touches nothing else."""
    O = oracle
    L = nn._lib.lib()
    ts = np.array([0.5, -0.25, 0.0, 1.0])
    try:
        L.nnhip_tune_set(b"multi_gpu_oversubscribe", 1)
        y0 = np.array([[1.0, 1.0, 1.0], [1.5, 0.5, 2.0]])        # N = 2 AoS, 5 shards -> shards 0, 1, 3 are empty
        rc, t_out, out, ny, st = _call(nn, L, y0, 2, 3, 1, ts, 5)
        assert rc == 0, nn._lib.last_error()
        ref = O.solve_ode_batch(O.RHS_LORENZ, LOR, y0, 2, 3, ts, O.new_options(), "dopri54", layout=1)
        assert np.array_equal(t_out, ref["t"]) and st.n_t_out == 4
        assert np.array_equal(out, ref["y"]) and np.array_equal(ny, ref["ny"])
        rc, t_out, out, ny, st = _call(nn, L, np.empty((0, 3)), 0, 3, 1, ts, 3)
        assert rc == 0 and np.array_equal(t_out, np.sort(ts)) and st.n_t_out == 4 and st.steps_total == 0 and st.ny_min == 0
    finally:
        L.nnhip_tune_set(b"multi_gpu_oversubscribe", 0)


def test_worker_errors_reach_the_caller(nn, dev):
    """A failure inside a worker thread is reported through the CALLER's nnhip_last_error(), naming the device and its index range."""
    L = nn._lib.lib()
    y0 = np.ones((10, 3))
    ts = np.array([0.0, 1.0])
    try:
        L.nnhip_tune_set(b"multi_gpu_oversubscribe", 1)
        L.nnhip_ode_integrator_id(b"no_such_integrator")       # leaves an unrelated message behind
        rc, *_ = _call(nn, L, y0, 10, 3, 1, ts, 2, rhs=2, params=[10.0])   # Lorenz needs 3 parameters
        assert rc == nn._lib.NNHIP_EVALUE
        msg = nn._lib.last_error()
        assert "device 0" in msg and "needs 3 parameters" in msg, msg
        rc, *_ = _call(nn, L, y0, 10, 3, 1, ts, 2, opt=nn.newODEoptions(dtMin=0.0))   # would never terminate: refused in the worker
        assert rc != 0 and "device" in nn._lib.last_error()
    finally:
        L.nnhip_tune_set(b"multi_gpu_oversubscribe", 0)
    import torch
    rc, *_ = _call(nn, L, y0, 10, 3, 1, ts, torch.cuda.device_count() + 1)   # more GPUs than the box has -> refused with a message
    assert rc == nn._lib.NNHIP_EVALUE and "HIP device" in nn._lib.last_error()


def test_repeated_calls_do_not_leak_pinned_staging(nn, dev):
    """Every call spawns worker threads that allocate a pinned staging buffer for the requested-time arrays; it is released when
    the worker ends (was: leaked per call per GPU)."""
    import torch
    L = nn._lib.lib()
    y0 = np.ones((64, 3))
    ts = np.linspace(0.0, 0.1, 33)
    try:
        L.nnhip_tune_set(b"multi_gpu_oversubscribe", 1)
        _call(nn, L, y0, 64, 3, 1, ts, 4)
        free0 = torch.cuda.mem_get_info()[0]
        import resource
        rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
        for _ in range(60):
            rc, *_ = _call(nn, L, y0, 64, 3, 1, ts, 4)
            assert rc == 0
        assert torch.cuda.mem_get_info()[0] >= free0 - (8 << 20)
        assert resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - rss0 < 64 * 1024   # KiB
    finally:
        L.nnhip_tune_set(b"multi_gpu_oversubscribe", 0)


@pytest.mark.parametrize("layout,dim", [(0, 1), (0, 3), (1, 3)])
def test_rccl_allgather_states_all_devices(nn, dev, layout, dim):
    """nnhip_allgather_states_f64_dev (one process, G devices, RCCL over xGMI) at n_gpus = device_count: every device ends up with
    the whole tensor, equal and ragged shards.  Needs >= 2 devices (the one-device case is test_rccl_allgather_states_single_device)."""
    import torch
    G = torch.cuda.device_count()
    if G < 2:
        pytest.skip(f"needs >= 2 HIP devices, this box has {G} (the 8-GPU run is the driver's)")
    L = nn._lib.lib()
    rng = np.random.default_rng(3)
    for counts in ([500] * G, [300 + 17 * r for r in range(G)], [0 if r == 1 else 200 + 5 * r for r in range(G)], [0] * (G - 1) + [64]):  # equal, ragged, an empty shard, all but one empty
        N = sum(counts)
        full_ref = rng.normal(size=(dim, N)) if layout == 0 else rng.normal(size=(N, dim))
        lo = np.concatenate([[0], np.cumsum(counts)])
        shards, fulls = [], []
        for r in range(G):
            with torch.cuda.device(r):
                sh = full_ref[:, lo[r]:lo[r + 1]] if layout == 0 else full_ref[lo[r]:lo[r + 1]]
                shards.append(torch.from_numpy(np.ascontiguousarray(sh)).to(f"cuda:{r}"))
                fulls.append(torch.zeros(full_ref.shape, dtype=torch.float64, device=f"cuda:{r}"))
        for r in range(G):
            torch.cuda.synchronize(r)
        sp = (C.c_void_p * G)(*[s.data_ptr() for s in shards])
        fp = (C.c_void_p * G)(*[f.data_ptr() for f in fulls])
        cn = (C.c_int64 * G)(*counts)
        rc = L.nnhip_allgather_states_f64_dev(G, sp, cn, dim, layout, fp, None)
        assert rc == 0, L.nnhip_multigpu_last_error()
        for r in range(G):
            torch.cuda.synchronize(r)
            assert np.array_equal(fulls[r].cpu().numpy(), full_ref), (r, counts)


@pytest.mark.parametrize("layout", [0, 1], ids=["soa", "aos"])
@pytest.mark.parametrize("n_shards", [1, 3, 8])
def test_sharded_parameter_sweep(nn, dev, layout, n_shards):
    """nnhip_ode_solve_batch_multi_gpu_sweep_f64: N solveODE calls with their own ctx each (per-IVP sigma, rho), sharded — every shard
    reads its columns of the caller's [k][N] table in place; bits and per-IVP counters of the single-device sweep."""
    import torch
    L = nn._lib.lib()
    ndev = torch.cuda.device_count()
    n, dim = 1001, 3
    rng = np.random.default_rng(8)
    y0 = np.stack([1.0 + rng.random(n), np.ones(n), np.ones(n)])
    y0l = np.ascontiguousarray(y0 if layout == 0 else y0.T)
    sw = np.ascontiguousarray(np.stack([np.full(n, 10.0), rng.uniform(20, 30, n)]))
    ts = np.array([-0.1, 0.0, 0.2, 0.3])
    tr, yr, cr = nn.solveODE(nn.Rhs.lorenz(), y0l, ts, nn.newODEoptions(), integrator="tsit54", layout=layout, sweep=sw, return_counts=True)
    out = np.full((len(ts),) + y0l.shape, -777.0)
    t_out = np.full(len(ts), -1.0)
    ny = np.full(n, -5, dtype=np.int32)
    steps = np.full(n, -5, dtype=np.int64)
    rej = np.full(n, -5, dtype=np.int64)
    st = nn.ode.Stats()
    p = np.asarray(LOR)
    opt = nn.newODEoptions()
    try:
        L.nnhip_tune_set(b"multi_gpu_oversubscribe", 1 if n_shards > ndev else 0)
        rc = L.nnhip_ode_solve_batch_multi_gpu_sweep_f64(C.byref(opt), 2, 2, p.ctypes.data_as(dp), 3, sw.ctypes.data, 2, y0l.ctypes.data, n, dim, layout,
                                                         ts.ctypes.data_as(dp), len(ts), t_out.ctypes.data_as(dp), out.ctypes.data, ny.ctypes.data,
                                                         steps.ctypes.data, rej.ctypes.data, 0, C.byref(st), n_shards)
        assert rc == 0, nn._lib.last_error()
    finally:
        L.nnhip_tune_set(b"multi_gpu_oversubscribe", 0)
    assert np.array_equal(t_out, tr) and np.array_equal(out, yr)
    assert np.array_equal(ny, cr["ny"]) and np.array_equal(steps, cr["steps"]) and np.array_equal(rej, cr["rejected"])
    assert st.steps_total == int(cr["steps"].sum())


def _c5_composite(nn, L, G, counts, n_steps, layout, dim, rhs, params, integ, use_gather_streams):
    """nnhip_ode_fixed_stream_multi_gpu_f64_dev on G devices; returns (full tensors per device, reference full tensor)."""
    import torch
    N = sum(counts)
    rng = np.random.default_rng(5)
    y0 = 1.0 + rng.random((dim, N)) if layout == 0 else 1.0 + rng.random((N, dim))
    dt = 2.0 ** -10
    opt = nn.newODEoptions(dt=dt)
    tEnd = n_steps * dt
    lo = np.concatenate([[0], np.cumsum(counts)])
    ys, scr, fulls, streams, gstreams = [], [], [], [], []
    for r in range(G):
        with torch.cuda.device(r):
            sh = y0[:, lo[r]:lo[r + 1]] if layout == 0 else y0[lo[r]:lo[r + 1]]
            ys.append(torch.from_numpy(np.ascontiguousarray(sh)).to(f"cuda:{r}"))
            scr.append(torch.empty_like(ys[-1]))
            fulls.append(torch.zeros(y0.shape, dtype=torch.float64, device=f"cuda:{r}"))
            streams.append(torch.cuda.Stream(device=r))
            gstreams.append(torch.cuda.Stream(device=r))
    for r in range(G):
        torch.cuda.synchronize(r)
    arr = lambda xs: (C.c_void_p * G)(*xs)
    pp = np.asarray(params, dtype=np.float64)
    nst = C.c_int64(0)
    fin = (C.c_void_p * G)()
    rc = L.nnhip_ode_fixed_stream_multi_gpu_f64_dev(C.byref(opt), nn.ode.integrator_id(integ), rhs.kind, pp.ctypes.data_as(C.POINTER(C.c_double)) if pp.size else None, int(pp.size),
                                                    G, (C.c_int64 * G)(*counts), dim, layout, 0.0, tEnd, arr([y.data_ptr() for y in ys]),
                                                    arr([s.data_ptr() for s in scr]), arr([f.data_ptr() for f in fulls]),
                                                    arr([s.cuda_stream for s in streams]), arr([s.cuda_stream for s in gstreams]) if use_gather_streams else None,
                                                    C.byref(nst), fin)
    assert rc == 0, L.nnhip_last_error()
    for r in range(G):
        torch.cuda.synchronize(r)
    assert nst.value == n_steps
    for r in range(G):
        assert (fin[r] or 0) in (ys[r].data_ptr(), scr[r].data_ptr())  # (an empty shard's tensors have a null data pointer)
    with torch.cuda.device(0):
        t, yref = nn.solveODE(rhs, torch.from_numpy(y0).to("cuda:0"), [0.0, tEnd], opt, integrator=integ, layout=layout)
    return fulls, yref[-1].cpu().numpy()


@pytest.mark.parametrize("gather_streams", [False, True], ids=["same_stream", "gather_stream"])
@pytest.mark.parametrize("layout,dim", [(0, 1), (0, 3), (1, 3)])
def test_c5_one_call_single_device_rccl(nn, dev, layout, dim, gather_streams):
    """BASELINE config C5 behind ONE C entry (nnhip_ode_fixed_stream_multi_gpu_f64_dev): shard resident on its device, step-streaming solve on
    the device's stream, RCCL reassembly into a device-resident full tensor.  n_gpus = 1 is what a one-GPU box can run: the whole
    composition incl. ncclCommInitAll / ncclAllGather, against the fused solve."""
    L = nn._lib.lib()
    rhs = nn.Rhs.neg_y() if dim == 1 else nn.Rhs.lorenz()
    params = [] if dim == 1 else LOR
    fulls, ref = _c5_composite(nn, L, 1, [4096 if dim == 1 else 1500], 64, layout, dim, rhs, params, "rk4", gather_streams)
    assert np.array_equal(fulls[0].cpu().numpy(), ref)


@pytest.mark.parametrize("ragged", [0, 1, 2], ids=["equal", "ragged", "ragged_with_empty_shard"])
def test_c5_one_call_all_devices(nn, dev, ragged):
    """The same at n_gpus = device_count (equal shards: ncclAllGather; ragged: grouped ncclBroadcasts), gather overlapped on its own
    streams.  Needs >= 2 devices: skipped on the one-GPU boxes, runs on the first multi-GPU lease."""
    import torch
    G = torch.cuda.device_count()
    if G < 2:
        pytest.skip(f"needs >= 2 HIP devices, this box has {G} (the 8-GPU run is the driver's)")
    L = nn._lib.lib()
    counts = [0 if (ragged == 2 and r == G - 1) else 3000 + (37 * r if ragged else 0) for r in range(G)]
    for layout, dim, rhs, params in ((0, 1, nn.Rhs.neg_y(), []), (0, 3, nn.Rhs.lorenz(), LOR), (1, 3, nn.Rhs.lorenz(), LOR)):
        fulls, ref = _c5_composite(nn, L, G, counts, 100, layout, dim, rhs, params, "rk4", True)
        for r in range(G):
            assert np.array_equal(fulls[r].cpu().numpy(), ref), (r, layout, dim)


@pytest.mark.parametrize("layout", [0, 1], ids=["soa", "aos"])
def test_fused_solve_multi_gpu_dev_one_call(nn, dev, layout):
    """nnhip_ode_solve_batch_multi_gpu_f64_dev: device-resident shards, fused adaptive solve with a dense tspan per device, the whole
    trajectory tensor reassembled on every device.  Runs at n_gpus = 1 everywhere and at device_count where there are more."""
    import torch
    L = nn._lib.lib()
    for G in sorted({1, torch.cuda.device_count()}):
        counts = [700 + 13 * r for r in range(G)]
        N, dim = sum(counts), 3
        rng = np.random.default_rng(6)
        y0 = np.stack([1.0 + rng.random(N), np.ones(N), np.ones(N)])
        y0l = np.ascontiguousarray(y0 if layout == 0 else y0.T)
        ts = np.array([-0.1, 0.0, 0.2, 0.3, 0.45])
        n_t = len(ts)
        opt = nn.newODEoptions()
        lo = np.concatenate([[0], np.cumsum(counts)])
        y0s, outs, nys, wss, fulls, streams = [], [], [], [], [], []
        wsb = int(L.nnhip_ode_solve_workspace_bytes(n_t))
        for r in range(G):
            with torch.cuda.device(r):
                sh = y0l[:, lo[r]:lo[r + 1]] if layout == 0 else y0l[lo[r]:lo[r + 1]]
                y0s.append(torch.from_numpy(np.ascontiguousarray(sh)).to(f"cuda:{r}"))
                outs.append(torch.empty((n_t,) + tuple(y0s[-1].shape), dtype=torch.float64, device=f"cuda:{r}"))
                nys.append(torch.empty(counts[r], dtype=torch.int32, device=f"cuda:{r}"))
                wss.append(torch.empty(max(wsb, 8), dtype=torch.uint8, device=f"cuda:{r}"))
                fulls.append(torch.zeros((n_t,) + y0l.shape, dtype=torch.float64, device=f"cuda:{r}"))
                streams.append(torch.cuda.Stream(device=r))
        for r in range(G):
            torch.cuda.synchronize(r)
        arr = lambda xs: (C.c_void_p * G)(*xs)
        p = np.asarray(LOR)
        t_out = np.empty(n_t)
        rc = L.nnhip_ode_solve_batch_multi_gpu_f64_dev(C.byref(opt), nn.ode.integrator_id("tsit54"), nn.Rhs.LORENZ, p.ctypes.data_as(C.POINTER(C.c_double)), 3, G,
                                                       (C.c_int64 * G)(*counts), dim, layout, ts.ctypes.data_as(C.POINTER(C.c_double)), n_t,
                                                       t_out.ctypes.data_as(C.POINTER(C.c_double)), arr([y.data_ptr() for y in y0s]),
                                                       arr([o.data_ptr() for o in outs]), arr([q.data_ptr() for q in nys]), 0, arr([w.data_ptr() for w in wss]),
                                                       wsb, arr([f.data_ptr() for f in fulls]), arr([s.cuda_stream for s in streams]), None)
        assert rc == 0, L.nnhip_last_error()
        for r in range(G):
            torch.cuda.synchronize(r)
        with torch.cuda.device(0):
            tr, yr = nn.solveODE(nn.Rhs.lorenz(), torch.from_numpy(y0l).to("cuda:0"), ts, opt, integrator="tsit54", layout=layout)
        assert np.array_equal(t_out[:len(tr)], tr)
        for r in range(G):
            assert np.array_equal(fulls[r].cpu().numpy(), yr.cpu().numpy()), (G, r)
