"""Run by tests/test_host_logic_under_fake_hip.py in a subprocess with tests/cpp/fake_hip.cpp in front of the HIP runtime (LD_PRELOAD) and as "librccl.so.1":
the HOST logic of the one-call multi-GPU entries, the context-block sharding, the per-thread context bindings and the RCCL reassembly on a fake node of
FAKE_HIP_DEVICES devices.  Kernel launches do nothing there, so nothing is SOLVED: what is checked is that every entry returns, that every copy stays inside
the allocation it addresses (the stand-in aborts otherwise), that data the host moves ends up where it belongs, and that nothing is leaked.
Prints one line per scenario and "ALL OK" at the end; any failure is an exception (non-zero exit status)."""
import ctypes as C
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numericalnim_amd as nn  # noqa: E402
from test_ctx_block import MATVEC_SRC, ZCROSS_SRC  # noqa: E402

F = C.CDLL(os.environ["FAKE_HIP_LIB"])
F.fake_hip_launches.restype = C.c_long
F.fake_hip_live_device_allocations.restype = C.c_long
L = nn._lib.lib()
dp = C.POINTER(C.c_double)
NDEV = int(os.environ.get("FAKE_HIP_DEVICES", "1"))
assert L.nnhip_device_count() == NDEV, (L.nnhip_device_count(), NDEV)


def dev_alloc(device, nbytes):
    assert F.hipSetDevice(device) == 0
    p = C.c_void_p()
    assert F.hipMalloc(C.byref(p), C.c_size_t(max(nbytes, 8))) == 0
    return p.value


def dev_view(ptr, shape):
    n = int(np.prod(shape))
    return np.ctypeslib.as_array((C.c_double * max(n, 1)).from_address(ptr))[:n].reshape(shape)


def arr(xs):
    return (C.c_void_p * len(xs))(*xs)


def host_bind(f, shared, per, aux, stride):
    ptr = lambda a: None if a is None else a.ctypes.data_as(dp)  # noqa: E731
    rc = L.nnhip_ode_rhs_bind_ctx_f64(f.kind, ptr(shared), 0 if shared is None else shared.size, ptr(per), 0 if per is None else per.shape[0],
                                      ptr(aux), 0 if aux is None else aux.shape[0], stride, 0)
    assert rc == 0, nn._lib.last_error()


def scenario_sharded_context_host_entry():
    """nnhip_ode_solve_batch_multi_gpu_f64 with per-IVP 4 x 4 matrices (the sharding is the same for 16 x 16; these compile in a second): one shard per device, then more shards than devices (over-subscription)."""
    n, d = 1003, 4
    rng = np.random.default_rng(1)
    per = np.ascontiguousarray(rng.standard_normal((d * d, n)))
    g = rng.standard_normal(d)
    y0 = np.ascontiguousarray(rng.random((d, n)))
    f = nn.Rhs.custom(4, MATVEC_SRC, keys=("s",), tvalues={"g": 4, "A": 16}, per_ivp=("A",), name="matvec4_fake")
    host_bind(f, g, per, None, n)
    opt = nn.newODEoptions(absTol=1e-7, relTol=1e-7, dtMin=1e-8, dtMax=0.25)
    p = np.array([0.75])
    ts = np.array([0.0, 0.3, 1.0])
    out = np.full((3, d, n), -7.0)
    t_out = np.empty(3)
    ny = np.empty(n, dtype=np.int32)
    st = nn.ode.Stats()
    launches0 = F.fake_hip_launches()
    for shards in sorted({min(2, NDEV), NDEV, 8}):
        L.nnhip_tune_set(b"multi_gpu_oversubscribe", 1 if shards > NDEV else 0)
        try:
            rc = L.nnhip_ode_solve_batch_multi_gpu_f64(C.byref(opt), nn.ode.integrator_id("tsit54"), f.kind, p.ctypes.data_as(dp), 1, y0.ctypes.data, n, d, 0,
                                                       ts.ctypes.data_as(dp), 3, t_out.ctypes.data_as(dp), out.ctypes.data, ny.ctypes.data, 0, C.byref(st), shards)
            assert rc == 0, (shards, nn._lib.last_error())
        finally:
            L.nnhip_tune_set(b"multi_gpu_oversubscribe", 0)
        assert np.array_equal(t_out, ts)
    assert F.fake_hip_launches() > launches0
    # the single-device entry on the same binding
    rc = L.nnhip_ode_solve_batch_f64(C.byref(opt), nn.ode.integrator_id("tsit54"), f.kind, p.ctypes.data_as(dp), 1, y0.ctypes.data, n, d, 0, ts.ctypes.data_as(dp), 3,
                                     t_out.ctypes.data_as(dp), out.ctypes.data, None, None, None, 0, None, 0)
    assert rc == 0, nn._lib.last_error()
    # a batch larger than the bound block is refused, not copied out of bounds
    big = np.ascontiguousarray(rng.random((d, n + 5)))
    outb = np.empty((3, d, n + 5))
    rc = L.nnhip_ode_solve_batch_multi_gpu_f64(C.byref(opt), nn.ode.integrator_id("tsit54"), f.kind, p.ctypes.data_as(dp), 1, big.ctypes.data, n + 5, d, 0,
                                               ts.ctypes.data_as(dp), 3, t_out.ctypes.data_as(dp), outb.ctypes.data, None, 0, None, NDEV)
    assert rc != 0 or NDEV == 1
    L.nnhip_ode_rhs_release(f.kind)
    return "sharded context block through the host entry: %d launches" % (F.fake_hip_launches() - launches0)


def scenario_mutable_slots_round_trip():
    """Mutable slots go out to the shards by column range and come back in the caller's order (kernels do nothing here: what comes back must be what went out)."""
    n = 501
    f = nn.Rhs.custom(3, ZCROSS_SRC, keys=("sigma", "rho", "beta"), defaults=dict(sigma=10.0, rho=28.0, beta=8.0 / 3.0), n_aux=3, name="zcross_fake")
    aux0 = np.ascontiguousarray(np.random.default_rng(2).standard_normal((3, n)))
    host_bind(f, None, None, aux0, n)
    y0 = np.ones((3, n))
    opt = nn.newODEoptions(absTol=1e-6, relTol=1e-6, dtMin=1e-9, dtMax=0.1)
    p = np.array([10.0, 28.0, 8.0 / 3.0])
    ts = np.array([0.0, 3.0])
    out = np.empty((2, 3, n))
    t_out = np.empty(2)
    got = np.empty((3, n))
    for shards in sorted({NDEV, 5}):
        L.nnhip_tune_set(b"multi_gpu_oversubscribe", 1 if shards > NDEV else 0)
        try:
            rc = L.nnhip_ode_solve_batch_multi_gpu_f64(C.byref(opt), nn.ode.integrator_id("dopri54"), f.kind, p.ctypes.data_as(dp), 3, y0.ctypes.data, n, 3, 0,
                                                       ts.ctypes.data_as(dp), 2, t_out.ctypes.data_as(dp), out.ctypes.data, None, 0, None, shards)
            assert rc == 0, nn._lib.last_error()
        finally:
            L.nnhip_tune_set(b"multi_gpu_oversubscribe", 0)
        got[:] = -1.0
        assert L.nnhip_ode_rhs_read_aux_f64(f.kind, got.ctypes.data_as(dp)) == 0, nn._lib.last_error()
        assert np.array_equal(got, aux0), shards
    L.nnhip_ode_rhs_release(f.kind)
    return "mutable slots out to the shards and back"


def scenario_two_threads_bind_their_own_contexts():
    n, d = 300, 4
    f = nn.Rhs.custom(4, MATVEC_SRC, keys=("s",), tvalues={"g": 4, "A": 16}, per_ivp=("A",), name="matvec4_threads_fake")
    opt = nn.newODEoptions(absTol=1e-7, relTol=1e-7, dtMin=1e-8, dtMax=0.25)
    errors = []

    def worker(k):
        try:
            rng = np.random.default_rng(10 + k)
            per = np.ascontiguousarray(rng.standard_normal((d * d, n)))
            g = rng.standard_normal(d)
            y0 = np.ascontiguousarray(rng.random((d, n)))
            p = np.array([0.75])
            ts = np.array([0.0, 1.0])
            out = np.empty((2, d, n))
            t_out = np.empty(2)
            for _ in range(40):
                host_bind(f, g, per, None, n)
                rc = L.nnhip_ode_solve_batch_f64(C.byref(opt), nn.ode.integrator_id("tsit54"), f.kind, p.ctypes.data_as(dp), 1, y0.ctypes.data, n, d, 0, ts.ctypes.data_as(dp), 2,
                                                 t_out.ctypes.data_as(dp), out.ctypes.data, None, None, None, 0, None, k % NDEV)
                assert rc == 0, nn._lib.last_error()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    L.nnhip_ode_rhs_release(f.kind)
    return "4 threads x 40 bind + solve of one source with their own contexts"


def scenario_rccl_reassembly():
    """nnhip_allgather_states_f64_dev: every device ends up with the whole tensor, each shard where its index range belongs — equal shards (one all-gather, SoA: per
    plane), ragged shards and an empty one (grouped broadcasts)."""
    if NDEV < 2:
        return "RCCL reassembly: skipped (one device)"
    d = 3
    done = 0
    for counts in ([40] * NDEV, [57, 0, 31, 8, 13, 1, 2, 9][:NDEV], [5] + [0] * (NDEV - 1)):
        n = sum(counts)
        lo = np.concatenate([[0], np.cumsum(counts)])
        for layout in (0, 1):
            whole = np.random.default_rng(sum(counts) + layout).standard_normal((d, n) if layout == 0 else (n, d))
            shards, fulls = [], []
            for r in range(NDEV):
                part = whole[:, lo[r]:lo[r + 1]] if layout == 0 else whole[lo[r]:lo[r + 1]]
                ps = dev_alloc(r, part.size * 8)
                dev_view(ps, part.shape)[...] = part
                shards.append(ps)
                pf = dev_alloc(r, whole.size * 8)
                dev_view(pf, whole.shape)[...] = -3.0
                fulls.append(pf)
            rc = L.nnhip_allgather_states_f64_dev(NDEV, arr(shards), (C.c_int64 * NDEV)(*counts), d, layout, arr(fulls), None)
            assert rc == 0, L.nnhip_multigpu_last_error()
            for r in range(NDEV):
                assert np.array_equal(dev_view(fulls[r], whole.shape), whole), (counts, layout, r)
            for p in shards + fulls:
                assert F.hipFree(C.c_void_p(p)) == 0
            done += 1
    return "RCCL reassembly on %d devices: %d tensors placed" % (NDEV, done)


def scenario_device_resident_shards():
    """The two one-call entries whose shards live on the devices, with a sharded context block and the gather of the results: whatever the (idle) kernels left in
    the shard outputs must arrive, in place, in every device's full tensor."""
    if NDEV < 3:
        return "device-resident shards: skipped (needs 3 devices)"
    counts = [257, 0, 300] + [11] * (NDEV - 3)
    G, n, d = NDEV, sum(counts), 4
    lo = np.concatenate([[0], np.cumsum(counts)])
    rng = np.random.default_rng(4)
    per = np.ascontiguousarray(rng.standard_normal((d * d, n)))
    f = nn.Rhs.custom(4, MATVEC_SRC, keys=("s",), tvalues={"g": 4, "A": 16}, per_ivp=("A",), name="matvec4_dev_fake")
    host_bind(f, rng.standard_normal(d), per, None, n)
    opt = nn.newODEoptions(absTol=1e-7, relTol=1e-7, dtMin=1e-8, dtMax=0.25)
    p = np.array([0.75])
    ts = np.array([0.0, 0.4, 1.0])
    t_out = np.empty(3)
    wsb = int(L.nnhip_ode_solve_workspace_bytes(3))
    y0s, outs, wss, fulls, streams, patterns = [], [], [], [], [], []
    for r in range(G):
        y0s.append(dev_alloc(r, d * counts[r] * 8))
        outs.append(dev_alloc(r, 3 * d * counts[r] * 8))
        pat = rng.standard_normal((3, d, counts[r]))
        dev_view(outs[r], pat.shape)[...] = pat
        patterns.append(pat)
        wss.append(dev_alloc(r, wsb))
        fulls.append(dev_alloc(r, 3 * d * n * 8))
        s = C.c_void_p()
        assert F.hipSetDevice(r) == 0 and F.hipStreamCreateWithFlags(C.byref(s), 1) == 0
        streams.append(s.value)
    rc = L.nnhip_ode_solve_batch_multi_gpu_f64_dev(C.byref(opt), nn.ode.integrator_id("dopri54"), f.kind, p.ctypes.data_as(dp), 1, G, (C.c_int64 * G)(*counts), d, 0,
                                                   ts.ctypes.data_as(dp), 3, t_out.ctypes.data_as(dp), arr(y0s), arr(outs), None, 0, arr(wss), wsb, arr(fulls), arr(streams), None)
    assert rc == 0, nn._lib.last_error()
    want = np.concatenate(patterns, axis=2)
    for r in range(G):
        assert np.array_equal(dev_view(fulls[r], (3, d, n)), want), r
    # the step-streaming loop over the same shards (no gather)
    optf = nn.newODEoptions(dt=2.0 ** -7)
    ys = [dev_alloc(r, d * counts[r] * 8) for r in range(G)]
    scr = [dev_alloc(r, d * counts[r] * 8) for r in range(G)]
    fin = (C.c_void_p * G)()
    nst = C.c_int64(0)
    rc = L.nnhip_ode_fixed_stream_multi_gpu_f64_dev(C.byref(optf), nn.ode.integrator_id("rk4"), f.kind, p.ctypes.data_as(dp), 1, G, (C.c_int64 * G)(*counts), d, 0, 0.0, 0.5,
                                                    arr(ys), arr(scr), None, arr(streams), None, C.byref(nst), fin)
    assert rc == 0 and nst.value == 64, (rc, nst.value, nn._lib.last_error())
    for ptr in y0s + outs + wss + fulls + ys + scr:
        assert F.hipFree(C.c_void_p(ptr)) == 0
    L.nnhip_ode_rhs_release(f.kind)
    return "device-resident shards with a sharded context block, gathered on %d devices" % G


def scenario_streaming_drivers():
    """The host side of the step-streaming drivers: no kernel runs, so no IVP ever reports work left — the polling loops must end by themselves with exactly the
    launches their schedule issues before a poll can stop them — the figures tests/cpp/test_poll_schedule.cpp replays from adv_poll_schedule.hpp for a batch
    that needs 0 iterations (BASELINE's C3 / C4 options: 100 unpolled + the group of 2 enqueued ahead = 102, where a batch that needs 102 iterations takes 104; a caller's
    check_every = 8: 8 + 8), max_launches cuts where it says, the fixed-step loop launches once per step; every table / state copy stays inside its allocation."""
    n = 5000
    lor = nn.Rhs.lorenz()
    p = np.array([10.0, 28.0, 8.0 / 3.0])
    opt = nn.newODEoptions()
    integ = nn.ode.integrator_id("dopri54")
    y = dev_alloc(0, 3 * n * 8)
    wsb = int(L.nnhip_ode_adaptive_stream_workspace_bytes(n, 3))
    ws = dev_alloc(0, wsb)
    nl = C.c_int64(0)
    s = C.c_void_p()
    assert F.hipStreamCreateWithFlags(C.byref(s), 1) == 0
    seen = []
    for check_every, max_launches, want in ((0, 0, 102), (8, 0, 16), (0, 50, 50), (5, 12, 10), (-1, 0, 16)):
        l0 = F.fake_hip_launches()
        L.nnhip_tune_set(b"adv_auto_poll", 0 if check_every < 0 else 1)   # (-1: check_every <= 0 WITHOUT the opt-in knob = uniform groups of 8, the default)
        rc = L.nnhip_ode_adaptive_stream_f64_dev(C.byref(opt), integ, lor.kind, p.ctypes.data_as(dp), 3, n, 3, 0, 0.0, 1.0, y, ws, wsb, check_every, max_launches, C.byref(nl), s)
        assert rc in (0, nn._lib.NNHIP_TRUNCATED), nn._lib.last_error()
        assert nl.value == want, (check_every, max_launches, nl.value, want)
        assert F.fake_hip_launches() - l0 == want + 1      # + the kernel that fills (t, dt)
        seen.append(nl.value)
    # the dense driver: requested times on both sides of tStart
    ts = np.concatenate([np.linspace(-0.3, -0.05, 4), np.linspace(0.0, 1.0, 7)])
    wsd = int(L.nnhip_ode_adaptive_stream_dense_workspace_bytes(n, 3, len(ts)))
    wd = dev_alloc(0, wsd)
    y0 = dev_alloc(0, 3 * n * 8)
    yout = dev_alloc(0, len(ts) * 3 * n * 8)
    ny = dev_alloc(0, n * 4)
    t_out = np.empty(len(ts))
    L.nnhip_tune_set(b"adv_auto_poll", 1)
    rc = L.nnhip_ode_adaptive_stream_dense_f64_dev(C.byref(opt), integ, lor.kind, p.ctypes.data_as(dp), 3, y0, n, 3, 0, ts.ctypes.data_as(dp), len(ts), t_out.ctypes.data_as(dp),
                                                   yout, ny, wd, wsd, 0, 0, C.byref(nl), s)
    assert rc == 0, nn._lib.last_error()
    L.nnhip_tune_set(b"adv_auto_poll", 0)
    assert np.array_equal(t_out, np.sort(ts)) and nl.value == (100 + 2) + (30 + 2), nl.value   # forward [0, 1] and backward [0, 0.3] at dtMax = 0.01
    # the fixed-step loop: one launch per RK4 step
    optf = nn.newODEoptions(dt=2.0 ** -10)
    yf = dev_alloc(0, n * 8)
    sc = dev_alloc(0, n * 8)
    ns = C.c_int64(0)
    fin = C.c_void_p()
    l0 = F.fake_hip_launches()
    rc = L.nnhip_ode_fixed_stream_f64_dev(C.byref(optf), nn.ode.integrator_id("rk4"), nn.Rhs.neg_y().kind, None, 0, n, 1, 0, 0.0, 1000 * 2.0 ** -10, yf, sc, C.byref(ns), C.byref(fin), s)
    assert rc == 0 and ns.value == 1000 and F.fake_hip_launches() - l0 == 1000 and fin.value == yf, (rc, ns.value, nn._lib.last_error())
    # the order of integration over a device key array
    keys = dev_alloc(0, (1 << 16) * 8)
    order = dev_alloc(0, (1 << 16) * 4)
    assert L.nnhip_ode_bin_order_f64_dev(keys, 1 << 16, order, s) == 0, nn._lib.last_error()
    for ptr in (y, ws, wd, y0, yout, ny, yf, sc, keys, order):
        assert F.hipFree(C.c_void_p(ptr)) == 0
    return "streaming drivers: launches %s, dense both directions 134, fixed-step 1000" % seen


def scenario_consumers():
    """The host side of the section-8-f4 consumers (this round's quad_plan.hpp / consumer_kernels.hpp re-factoring runs here through the C ABI): for every
    function-form case of the reference's text the number of result rows the entry reports equals the text's, the table uploads stay in bounds; the discrete forms,
    the spline evaluation with every ExtrapolateKind (ExtrapolateKind.Error is refused) and the slope estimate run through their host-pointer entries."""
    import json
    V = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_text_quad_vectors.json")))
    fh = lambda xs: np.array([float.fromhex(x) for x in xs])  # noqa: E731
    src = "for (int c = 0; c < {d}; ++c) dy[c] = ((p[0] * t + p[1]) * t) * (1.0 + (double)c) + p[2];"
    fs = {d: nn.Rhs.custom(d, src.format(d=d), keys=("a", "b", "c"), name="poly%d_fake" % d) for d in (1, 3)}
    n_items = 3
    checked = 0
    for c in V["cumquad_fn"]:
        d = max(c["dim"], 1)
        X = fh(c["X"])
        p = np.ascontiguousarray(fh(c["params"])[:3])
        out = np.empty((len(X), d, n_items))
        rows = C.c_int(-1)
        fn = L.nnhip_cumtrapz_fn_batch_f64 if c["rule"] == "trapz" else L.nnhip_cumsimpson_fn_batch_f64
        rc = fn(fs[d].kind, p.ctypes.data_as(dp), 3, None, 0, n_items, d, 0, X.ctypes.data_as(dp), len(X), float.fromhex(c["dx"]), out.ctypes.data_as(dp), C.byref(rows), 0)
        assert rc == 0, (c["name"], nn._lib.last_error())
        assert rows.value == c["rows"], (c["name"], rows.value, c["rows"])
        checked += 1
    for c in V["cumquad_discrete"]:
        if not c["strictly_ascending"]:
            continue
        X = fh(c["X"])
        Y = np.ascontiguousarray(np.tile(np.stack([fh(y) for y in c["Y"]], axis=1), (1, 40)))
        out = np.empty_like(Y)
        assert L.nnhip_cumtrapz_batch_f64(X.ctypes.data_as(dp), len(X), Y.ctypes.data_as(dp), Y.shape[1], out.ctypes.data_as(dp), 0) == 0, nn._lib.last_error()
        rc = L.nnhip_cumsimpson_batch_f64(X.ctypes.data_as(dp), len(X), Y.ctypes.data_as(dp), Y.shape[1], out.ctypes.data_as(dp), 0)
        assert (rc != 0) == isinstance(c["cumsimpson"], dict), (c["name"], rc)      # fewer than 3 points: the reference's ValueError
    for c in V["hermite"]:
        X, Yv, dYv, xq = fh(c["X"]), fh(c["Y"]), fh(c["dY"]), fh(c["xq"])
        Y = np.ascontiguousarray(np.tile(Yv[:, None], (1, 70)))
        dY = np.ascontiguousarray(np.tile(dYv[:, None], (1, 70)))
        out = np.empty((len(xq), 70))
        for extrap in range(4):
            for deriv in (0, 1):
                rc = L.nnhip_hermite_spline_eval_batch_f64(X.ctypes.data_as(dp), len(X), Y.ctypes.data_as(dp), dY.ctypes.data_as(dp), 70, xq.ctypes.data_as(dp), len(xq), deriv, extrap,
                                                           0.5, out.ctypes.data_as(dp), 0)
                assert rc == 0, nn._lib.last_error()
        outside = np.array([X.min() - 1.0])   # (knots may arrive unsorted: two of the cases)
        assert L.nnhip_hermite_spline_eval_batch_f64(X.ctypes.data_as(dp), len(X), Y.ctypes.data_as(dp), dY.ctypes.data_as(dp), 70, outside.ctypes.data_as(dp), 1, 0, 4, 0.0,
                                                     out.ctypes.data_as(dp), 0) != 0
    for f in fs.values():
        L.nnhip_ode_rhs_release(f.kind)
    return "consumers: %d function-form cases with the reference text's row counts, discrete forms, spline evaluation" % checked


def main():
    only = sys.argv[1:]
    for sc in (scenario_sharded_context_host_entry, scenario_mutable_slots_round_trip, scenario_two_threads_bind_their_own_contexts, scenario_rccl_reassembly,
               scenario_device_resident_shards, scenario_streaming_drivers, scenario_consumers):
        if only and sc.__name__.replace("scenario_", "") not in only:
            continue
        print(sc.__name__, "->", sc(), flush=True)
    L.nnhip_release()
    # (threading.Thread.join returns when the Python thread state is gone — the OS thread may still be running its thread_local destructors, which is where a worker's
    # own context bindings are freed: give them a moment before counting)
    import time
    for _ in range(100):
        if F.fake_hip_live_device_allocations() == 0:
            break
        time.sleep(0.05)
    print("live device allocations after nnhip_release():", F.fake_hip_live_device_allocations(), flush=True)
    F.fake_hip_dump_live()
    print("ALL OK", flush=True)


if __name__ == "__main__":
    main()
