"""Run by tests/test_end_to_end_on_isa_node.py in a subprocess on the ISA-backed fake node (tests/isa_backed_node.py: fake HIP runtime + RCCL, every launch executed
by the gfx950 interpreter from the library's own code objects): the library's entries END TO END — real host logic, real compiled device code — with their RESULTS
compared against the oracle and the reference's text, on the code paths that need several GPUs or were written in a round without one.  Small batches: the
interpreter executes ~30 000 wave-instructions per second.  Prints one line per scenario and "ALL OK"."""
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import isa_backed_node  # noqa: E402

NODE = isa_backed_node.attach()

import numericalnim_amd as nn  # noqa: E402
from golden_util import fh, load_cases  # noqa: E402
from oracle import oracle as O  # noqa: E402
from test_ctx_block import MATVEC_SRC, ZCROSS_SRC  # noqa: E402

O.build()
O.lib()
F = NODE.F
L = nn._lib.lib()
dp = C.POINTER(C.c_double)
NDEV = int(os.environ.get("FAKE_HIP_DEVICES", "1"))
FULL = bool(os.environ.get("NNHIP_ISA_NODE_FULL"))
KEYS = {1: ("a",), 2: ("sigma", "rho", "beta"), 3: ("c",), 4: ("a", "b"), 5: ("mu",)}


def check_node():
    assert not NODE.errors, NODE.errors[0][:3000]


def dev_alloc(device, nbytes):
    assert F.hipSetDevice(device) == 0
    p = C.c_void_p()
    assert F.hipMalloc(C.byref(p), C.c_size_t(max(nbytes, 8))) == 0
    return p.value


def dev_view(ptr, shape, dtype=np.float64):
    n = int(np.prod(shape))
    ct = {np.float64: C.c_double, np.int32: C.c_int32, np.uint32: C.c_uint32}[dtype]
    return np.ctypeslib.as_array((ct * max(n, 1)).from_address(ptr))[:n].reshape(shape)


def arr(xs):
    return (C.c_void_p * len(xs))(*xs)


def host_bind(f, shared, per, aux, stride):
    ptr = lambda a: None if a is None else a.ctypes.data_as(dp)  # noqa: E731
    rc = L.nnhip_ode_rhs_bind_ctx_f64(f.kind, ptr(shared), 0 if shared is None else shared.size, ptr(per), 0 if per is None else per.shape[0],
                                      ptr(aux), 0 if aux is None else aux.shape[0], stride, 0)
    assert rc == 0, nn._lib.last_error()


def scenario_golden_fixtures():
    """tests/golden/ode_golden.json through nn.solveODE (host arrays -> nnhip_ode_solve_batch_f64): output times, rows, row counts == the reference's TEXT, accepted /
    rejected counts == the oracle's — tests/test_gpu_golden.py's assertions, without a GPU.  Default: every integrator on Lorenz + the t-dependent, rejecting, dtMin-escape, shifted-tStart, dropped-rows, both-sided dense and lanes-per-system fixtures; NNHIP_ISA_NODE_FULL=1: all 101 fixtures x layouts (160 solves, ~11 min)."""
    ref = {c["name"]: c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "reference_text_vectors.json")))["cases"]}
    cases = load_cases()
    if not FULL:
        import re
        want = r"^(lorenz_default_(?!rk21).*|affine_t_dopri54|rejecting_vdp_dopri54|dtmin_escape_tsit54|tstart_shift_.*|quirk_.*|ring4_tsit54|ring16_(tsit54|rk4))$"
        cases = [c for c in cases if re.match(want, c["name"])]
    done = 0
    for case in cases:
        n = len(case["y0"])
        y0 = np.stack([fh(y) for y in case["y0"]])
        for layout in ((0,) if case["dim"] == 0 else ((0, 1) if FULL or case["name"].startswith(("ring", "quirk", "lorenz_default_tsit54")) else (done % 2,))):
            y0a = y0[:, 0].copy() if case["dim"] == 0 else (np.ascontiguousarray(y0.T) if layout == 0 else y0.copy())
            f = nn.Rhs(case["rhs_kind"], KEYS.get(case["rhs_kind"], ()), dict(zip(KEYS.get(case["rhs_kind"], ()), fh(case["params"]))))
            t1 = time.time()
            t, y, cnt = nn.solveODE(f, y0a, fh(case["tspan"]), nn.newODEoptions(**case["options"]), integrator=case["integrator"], layout=layout, return_counts=True)
            check_node()
            rt = ref[case["name"]]
            assert [float(v).hex() for v in t] == rt["t"], case["name"]
            y, ny, steps, rej = np.asarray(y), np.asarray(cnt["ny"]), np.asarray(cnt["steps"]), np.asarray(cnt["rejected"])
            for i, exp in enumerate(case["ivps"]):
                g = y[:, i] if case["dim"] == 0 else (y[:, :, i] if layout == 0 else y[:, i, :])
                gi = g.reshape(len(t), -1)
                assert ny[i] == rt["ivps"][i]["n_y"] == exp["n_y"], (case["name"], i)
                assert [float(v).hex() for v in gi[:exp["n_y"]].ravel()] == rt["ivps"][i]["y"], (case["name"], layout, i, "differs from the reference's text")
                assert np.isnan(gi[exp["n_y"]:]).all()
                assert (int(steps[i]), int(rej[i])) == (exp["steps"], exp["rejected"]), (case["name"], i)
            done += 1
            if os.environ.get("NNHIP_ISA_NODE_VERBOSE"):
                print("   %-32s layout %d  %.1f s" % (case["name"], layout, time.time() - t1), flush=True)
    return "%d golden solves == the reference's text" % done


def scenario_sharded_context_results():
    """VERDICT r04 #7 with results: every system its own 4 x 4 matrix (per-IVP ctx.tValues) + a shared forcing vector, the batch cut into 3 shards on 3 devices and
    into 8 on 3 (over-subscription): device r's kernels read columns [lo_r, hi_r) of the context block — rows, row counts, step totals == the oracle's N closures."""
    n, d = 61, 4
    rng = np.random.default_rng(23)
    A = rng.standard_normal((n, d, d)) * 0.35 - 0.6 * np.eye(d)[None]
    g = rng.standard_normal(d) * 0.2
    s = 0.75
    y0 = np.ascontiguousarray(0.5 + rng.random((d, n)))
    per = np.ascontiguousarray(A.reshape(n, d * d).T)
    f = nn.Rhs.custom(4, MATVEC_SRC, keys=("s",), tvalues={"g": 4, "A": 16}, per_ivp=("A",), name="matvec4_isa")
    kw = dict(absTol=1e-7, relTol=1e-7, dtMin=1e-8, dtMax=0.25)
    ts = np.array([0.0, 0.3, 1.0])
    ref = O.solve_ode_batch_ctx(O.RHS_MATVEC, [s] + list(g), per, None, y0, n, d, ts, O.new_options(**kw), "tsit54")
    host_bind(f, np.ascontiguousarray(g), per, None, n)
    opt = nn.newODEoptions(**kw)
    p = np.array([s])
    for shards in sorted({NDEV, 8}):
        out = np.full((3, d, n), -7.0)
        t_out = np.empty(3)
        ny = np.empty(n, dtype=np.int32)
        st = nn.ode.Stats()
        L.nnhip_tune_set(b"multi_gpu_oversubscribe", 1 if shards > NDEV else 0)
        try:
            rc = L.nnhip_ode_solve_batch_multi_gpu_f64(C.byref(opt), nn.ode.integrator_id("tsit54"), f.kind, p.ctypes.data_as(dp), 1, y0.ctypes.data, n, d, 0,
                                                       ts.ctypes.data_as(dp), 3, t_out.ctypes.data_as(dp), out.ctypes.data, ny.ctypes.data, 0, C.byref(st), shards)
        finally:
            L.nnhip_tune_set(b"multi_gpu_oversubscribe", 0)
        check_node()
        assert rc == 0, nn._lib.last_error()
        assert np.array_equal(t_out, ref["t"]) and np.array_equal(out, ref["y"]) and np.array_equal(ny, ref["ny"]), shards
        assert st.steps_total == int(ref["steps"].sum())
    L.nnhip_ode_rhs_release(f.kind)
    return "per-IVP matrices travel with their shard: == the oracle's closures at %d and 8 shards" % NDEV


def scenario_mutable_slots_results():
    """A closure that mutates its ctx, sharded: every shard's device updates its own columns of the mutable slots; read back in the caller's order == the oracle's
    closures' final environments (crossings of z = 25 counted, last z, number of calls)."""
    n = 23
    f = nn.Rhs.custom(3, ZCROSS_SRC, keys=("sigma", "rho", "beta"), defaults=dict(sigma=10.0, rho=28.0, beta=8.0 / 3.0), n_aux=3, name="zcross_isa")
    y0 = np.ascontiguousarray(np.stack([1.0 + np.arange(n) * 1e-3, np.ones(n), np.ones(n) * 20.0]))
    kw = dict(absTol=1e-5, relTol=1e-5, dtMin=1e-9, dtMax=0.1)
    ts = np.array([0.0, 1.5])
    aux0 = np.zeros((3, n))
    ref = O.solve_ode_batch_ctx(O.RHS_LORENZ_ZCROSS, [10.0, 28.0, 8.0 / 3.0], None, aux0.copy(), y0, n, 3, ts, O.new_options(**kw), "dopri54")
    assert ref["aux"][0].max() >= 1.0
    host_bind(f, None, None, aux0, n)
    opt = nn.newODEoptions(**kw)
    p = np.array([10.0, 28.0, 8.0 / 3.0])
    out = np.empty((2, 3, n))
    t_out = np.empty(2)
    rc = L.nnhip_ode_solve_batch_multi_gpu_f64(C.byref(opt), nn.ode.integrator_id("dopri54"), f.kind, p.ctypes.data_as(dp), 3, y0.ctypes.data, n, 3, 0,
                                               ts.ctypes.data_as(dp), 2, t_out.ctypes.data_as(dp), out.ctypes.data, None, 0, None, NDEV)
    check_node()
    assert rc == 0, nn._lib.last_error()
    got = np.empty((3, n))
    assert L.nnhip_ode_rhs_read_aux_f64(f.kind, got.ctypes.data_as(dp)) == 0
    assert np.array_equal(out, ref["y"]) and np.array_equal(got, ref["aux"])
    L.nnhip_ode_rhs_release(f.kind)
    return "mutable slots come back from their shards: == the oracle's environments (%d crossings counted)" % int(ref["aux"][0].sum())


def scenario_two_threads_results():
    """Two host threads solve the SAME compiled source with DIFFERENT contexts at the same time: each gets its own context's result every time."""
    n, d = 17, 4
    f = nn.Rhs.custom(4, MATVEC_SRC, keys=("s",), tvalues={"g": 4, "A": 16}, per_ivp=("A",), name="matvec4_threads_isa")
    kw = dict(absTol=1e-7, relTol=1e-7, dtMin=1e-8, dtMax=0.25)
    rng = np.random.default_rng(5)
    y0 = np.ascontiguousarray(0.5 + rng.random((d, n)))
    jobs = []
    for k in range(2):
        A = rng.standard_normal((n, d, d)) * 0.35 - 0.6 * np.eye(d)[None]
        g = rng.standard_normal(d) * 0.2
        per = np.ascontiguousarray(A.reshape(n, d * d).T)
        ref = O.solve_ode_batch_ctx(O.RHS_MATVEC, [0.75] + list(g), per, None, y0, n, d, [0.0, 1.0], O.new_options(**kw), "tsit54")["y"][-1]
        jobs.append((np.ascontiguousarray(g), per, ref))
    bad, errors = [0, 0], []
    opt = nn.newODEoptions(**kw)

    def worker(k):
        try:
            g, per, ref = jobs[k]
            p = np.array([0.75])
            ts = np.array([0.0, 1.0])
            out = np.empty((2, d, n))
            t_out = np.empty(2)
            for _ in range(6):
                host_bind(f, g, per, None, n)
                rc = L.nnhip_ode_solve_batch_f64(C.byref(opt), nn.ode.integrator_id("tsit54"), f.kind, p.ctypes.data_as(dp), 1, y0.ctypes.data, n, d, 0, ts.ctypes.data_as(dp), 2,
                                                 t_out.ctypes.data_as(dp), out.ctypes.data, None, None, None, 0, None, k % NDEV)
                assert rc == 0, nn._lib.last_error()
                if not np.array_equal(out[1], ref):
                    bad[k] += 1
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    check_node()
    assert not errors, errors
    assert bad == [0, 0], bad
    L.nnhip_ode_rhs_release(f.kind)
    return "two threads, one source, their own contexts: 12 solves, each its own result"


def scenario_streaming_results():
    """The step-streaming drivers with device-resident state: the adaptive loop over this round's lean kernels with the automatic polling schedule (flags stored by the
    kernels into page-locked host memory, read by the host between groups) — BASELINE's C3 / C4 options: 102 iterations, 104 launches, the fused solve's bits; the
    dense driver on both sides of tStart == the fused solve; the fixed-step loop."""
    lines = []
    opt = nn.newODEoptions()
    s = C.c_void_p()
    assert F.hipStreamCreateWithFlags(C.byref(s), 1) == 0
    for name, f, y0, layout, d, integ, kind, par in (
            ("C3", nn.Rhs.lorenz(), np.stack([1.0 + (np.arange(70) % 1024) * 2.0 ** -20, np.ones(70), np.ones(70)]), 0, 3, "dopri54", O.RHS_LORENZ, [10.0, 28.0, 8.0 / 3.0]),
            ("C4", nn.Rhs.ring(0.1), 1.0 + np.arange(16)[None, :] / 16 + ((np.arange(20) % 1024) * 2.0 ** -20)[:, None], 1, 16, "tsit54", O.RHS_RING, [0.1])):
        y0 = np.ascontiguousarray(y0)
        n = y0.shape[1 - layout] if d > 1 else y0.size
        yd = dev_alloc(0, y0.size * 8)
        dev_view(yd, y0.shape)[...] = y0
        wsb = int(L.nnhip_ode_adaptive_stream_workspace_bytes(n, d))
        ws = dev_alloc(0, wsb)
        nl = C.c_int64(0)
        p = np.array(par)
        L.nnhip_tune_set(b"adv_lean", 1)        # the opt-in pair of round 5: the lean kernels, the library's own polling schedule
        L.nnhip_tune_set(b"adv_auto_poll", 1)
        rc = L.nnhip_ode_adaptive_stream_f64_dev(C.byref(opt), nn.ode.integrator_id(integ), f.kind, p.ctypes.data_as(dp), len(par), n, d, layout, 0.0, 1.0, yd, ws, wsb, 0, 0,
                                                 C.byref(nl), s)
        check_node()
        assert rc == 0, nn._lib.last_error()
        ref = O.solve_ode_batch(kind, par, y0, n, d, [0.0, 1.0], O.new_options(), integ, layout=layout)
        assert np.array_equal(dev_view(yd, y0.shape), ref["y"][-1]), name
        assert int(ref["steps"].max()) == 102 and nl.value == 104, (name, nl.value)
        lines.append("%s %d launches" % (name, nl.value))
        # the general kernels through the same driver (knob adv_lean = 0): the same bits, the same launches
        dev_view(yd, y0.shape)[...] = y0
        L.nnhip_tune_set(b"adv_lean", 0)
        try:
            rc = L.nnhip_ode_adaptive_stream_f64_dev(C.byref(opt), nn.ode.integrator_id(integ), f.kind, p.ctypes.data_as(dp), len(par), n, d, layout, 0.0, 1.0, yd, ws, wsb, 0, 0,
                                                     C.byref(nl), s)
        finally:
            L.nnhip_tune_set(b"adv_auto_poll", 0)
        check_node()
        assert rc == 0 and nl.value == 104 and np.array_equal(dev_view(yd, y0.shape), ref["y"][-1]), (name, "general kernel")
        F.hipFree(C.c_void_p(yd))
        F.hipFree(C.c_void_p(ws))
    # dense driver, both directions
    n = 9
    y0 = np.ascontiguousarray(np.stack([1.0 + np.arange(n) * 2.0 ** -10, np.ones(n), np.ones(n)]))
    ts = np.concatenate([np.linspace(-0.06, -0.02, 3), np.linspace(0.0, 0.2, 4)])
    par = [10.0, 28.0, 8.0 / 3.0]
    p = np.array(par)
    ref = O.solve_ode_batch(O.RHS_LORENZ, par, y0, n, 3, ts, O.new_options(), "dopri54")
    y0d = dev_alloc(0, y0.size * 8)
    dev_view(y0d, y0.shape)[...] = y0
    yout = dev_alloc(0, len(ts) * 3 * n * 8)
    nyd = dev_alloc(0, n * 4)
    wsd = int(L.nnhip_ode_adaptive_stream_dense_workspace_bytes(n, 3, len(ts)))
    wd = dev_alloc(0, wsd)
    t_out = np.empty(len(ts))
    nl = C.c_int64(0)
    rc = L.nnhip_ode_adaptive_stream_dense_f64_dev(C.byref(opt), nn.ode.integrator_id("dopri54"), nn.Rhs.lorenz().kind, p.ctypes.data_as(dp), 3, y0d, n, 3, 0, ts.ctypes.data_as(dp),
                                                   len(ts), t_out.ctypes.data_as(dp), yout, nyd, wd, wsd, 0, 0, C.byref(nl), s)
    check_node()
    assert rc == 0, nn._lib.last_error()
    assert np.array_equal(t_out, ref["t"]) and np.array_equal(dev_view(yout, (len(ts), 3, n)), ref["y"], equal_nan=True)
    assert np.array_equal(dev_view(nyd, (n,), np.int32), ref["ny"])
    lines.append("dense both directions %d launches" % nl.value)
    for ptr in (y0d, yout, nyd, wd):
        F.hipFree(C.c_void_p(ptr))
    return "streaming drivers with results: " + ", ".join(lines)


def scenario_c5_shape_with_results():
    """Config C5's shape on the fake node: shards of a scalar RK4 batch resident on every device, the step-streaming solve per device (worker thread + stream each),
    the RCCL reassembly — every device ends up with the whole final state == the oracle's."""
    if NDEV < 2:
        return "C5 shape: skipped (one device)"
    counts = [700, 650, 0, 513][:NDEV] + [40] * max(0, NDEV - 4)
    G, n = NDEV, sum(counts)
    lo = np.concatenate([[0], np.cumsum(counts)])
    y0 = 1.0 + (np.arange(n) % (1 << 20)) * 2.0 ** -20
    steps, dt = 16, 2.0 ** -10
    opt = nn.newODEoptions(dt=dt)
    ref = O.solve_ode_batch(O.RHS_NEG_Y, [], y0, n, 0, [0.0, steps * dt], O.new_options(dt=dt), "rk4")["y"][-1, 0]
    ys, scr, fulls = [], [], []
    for r in range(G):
        ys.append(dev_alloc(r, counts[r] * 8))
        dev_view(ys[r], (counts[r],))[...] = y0[lo[r]:lo[r + 1]]
        scr.append(dev_alloc(r, counts[r] * 8))
        fulls.append(dev_alloc(r, n * 8))
    fin = (C.c_void_p * G)()
    nst = C.c_int64(0)
    L.nnhip_tune_set(b"stream_graph", 0)  # (eager launches: the node executes launches, it does not replay captured graphs)
    try:
        rc = L.nnhip_ode_fixed_stream_multi_gpu_f64_dev(C.byref(opt), nn.ode.integrator_id("rk4"), nn.Rhs.neg_y().kind, None, 0, G, (C.c_int64 * G)(*counts), 1, 0, 0.0, steps * dt,
                                                        arr(ys), arr(scr), arr(fulls), arr([None] * G), None, C.byref(nst), fin)
    finally:
        L.nnhip_tune_set(b"stream_graph", -1)
    check_node()
    assert rc == 0 and nst.value == steps, (rc, nst.value, nn._lib.last_error(), L.nnhip_multigpu_last_error())
    for r in range(G):
        assert np.array_equal(dev_view(fulls[r], (n,)), ref), r
    for ptr in ys + scr + fulls:
        F.hipFree(C.c_void_p(ptr))
    return "C5 shape on %d devices: every device holds the oracle's %d final states" % (G, n)


def scenario_device_resident_shards_results():
    """The two one-call entries whose shards live on the devices, on a batch with per-IVP matrices: shard r reads columns [lo_r, hi_r) of the host-bound context
    block, the fused results are reassembled on every device (RCCL), then the fixed-step streaming loop over the same shards — == the oracle's closures."""
    if NDEV < 3:
        return "device-resident shards: skipped (needs 3 devices)"
    counts = [21, 0, 17] + [3] * (NDEV - 3)
    G, n, d = NDEV, sum(counts), 4
    lo = np.concatenate([[0], np.cumsum(counts)])
    rng = np.random.default_rng(29)
    A = rng.standard_normal((n, d, d)) * 0.35 - 0.6 * np.eye(d)[None]
    g = rng.standard_normal(d) * 0.2
    s = 0.75
    y0 = np.ascontiguousarray(0.5 + rng.random((d, n)))
    per = np.ascontiguousarray(A.reshape(n, d * d).T)
    f = nn.Rhs.custom(4, MATVEC_SRC, keys=("s",), tvalues={"g": 4, "A": 16}, per_ivp=("A",), name="matvec4_dev_isa")
    host_bind(f, np.ascontiguousarray(g), per, None, n)
    kw = dict(absTol=1e-7, relTol=1e-7, dtMin=1e-8, dtMax=0.25)
    opt = nn.newODEoptions(**kw)
    p = np.array([s])
    ts = np.array([0.0, 0.4, 1.0])
    ref = O.solve_ode_batch_ctx(O.RHS_MATVEC, [s] + list(g), per, None, y0, n, d, ts, O.new_options(**kw), "dopri54")
    wsb = int(L.nnhip_ode_solve_workspace_bytes(3))
    y0s, outs, wss, fulls = [], [], [], []
    for r in range(G):
        y0s.append(dev_alloc(r, d * counts[r] * 8))
        dev_view(y0s[r], (d, counts[r]))[...] = y0[:, lo[r]:lo[r + 1]]
        outs.append(dev_alloc(r, 3 * d * counts[r] * 8))
        wss.append(dev_alloc(r, wsb))
        fulls.append(dev_alloc(r, 3 * d * n * 8))
    t_out = np.empty(3)
    rc = L.nnhip_ode_solve_batch_multi_gpu_f64_dev(C.byref(opt), nn.ode.integrator_id("dopri54"), f.kind, p.ctypes.data_as(dp), 1, G, (C.c_int64 * G)(*counts), d, 0,
                                                   ts.ctypes.data_as(dp), 3, t_out.ctypes.data_as(dp), arr(y0s), arr(outs), None, 0, arr(wss), wsb, arr(fulls), arr([None] * G), None)
    check_node()
    assert rc == 0, (nn._lib.last_error(), L.nnhip_multigpu_last_error())
    for r in range(G):
        assert np.array_equal(dev_view(outs[r], (3, d, counts[r])), ref["y"][:, :, lo[r]:lo[r + 1]]), r
        assert np.array_equal(dev_view(fulls[r], (3, d, n)), ref["y"]), ("gathered", r)
    optf = nn.newODEoptions(dt=2.0 ** -5)
    reff = O.solve_ode_batch_ctx(O.RHS_MATVEC, [s] + list(g), per, None, y0, n, d, [0.0, 0.5], O.new_options(dt=2.0 ** -5), "rk4")
    ys = [dev_alloc(r, d * counts[r] * 8) for r in range(G)]
    scr = [dev_alloc(r, d * counts[r] * 8) for r in range(G)]
    for r in range(G):
        dev_view(ys[r], (d, counts[r]))[...] = y0[:, lo[r]:lo[r + 1]]
    fin = (C.c_void_p * G)()
    nst = C.c_int64(0)
    L.nnhip_tune_set(b"stream_graph", 0)
    try:
        rc = L.nnhip_ode_fixed_stream_multi_gpu_f64_dev(C.byref(optf), nn.ode.integrator_id("rk4"), f.kind, p.ctypes.data_as(dp), 1, G, (C.c_int64 * G)(*counts), d, 0, 0.0, 0.5,
                                                        arr(ys), arr(scr), None, arr([None] * G), None, C.byref(nst), fin)
    finally:
        L.nnhip_tune_set(b"stream_graph", -1)
    check_node()
    assert rc == 0 and nst.value == 16, (rc, nst.value, nn._lib.last_error())
    for r in range(G):
        if counts[r]:
            got = ys[r] if fin[r] == ys[r] else scr[r]
            assert np.array_equal(dev_view(got, (d, counts[r])), reff["y"][-1][:, lo[r]:lo[r + 1]]), r
    for ptr in y0s + outs + wss + fulls + ys + scr:
        F.hipFree(C.c_void_p(ptr))
    L.nnhip_ode_rhs_release(f.kind)
    return "device-resident shards read their columns of the context: fused + gathered + streamed == the oracle's closures on %d devices" % G


def scenario_consumers_results():
    """The section-8-f4 consumers through their host entries, results against the reference's text: function-form cumtrapz / cumsimpson (run-time compiled integrand),
    the discrete forms, newHermiteSpline's slopes and eval / derivEval with every ExtrapolateKind."""
    V = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_text_quad_vectors.json")))
    src = "for (int c = 0; c < {d}; ++c) dy[c] = ((p[0] * t + p[1]) * t) * (1.0 + (double)c) + p[2];"
    fs = {d: nn.Rhs.custom(d, src.format(d=d), keys=("a", "b", "c"), name="poly%d_isa" % d) for d in (1, 3)}
    n_items, done = 3, 0
    cases = V["cumquad_fn"] if FULL else V["cumquad_fn"][::9]
    for c in cases:
        d = max(c["dim"], 1)
        X = fh(c["X"])
        p = np.ascontiguousarray(fh(c["params"])[:3])
        out = np.full((len(X), d, n_items), -7.0)
        rows = C.c_int(-1)
        fn = L.nnhip_cumtrapz_fn_batch_f64 if c["rule"] == "trapz" else L.nnhip_cumsimpson_fn_batch_f64
        rc = fn(fs[d].kind, p.ctypes.data_as(dp), 3, None, 0, n_items, d, 0, X.ctypes.data_as(dp), len(X), float.fromhex(c["dx"]), out.ctypes.data_as(dp), C.byref(rows), 0)
        check_node()
        assert rc == 0 and rows.value == c["rows"], (c["name"], nn._lib.last_error())
        got = out.reshape(-1)[:rows.value * d * n_items].reshape(rows.value, d, n_items)
        for i in (0, n_items - 1):
            assert np.array_equal(got[:, :, i].ravel(), fh(c["out"])), c["name"]
        done += 1
    for c in V["cumquad_discrete"]:
        if not c["strictly_ascending"]:
            continue
        X = fh(c["X"])
        Y = np.ascontiguousarray(np.tile(np.stack([fh(y) for y in c["Y"]], axis=1), (1, 30)))
        out = np.empty_like(Y)
        assert L.nnhip_cumtrapz_batch_f64(X.ctypes.data_as(dp), len(X), Y.ctypes.data_as(dp), Y.shape[1], out.ctypes.data_as(dp), 0) == 0
        check_node()
        assert np.array_equal(out, np.tile(np.stack([fh(v) for v in c["cumtrapz"]], axis=1), (1, 30))), c["name"]
        if not isinstance(c["cumsimpson"], dict):
            assert L.nnhip_cumsimpson_batch_f64(X.ctypes.data_as(dp), len(X), Y.ctypes.data_as(dp), Y.shape[1], out.ctypes.data_as(dp), 0) == 0
            check_node()
            assert np.array_equal(out, np.tile(np.stack([fh(v) for v in c["cumsimpson"]], axis=1), (1, 30))), c["name"]
    EX = {"Constant": 0, "Edge": 1, "Linear": 2, "Native": 3}
    for c in V["hermite"]:
        X, Yv, dYv, xq = fh(c["X"]), fh(c["Y"]), fh(c["dY"]), fh(c["xq"])
        M = 70
        Y = np.ascontiguousarray(np.tile(Yv[:, None], (1, M)))
        dY = np.ascontiguousarray(np.tile(dYv[:, None], (1, M)))
        out = np.empty((len(xq), M))
        for ex, rec in c["with_dY"].items():
            if ex not in EX:
                continue
            for deriv, key in ((0, "eval"), (1, "derivEval")):
                rc = L.nnhip_hermite_spline_eval_batch_f64(X.ctypes.data_as(dp), len(X), Y.ctypes.data_as(dp), dY.ctypes.data_as(dp), M, xq.ctypes.data_as(dp), len(xq), deriv, EX[ex],
                                                           float.fromhex(c["extrap_value"]), out.ctypes.data_as(dp), 0)
                check_node()
                assert rc == 0 and np.array_equal(out, np.tile(fh(rec[key])[:, None], (1, M))), (c["name"], ex, key)
    for f in fs.values():
        L.nnhip_ode_rhs_release(f.kind)
    return "consumers with results: %d function-form cases, the discrete forms, the spline with every ExtrapolateKind == the reference's text" % done


def scenario_bin_order_results():
    """nnhip_ode_bin_order_f64_dev with its kernels running (key range, bin count, bin place): a permutation, ascending from slice to slice — for keys whose range
    touches zero (linear bins, this round), same-signed keys over six decades (logarithmic image) and keys with NaN / inf (last)."""
    n = 3 * 4096 + 77
    rng = np.random.default_rng(3)
    s = C.c_void_p()
    assert F.hipStreamCreateWithFlags(C.byref(s), 1) == 0
    kd = dev_alloc(0, n * 8)
    od = dev_alloc(0, n * 4)
    shapes = {"uniform_with_a_zero": np.concatenate([[0.0], rng.uniform(0.0, 10.0, n - 1)]), "both_signs": rng.uniform(-5.0, 10.0, n),
              "one_sign_six_decades": -10.0 ** rng.uniform(-6.0, 0.0, n), "with_nan_and_inf": rng.uniform(-1.0, 1.0, n)}
    shapes["with_nan_and_inf"][::97] = np.nan
    shapes["with_nan_and_inf"][5::313] = np.inf
    for name, keys in shapes.items():
        rng.shuffle(keys)
        dev_view(kd, (n,))[...] = keys
        assert L.nnhip_ode_bin_order_f64_dev(kd, n, od, s) == 0, nn._lib.last_error()
        check_node()
        o = dev_view(od, (n,), np.uint32).astype(np.int64)
        assert np.array_equal(np.sort(o), np.arange(n)), name
        k = keys[o]
        fin = np.isfinite(k)
        assert fin[:fin.sum()].all(), name                                   # non-finite keys last
        kf = k[:fin.sum()]
        if name == "one_sign_six_decades":
            viol = np.maximum.accumulate(np.log10(-kf[::-1]))[::-1]          # ascending keys = descending magnitudes
            assert float((np.log10(-kf) - viol).max()) <= 6.0 / 4094 * 2.02, name
        else:
            viol = np.maximum.accumulate(kf) - kf
            assert float(viol.max()) <= 1.01 * (kf.max() - kf.min()) / 4094, (name, float(viol.max()))
    F.hipFree(C.c_void_p(kd))
    F.hipFree(C.c_void_p(od))
    return "order of integration: 4 key shapes, permutations ascending from slice to slice"


def main():
    only = sys.argv[1:]
    t00 = time.time()
    for sc in (scenario_golden_fixtures, scenario_sharded_context_results, scenario_mutable_slots_results, scenario_two_threads_results, scenario_streaming_results,
               scenario_c5_shape_with_results, scenario_device_resident_shards_results, scenario_consumers_results, scenario_bin_order_results):
        if only and sc.__name__.replace("scenario_", "") not in only:
            continue
        t0 = time.time()
        msg = sc()
        check_node()
        print("%s -> %s  [%.0f s]" % (sc.__name__, msg, time.time() - t0), flush=True)
    L.nnhip_release()
    for _ in range(100):
        if F.fake_hip_live_device_allocations() == 0:
            break
        time.sleep(0.05)
    print("kernel launches interpreted: %d, wave-instructions: %.1f M, live device allocations: %d, %.0f s" % (
        NODE.launches, NODE.instructions / 1e6, F.fake_hip_live_device_allocations(), time.time() - t00), flush=True)
    print("ALL OK", flush=True)


if __name__ == "__main__":
    main()
