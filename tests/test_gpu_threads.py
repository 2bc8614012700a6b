"""The C ABI is re-entrant (the only global state is thread-local — error string, staging buffer, graph cache —
plus the process-wide tuning knobs): concurrent calls from several host threads give the bits the same calls give one
after another.  ctypes releases the GIL around every foreign call, so the threads really overlap inside the library."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _jobs(nn):
    rng = np.random.default_rng(11)
    jobs = []
    for k, (integ, kind) in enumerate([("rk4", "negy"), ("dopri54", "lorenz"), ("tsit54", "lorenz"), ("vern65", "vdp"), ("bs32", "negy"),
                                       ("rk21", "lorenz"), ("heun3", "vdp"), ("dopri54", "user")]):
        n = 20000 + 1000 * k
        if kind == "negy":
            f, y0 = nn.Rhs.neg_y(), 1.0 + rng.random(n)
        elif kind == "lorenz":
            f, y0 = nn.Rhs.lorenz(), np.stack([1.0 + rng.random(n), np.ones(n), np.ones(n)])
        elif kind == "vdp":
            f, y0 = nn.Rhs.vanderpol(1.5), np.stack([2.0 * rng.random(n), rng.random(n)])
        else:
            f = nn.Rhs.custom(2, "dy[0] = y[1]; dy[1] = -p[0] * y[0] - 0.1 * y[1];", keys=("k",), defaults={"k": 2.0}, name="osc_threads")
            y0 = np.stack([rng.random(n), rng.random(n)])
        ts = [0.0, 0.3, 0.7, 1.0] if k % 2 else [0.0, 1.0]
        opt = nn.newODEoptions(dt=1e-2) if integ in ("rk4", "heun3") else nn.newODEoptions(absTol=1e-8, relTol=1e-8, dtMax=0.1, dtMin=1e-6)
        jobs.append((f, y0, ts, opt, integ))
    return jobs


def test_concurrent_host_pointer_solves_match_sequential():
    import torch
    import numericalnim_amd as nn
    assert torch.cuda.is_available()
    jobs = _jobs(nn)
    seq = [nn.solveODE(f, y0, ts, opt, integrator=integ)[1] for f, y0, ts, opt, integ in jobs]
    for rounds in range(3):
        got = [None] * len(jobs)
        errs = []

        def run(i):
            try:
                f, y0, ts, opt, integ = jobs[i]
                got[i] = nn.solveODE(f, y0, ts, opt, integrator=integ)[1]
            except Exception as e:  # noqa: BLE001
                errs.append((i, repr(e)))

        th = [threading.Thread(target=run, args=(i,)) for i in range(len(jobs))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for i in range(len(jobs)):
            assert np.array_equal(np.asarray(got[i]), np.asarray(seq[i]), equal_nan=True), (rounds, i, jobs[i][4])


def test_concurrent_device_pointer_solves_on_their_own_streams():
    import torch
    import numericalnim_amd as nn
    dev = torch.device("cuda", 0)
    jobs = _jobs(nn)[:6]
    y0s = [torch.from_numpy(np.ascontiguousarray(j[1])).to(dev) for j in jobs]
    seq = [nn.solveODE(j[0], y, j[2], j[3], integrator=j[4])[1].clone() for j, y in zip(jobs, y0s)]
    torch.cuda.synchronize()
    got = [None] * len(jobs)
    errs = []

    def run(i):
        try:
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s):
                for _ in range(3):
                    got[i] = nn.solveODE(jobs[i][0], y0s[i], jobs[i][2], jobs[i][3], integrator=jobs[i][4])[1]
            s.synchronize()
        except Exception as e:  # noqa: BLE001
            errs.append((i, repr(e)))

    th = [threading.Thread(target=run, args=(i,)) for i in range(len(jobs))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for i in range(len(jobs)):
        assert torch.equal(got[i], seq[i]) or bool(((got[i] == seq[i]) | (got[i].isnan() & seq[i].isnan())).all()), (i, jobs[i][4])


def test_errors_are_per_thread():
    """nnhip_last_error() is thread-local: a failing call in one thread does not disturb the message another thread reads."""
    import numericalnim_amd as nn
    L = nn._lib.lib()
    msgs = {}
    barrier = threading.Barrier(2)

    def bad():
        rc = L.nnhip_ode_integrator_id(b"no_such_integrator")
        assert rc < 0
        rc = L.nnhip_tune_set(b"no_such_knob_from_thread", 1)
        barrier.wait()
        msgs["bad"] = (rc, nn._lib.last_error())

    def good():
        rc = L.nnhip_tune_set(b"another_missing_knob", 1)
        barrier.wait()
        msgs["good"] = (rc, nn._lib.last_error())

    a, b = threading.Thread(target=bad), threading.Thread(target=good)
    a.start(); b.start(); a.join(); b.join()
    assert "no_such_knob_from_thread" in msgs["bad"][1] and "another_missing_knob" in msgs["good"][1], msgs


def test_release_frees_caches_and_everything_is_rebuilt_on_demand():
    """nnhip_release(): staging buffer, graph cache, pooled stream contexts and RCCL communicators go away; the next calls rebuild
    them and give the same bits.  Also callable from a thread that never used the library, and twice in a row."""
    import torch
    import numericalnim_amd as nn
    L = nn._lib.lib()
    dev = torch.device("cuda", 0)
    jobs = _jobs(nn)[:3]
    before = [nn.solveODE(f, y0, ts, opt, integrator=integ)[1] for f, y0, ts, opt, integ in jobs]              # host-pointer path
    yd = torch.from_numpy(np.ascontiguousarray(jobs[1][1])).to(dev)
    dense = nn.solveODE(jobs[1][0], yd, [0.0, 0.2, 0.4, 1.0], jobs[1][3], integrator=jobs[1][4])[1].clone()     # staging buffer
    s = torch.cuda.Stream(device=dev)
    L.nnhip_tune_set(b"stream_graph", 1)
    with torch.cuda.stream(s):
        ys1 = nn.fixedStream(nn.Rhs.neg_y(), torch.ones(5000, dtype=torch.float64, device=dev), 0.0, 1.0, nn.newODEoptions(dt=1e-2), integrator="rk4")[0].clone()
    s.synchronize()
    assert L.nnhip_release() == 0 and L.nnhip_release() == 0
    t = threading.Thread(target=lambda: L.nnhip_release())
    t.start(); t.join()
    after = [nn.solveODE(f, y0, ts, opt, integrator=integ)[1] for f, y0, ts, opt, integ in jobs]
    for a, b in zip(before, after):
        assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
    assert torch.equal(nn.solveODE(jobs[1][0], yd, [0.0, 0.2, 0.4, 1.0], jobs[1][3], integrator=jobs[1][4])[1], dense)
    with torch.cuda.stream(s):
        ys2 = nn.fixedStream(nn.Rhs.neg_y(), torch.ones(5000, dtype=torch.float64, device=dev), 0.0, 1.0, nn.newODEoptions(dt=1e-2), integrator="rk4")[0].clone()
    s.synchronize()
    L.nnhip_tune_set(b"stream_graph", 2)
    assert torch.equal(ys1, ys2)
