#!/usr/bin/env python3
"""Extra single-GPU measurements: step-streaming adaptive kernels at N = 1e7 (steady state), dense-output fused solves
(HBM-write bound), Hermite-spline consumer bandwidth."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import numericalnim_amd as nn
from numericalnim_amd import distributed as nd
dev = torch.device("cuda:0")


def timed(fn, reps=5, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return sorted(ts)[len(ts) // 2], r


out = {}
tight = dict(absTol=1e-10, relTol=1e-10, dtMin=1e-6, dtMax=1e-1)
for n in (1_000_000, 10_000_000):
    y0 = torch.from_numpy(np.stack([1.0 + (np.arange(n) % 1024) * 2.0 ** -20, np.ones(n), np.ones(n)])).to(dev)
    fs = nn.rhsBatch(nn.Rhs.lorenz(), 0.0, y0)
    tdev = torch.zeros(n, dtype=torch.float64, device=dev)
    dtdev = torch.full((n,), 1e-3, dtype=torch.float64, device=dev)
    opt = nn.newODEoptions(**tight)
    ynew = torch.empty_like(y0)
    for integ in ("rk4", "dopri54", "tsit54", "vern65"):
        if integ == "rk4":
            # RK4_step[Vector] (ode.nim:180-189) with per-IVP (t, dt): reads y(3)+t+dt, writes yNew(3) = 8*(2d+2) B per IVP-step
            # (r01 measured this with the FSAL slot written as well — 24 B per IVP more than it counted)
            s, _ = timed(lambda: nn.integratorStep(nn.Rhs.lorenz(), tdev, y0, None, dtdev, opt, integrator=integ, out=ynew))
            out[f"step_stream_lorenz_rk4_N{n:.0e}"] = dict(us=s * 1e6, GBps=8 * (2 * 3 + 2) * n / s / 1e9, algorithmic_bytes=8 * (2 * 3 + 2) * n)
            s, _ = timed(lambda: nn.integratorStep(nn.Rhs.lorenz(), tdev, y0, fs, dtdev, opt, integrator=integ, out=ynew))
            out[f"step_stream_lorenz_rk4_fsalslot_N{n:.0e}"] = dict(us=s * 1e6, GBps=8 * (3 * 3 + 2) * n / s / 1e9, algorithmic_bytes=8 * (3 * 3 + 2) * n)
            s, _ = timed(lambda: nn.integratorStep(nn.Rhs.lorenz(), 0.0, y0, None, 1e-3, opt, integrator=integ, out=ynew))
            out[f"step_stream_lorenz_rk4_uniform_N{n:.0e}"] = dict(us=s * 1e6, GBps=8 * (2 * 3) * n / s / 1e9, algorithmic_bytes=8 * (2 * 3) * n)
            continue
        s, _ = timed(lambda: nn.integratorStep(nn.Rhs.lorenz(), tdev, y0, fs, dtdev, opt, integrator=integ, out=ynew))
        out[f"step_stream_lorenz_{integ}_N{n:.0e}"] = dict(us=s * 1e6, GBps=8 * (4 * 3 + 5) * n / s / 1e9)
# the Python seam itself: 20 back-to-back integratorStep calls with preallocated outputs (what a per-step Python driver does) at C3's size, and the
# host time of one call (1024 IVPs: the kernel is a few microseconds, the loop is bound by the wrapper)
import time
for n, tag in ((1_000_000, "N1e+06"), (1024, "host_bound_N1024")):
    y0 = torch.from_numpy(np.stack([1.0 + (np.arange(n) % 1024) * 2.0 ** -20, np.ones(n), np.ones(n)])).to(dev)
    fs = nn.rhsBatch(nn.Rhs.lorenz(), 0.0, y0)
    tdev = torch.zeros(n, dtype=torch.float64, device=dev)
    dtdev = torch.full((n,), 1e-3, dtype=torch.float64, device=dev)
    opt = nn.newODEoptions(**tight)
    bufs = (torch.empty_like(y0), torch.empty_like(y0), torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev))
    rhs = nn.Rhs.lorenz()
    for integ in ("rk4", "dopri54"):
        def calls(k=20):
            for _ in range(k):
                nn.integratorStep(rhs, tdev, y0, fs, dtdev, opt, integrator=integ, out=bufs)
        s, _ = timed(calls, reps=5)
        calls(200); torch.cuda.synchronize()
        c0 = time.perf_counter(); calls(200); c1 = time.perf_counter(); torch.cuda.synchronize()
        out[f"python_seam_integratorStep_{integ}_{tag}"] = dict(us_per_call_back_to_back=s / 20 * 1e6, host_us_per_call_enqueue_only=(c1 - c0) / 200 * 1e6)
# dense output: C2-like, RK4 fused, 1e7 IVPs, 33 requested times -> 33 x 80 MB written
n = 10_000_000
y2 = nd.c2_y0_torch(0, n, dev)
o2 = nn.newODEoptions(dt=2.0 ** -10)
for nt in (2, 9, 33):
    ts = np.linspace(0.0, 0.25, nt)
    s, _ = timed(lambda: nn.solveODE(nn.Rhs.neg_y(), y2, ts, o2, integrator="rk4"), reps=3)
    out[f"fused_rk4_dense_nt{nt}"] = dict(ms=s * 1e3, out_GB=8.0 * n * nt / 1e9, write_GBps=8.0 * n * nt / s / 1e9, steps=256)
# hermite consumer bandwidth: 40 B per output element
t, y = nn.solveODE(nn.Rhs.neg_y(), y2, np.linspace(0.0, 0.25, 9), o2, integrator="rk4")
dy = -y
spl = nn.newHermiteSpline(t, y, dy)
xq = np.linspace(0.01, 0.24, 16)
s, _ = timed(lambda: spl.eval(xq), reps=3)
out["hermite_eval_16q_1e7"] = dict(ms=s * 1e3, GBps=40.0 * n * 16 / s / 1e9)
print(json.dumps(out, indent=1))
