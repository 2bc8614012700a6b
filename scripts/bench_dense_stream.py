#!/usr/bin/env python3
"""The dense adaptive streaming driver (nnhip_ode_adaptive_stream_dense_f64_dev: ODESolver INCLUDING the emission block ode.nim:512-530 through
the IntegratorProc seam) against its own byte model and against the loop without dense output.

  bytes per step             8*(2d+4)   state in/out (y, t, dt; FSAL is re-evaluated per launch by DOPRI54 / Tsit54)  as the non-dense loop
                           + 4          denseIndex read (lastIter = (t, y, dy), :526-530, stays in the launch's registers since round 3:
                                        the launch that takes a step also emits the requested times that step passed)
  bytes per emitted row      8*d + 4    the row itself, denseIndex written back

(a rejected attempt is retried inside the launch, so the state moves once per ACCEPTED step here — the non-dense loop moves it per attempt).
C3 shape (Lorenz, SoA, default options), 1e6 and 1e7 IVPs, 11 and 101 requested times; hipGraph-replayed polling groups and eager launches.
Wall clock of the whole call (both init kernels, every polling group, the finalize kernels, the speculative tail group), best of 3."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import numericalnim_amd as nn

dev = torch.device("cuda:0")
L = nn._lib.lib()
side = torch.cuda.Stream()
res = {}
d = 3
for n in (1_000_000, 10_000_000):
    y0 = torch.from_numpy(np.stack([1.0 + (np.arange(n) % 1024) * 2.0 ** -20, np.ones(n), np.ones(n)])).to(dev)
    for integ in ("dopri54", "tsit54"):
        t2, y2, cnt = nn.solveODE(nn.Rhs.lorenz(), y0, [0.0, 1.0], nn.newODEoptions(), integrator=integ, return_counts=True)
        iters = int(cnt["steps"].max())
        attempted = int(cnt["steps"].sum() + cnt["rejected"].sum())
        # the loop WITHOUT dense output, same call pattern, for the ratio
        base = None
        with torch.cuda.stream(side):
            for _ in range(4):
                y = y0.clone()
                side.synchronize()
                c0 = time.perf_counter()
                nn.adaptiveStream(nn.Rhs.lorenz(), y, 0.0, 1.0, nn.newODEoptions(), integrator=integ)
                side.synchronize()
                dt_ = time.perf_counter() - c0
                base = dt_ if base is None or dt_ < base else base
        for n_t in (11, 101):
            ts = np.linspace(0.0, 1.0, n_t)
            tf, yf = nn.solveODE(nn.Rhs.lorenz(), y0, ts, nn.newODEoptions(), integrator=integ)
            for mode, knob in (("graph", 1), ("eager", 0)):
                L.nnhip_tune_set(b"stream_graph", knob)
                best, out = None, None
                with torch.cuda.stream(side):
                    for _ in range(4):  # first call: capture
                        side.synchronize()
                        c0 = time.perf_counter()
                        t, y, ny, launches = nn.adaptiveStreamSolve(nn.Rhs.lorenz(), y0, ts, nn.newODEoptions(), integrator=integ)
                        side.synchronize()
                        dt_ = time.perf_counter() - c0
                        if best is None or dt_ < best:
                            best, out = dt_, (y, launches)
                rows = (n_t - 1) * n  # emitted rows besides y0's own (t0 is in tspan)
                accepted = int(cnt["steps"].sum())
                nbytes = accepted * (8 * (2 * d + 4) + 4) + rows * (8 * d + 4)
                res[f"C3_N{n:.0e}_{integ}_nt{n_t}_{mode}"] = dict(
                    ms=best * 1e3, launches=out[1], iterations=iters, us_per_iteration=best * 1e6 / iters, GBps=nbytes / best / 1e9,
                    frac_of_8TBps=nbytes / best / 8e12, bytes_model="accepted*(8(2d+4)+4) + rows*(8d+4)",
                    nondense_stream_ms=base * 1e3, ratio_to_nondense=best / base, equal_to_fused=bool(torch.equal(out[0], yf)))
            L.nnhip_tune_set(b"stream_graph", 2)
# C4 shape through the lanes-per-system dense kernel (16-component ring, AoS, Tsit54), 11 requested times
n, d = 1_000_000, 16
y16 = torch.from_numpy(1.0 + np.arange(16)[None, :] / 16 + ((np.arange(n) % 1024) * 2.0 ** -20)[:, None]).to(dev)
for integ in ("tsit54", "dopri54"):
    t2, y2, cnt = nn.solveODE(nn.Rhs.ring(0.1), y16, [0.0, 1.0], nn.newODEoptions(), integrator=integ, layout=1, return_counts=True)
    iters, accepted = int(cnt["steps"].max()), int(cnt["steps"].sum())
    ts = np.linspace(0.0, 1.0, 11)
    tf, yf = nn.solveODE(nn.Rhs.ring(0.1), y16, ts, nn.newODEoptions(), integrator=integ, layout=1)
    base = best = None
    with torch.cuda.stream(side):
        for _ in range(4):
            y = y16.clone(); side.synchronize(); c0 = time.perf_counter()
            nn.adaptiveStream(nn.Rhs.ring(0.1), y, 0.0, 1.0, nn.newODEoptions(), integrator=integ, layout=1)
            side.synchronize(); dt_ = time.perf_counter() - c0
            base = dt_ if base is None or dt_ < base else base
            side.synchronize(); c0 = time.perf_counter()
            t, y, ny, launches = nn.adaptiveStreamSolve(nn.Rhs.ring(0.1), y16, ts, nn.newODEoptions(), integrator=integ, layout=1)
            side.synchronize(); dt_ = time.perf_counter() - c0
            best = dt_ if best is None or dt_ < best else best
    nbytes = accepted * (8 * (2 * d + 4) + 4) + 10 * n * (8 * d + 4)
    res[f"C4_N1e+06_{integ}_nt11"] = dict(ms=best * 1e3, launches=launches, iterations=iters, us_per_iteration=best * 1e6 / iters, GBps=nbytes / best / 1e9, frac_of_8TBps=nbytes / best / 8e12,
                                         bytes_model="accepted*(8(2d+4)+4) + rows*(8d+4)", nondense_stream_ms=base * 1e3, ratio_to_nondense=best / base,
                                         equal_to_fused=bool(torch.equal(y, yf)))
print(json.dumps(res, indent=1))
