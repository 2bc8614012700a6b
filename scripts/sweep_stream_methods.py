#!/usr/bin/env python3
"""Sanity sweep of the HBM-resident adaptive loop: the five adaptive integrators on the C3 shape (Lorenz, 1e6 and 1e7 IVPs) and, with dense
output (11 requested times), through the dense streaming driver — whole-loop time per loop iteration."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import numericalnim_amd as nn
dev = torch.device("cuda:0")
side = torch.cuda.Stream()
res = {}
for n in (1_000_000, 10_000_000):
    y0 = torch.from_numpy(np.stack([1.0 + (np.arange(n) % 1024) * 2.0 ** -20, np.ones(n), np.ones(n)])).to(dev)
    for integ in ("rk21", "bs32", "dopri54", "tsit54", "vern65"):
        opt = nn.newODEoptions(dtMax=1e-2, dtMin=1e-4)
        t, yf, cnt = nn.solveODE(nn.Rhs.lorenz(), y0, [0.0, 1.0], opt, integrator=integ, return_counts=True)
        iters = int(cnt["steps"].max())
        ts = np.linspace(0.0, 1.0, 11)
        yd = nn.solveODE(nn.Rhs.lorenz(), y0, ts, opt, integrator=integ)[1]
        best = bestd = None
        with torch.cuda.stream(side):
            for _ in range(4):
                y = y0.clone(); side.synchronize(); c0 = time.perf_counter()
                ys, launches = nn.adaptiveStream(nn.Rhs.lorenz(), y, 0.0, 1.0, opt, integrator=integ)
                side.synchronize(); d = time.perf_counter() - c0
                best = d if best is None or d < best else best
                side.synchronize(); c0 = time.perf_counter()
                t2, y2, ny, l2 = nn.adaptiveStreamSolve(nn.Rhs.lorenz(), y0, ts, opt, integrator=integ)
                side.synchronize(); d = time.perf_counter() - c0
                bestd = d if bestd is None or d < bestd else bestd
        res[f"N{n:.0e}_{integ}"] = dict(iterations=iters, us_per_iteration=round(best * 1e6 / iters, 2), dense_us_per_iteration=round(bestd * 1e6 / iters, 2),
                                        equal=bool(torch.equal(ys, yf[-1])), dense_equal=bool(torch.equal(y2, yd)))
print(json.dumps(res, indent=1))
