#!/usr/bin/env python3
"""HBM-resident adaptive streaming (advance kernel, 8*(4d+5) B per attempted step) vs the fused solve: C3 shape at 1e6 and 1e7 IVPs."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import numericalnim_amd as nn
dev = torch.device("cuda:0")
res = {}
for n in (1_000_000, 10_000_000):
    y0 = torch.from_numpy(np.stack([1.0 + (np.arange(n) % 1024) * 2.0 ** -20, np.ones(n), np.ones(n)])).to(dev)
    for integ in ("dopri54", "tsit54"):
        opt = nn.newODEoptions()
        nn.adaptiveStream(nn.Rhs.lorenz(), y0.clone(), 0.0, 1.0, opt, integrator=integ)
        y = y0.clone(); torch.cuda.synchronize(); c0 = time.perf_counter()
        ys, launches = nn.adaptiveStream(nn.Rhs.lorenz(), y, 0.0, 1.0, opt, integrator=integ, check_every=16)
        torch.cuda.synchronize(); c1 = time.perf_counter()
        t, yf = nn.solveODE(nn.Rhs.lorenz(), y0, [0.0, 1.0], opt, integrator=integ)
        torch.cuda.synchronize(); c2 = time.perf_counter()
        steps = 102  # every IVP takes 102 steps with default options (controller pinned at dtMax)
        res[f"{integ}_N{n:.0e}"] = dict(stream_ms=(c1 - c0) * 1e3, launches=launches, us_per_launch=(c1 - c0) * 1e6 / launches,
                                        GBps=8 * (4 * 3 + 5) * n * steps / (c1 - c0) / 1e9, fused_ms=(c2 - c1) * 1e3, equal=bool(torch.equal(ys, yf[-1])))
print(json.dumps(res, indent=1))
