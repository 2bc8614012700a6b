#!/usr/bin/env python3
"""Dense-output solves (the reference's documented usage: tspan = linspace(...)): 1e6 Lorenz IVPs, 21 requested times."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import numericalnim_amd as nn
dev = torch.device("cuda:0"); n = 1_000_000
y0 = torch.from_numpy(np.stack([1.0 + (np.arange(n) % 1024) * 2.0 ** -20, np.ones(n), np.ones(n)])).to(dev)
res = {}
for nt in (2, 21, 101):
    ts = np.linspace(0, 1, nt)
    for m in ("rk4", "dopri54", "tsit54", "vern65"):
        tt = []
        for r in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
            nn.solveODE(nn.Rhs.lorenz(), y0, ts, nn.newODEoptions(dt=1e-3), integrator=m); e1.record(); torch.cuda.synchronize()
            tt.append(e0.elapsed_time(e1))
        res[f"{m}_nt{nt}_ms"] = round(sorted(tt[1:])[1], 3)
print(json.dumps(res, indent=1))
