#!/usr/bin/env python3
"""Sanity sweep: every integrator through the fused solve on the C3 shape (Lorenz, 1e6 IVPs) and the C4 shape (16-component ring, 1e6 systems),
default options with dt = 1e-2 for the fixed-step methods — time per solve and per right-hand-side evaluation, so that an instantiation that is
out of line with its stage count (a spill, a missed inlining) stands out."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import numericalnim_amd as nn
dev = torch.device("cuda:0")
n = 1_000_000
y3 = torch.from_numpy(np.stack([1.0 + (np.arange(n) % 1024) * 2.0 ** -20, np.ones(n), np.ones(n)])).to(dev)
y16 = torch.from_numpy(1.0 + np.arange(16)[None, :] / 16 + ((np.arange(n) % 1024) * 2.0 ** -20)[:, None]).to(dev)
STAGES = dict(rk4=4, dopri54=6, tsit54=6, vern65=8, bs32=4, rk21=2, heun2=2, ralston2=2, kutta3=3, heun3=3, ralston3=3, ssprk3=3, ralston4=4, kutta4=4)
res = {}
for shape, f, y0, layout in (("lorenz", nn.Rhs.lorenz(), y3, 0), ("ring16", nn.Rhs.ring(0.1), y16, 1)):
    for integ, st in STAGES.items():
        opt = nn.newODEoptions(dt=1e-2)
        tt = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
            t, y, cnt = nn.solveODE(f, y0, [0.0, 1.0], opt, integrator=integ, layout=layout, return_counts=True); e1.record()
            torch.cuda.synchronize(); tt.append(e0.elapsed_time(e1))
        ms = sorted(tt[1:])[1]
        att = float((cnt["steps"] + cnt["rejected"]).double().mean())
        res[f"{shape}_{integ}"] = dict(ms=round(ms, 3), attempted_steps=att, ps_per_ivp_stage=round(ms * 1e9 / n / att / st, 2))
print(json.dumps(res, indent=1))
