#!/usr/bin/env python3
"""A/B body of profiles/r04_tpi_occupancy_ab.json: fused thread-per-IVP solves (1e6 IVPs; Van der Pol, Lorenz, dy = -y; DOPRI54, Tsit54, Vern65, RK4; 2 and 5 rows), one JSON line.
Run once per library variant (NNHIP_LIB), interleaved, on one box."""
import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import numericalnim_amd as nn
dev = torch.device("cuda:0"); n = 1_000_000
rng = np.random.default_rng(2)
def t(fn, reps=9):
    tt=[]
    for r in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record(); out = fn(); e1.record(); torch.cuda.synchronize(); tt.append(e0.elapsed_time(e1))
    return sorted(tt[2:])[len(tt[2:])//2]
res={}
cases = (("vdp", nn.Rhs.vanderpol(2.0), np.stack([rng.uniform(1.5, 2.5, n), np.zeros(n)]), 5.0), ("lorenz", nn.Rhs.lorenz(), np.stack([rng.uniform(-10,10,n), rng.uniform(-10,10,n), rng.uniform(5,30,n)]), 1.0), ("negy", nn.Rhs.neg_y(), rng.uniform(0.5,1.5,n), 5.0))
for name, f, y0, te in cases:
    y0t = torch.from_numpy(y0).to(dev)
    opt = nn.newODEoptions(absTol=1e-6, relTol=1e-6, dtMin=1e-9, dtMax=1.0)
    for integ in ("dopri54", "tsit54", "vern65", "rk4"):
        o = opt if integ != "rk4" else nn.newODEoptions(dt=te/256)
        res[f"{name}_{integ}_nt2"] = round(t(lambda: nn.solveODE(f, y0t, [0.0, te], o, integrator=integ)), 4)
        if integ in ("dopri54",): res[f"{name}_{integ}_nt5"] = round(t(lambda: nn.solveODE(f, y0t, [0.0, te/4, te/2, 3*te/4, te], o, integrator=integ)), 4)
print(json.dumps(res))
