#!/usr/bin/env python3
"""Wavefront divergence in the fused adaptive kernels: a Van der Pol sweep over mu (step counts differ ~10x across IVPs),
IVPs in random order vs sorted by mu (neighbouring lanes then need similar step counts)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import numericalnim_amd as nn
dev = torch.device("cuda:0"); n = 1_000_000
rng = np.random.default_rng(0)
mu = rng.uniform(0.1, 20.0, n)
y0 = torch.from_numpy(np.stack([np.full(n, 2.0), np.zeros(n)])).to(dev)
opt = nn.newODEoptions(absTol=1e-6, relTol=1e-6, dtMin=1e-9, dtMax=1.0)
res = {}
for name, order in (("random_order", np.arange(n)), ("sorted_by_mu", np.argsort(mu))):
    sw = torch.from_numpy(mu[order][None, :].copy()).to(dev)
    tt = []
    for r in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
        t, y, cnt = nn.solveODE(nn.Rhs.vanderpol(), y0, [0.0, 10.0], opt, integrator="dopri54", sweep=sw, return_counts=True); e1.record()
        torch.cuda.synchronize(); tt.append(e0.elapsed_time(e1))
    st = (cnt["steps"] + cnt["rejected"]).double()
    w = st.reshape(-1, 64) if n % 64 == 0 else st[: n // 64 * 64].reshape(-1, 64)
    res[name] = dict(ms=sorted(tt)[1], attempted_steps_mean=float(st.mean()), attempted_steps_max=float(st.max()),
                     lane_utilisation=float(w.mean() / w.max(dim=1).values.mean()))
print(json.dumps(res, indent=1))
