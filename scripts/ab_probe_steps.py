#!/usr/bin/env python3
"""Automatic divergence binning: how many accepted steps does the probe solve need?  scripts/bench_divergence.py's Van der Pol sweep (DOPRI54,
1e6 IVPs in random order), sort_by = "auto" with probe_steps from 2 to 16 against the caller's key and the unsorted solve."""
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
sw = torch.from_numpy(mu[None, :].copy()).to(dev)
res = {}
def run(sort_by, probe_steps=0, counts=False):
    tt = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
        out = nn.solveODE(nn.Rhs.vanderpol(), y0, [0.0, 10.0], opt, integrator="dopri54", sweep=sw, sort_by=sort_by, probe_steps=probe_steps, return_counts=counts); e1.record()
        torch.cuda.synchronize(); tt.append(e0.elapsed_time(e1))
    return sorted(tt[1:])[1], out[1]
res["unsorted_ms"], ref = run(None)
res["sort_key_mu_ms"], y = run(sw[0])
for ps in (1, 2, 3, 4, 6, 8, 12, 16):
    ms, y = run("auto", ps)
    res[f"auto_probe_{ps}"] = dict(ms=ms, bit_identical=bool(torch.equal(torch.nan_to_num(y, nan=-1.0), torch.nan_to_num(ref, nan=-1.0))))
# with the per-IVP counters (steps, rejected, rows) written as well
res["with_counters"] = dict(unsorted_ms=run(None, 0, True)[0], sort_key_mu_ms=run(sw[0], 0, True)[0], auto_probe_8_ms=run("auto", 8, True)[0], auto_probe_12_ms=run("auto", 12, True)[0])
print(json.dumps(res, indent=1))
