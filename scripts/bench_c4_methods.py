#!/usr/bin/env python3
"""Streamed C4 shape (1e6 x 16-component ring, AoS) through the HBM-resident adaptive loop for every adaptive method: if the cheap
methods (RK21: 2 stages, BS32: 4) take as long per loop iteration as the 7- and 9-stage ones, the kernel is bound by its memory
access pattern, not by arithmetic."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import numericalnim_amd as nn
dev = torch.device("cuda:0")
n = 1_000_000
y16 = torch.from_numpy(1.0 + np.arange(16)[None, :] / 16 + ((np.arange(n) % 1024) * 2.0 ** -20)[:, None]).to(dev)
side = torch.cuda.Stream()
res = {}
for integ in ("rk21", "bs32", "dopri54", "tsit54", "vern65"):
    opt = nn.newODEoptions(dtMax=1e-2, dtMin=1e-4)
    t, yf, cnt = nn.solveODE(nn.Rhs.ring(0.1), y16, [0.0, 1.0], opt, integrator=integ, layout=1, return_counts=True)
    iters = int(cnt["steps"].max()); accepted = int(cnt["steps"].sum())
    per_step = 8 * (4 * 16 + 4) if integ == "vern65" else 8 * (2 * 16 + 4)   # FSAL travels through HBM only for Vern65 (see adv_fsal_in_hbm)
    best = None
    with torch.cuda.stream(side):
        for _ in range(4):
            y = y16.clone(); side.synchronize(); c0 = time.perf_counter()
            ys, launches = nn.adaptiveStream(nn.Rhs.ring(0.1), y, 0.0, 1.0, opt, integrator=integ, layout=1, check_every=8)
            side.synchronize(); c1 = time.perf_counter()
            best = min(best, c1 - c0) if best else c1 - c0
    res[integ] = dict(us_per_iteration=best * 1e6 / iters, iterations=iters, rejected=int(cnt["rejected"].sum()), bytes_per_step=per_step, GBps=per_step * accepted / best / 1e9, frac_of_8TBps=per_step * accepted / best / 8e12, equal=bool(torch.equal(ys, yf[-1])))
    print(integ, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in res[integ].items()}, flush=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r03_bench_c4_methods.json"), "w"), indent=1)
