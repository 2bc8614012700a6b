#!/usr/bin/env python3
"""A batch whose members are genuinely different solveODE calls (round-2 entries): a Van der Pol sweep over the stiffness mu in
random order, integrated (1) as handed over, (2) with divergence binning below the C ABI, (3) with every IVP owning its tspan end
and tolerances, and (4) through the IntegratorProc seam with dense output.  All results carry the reference's bits.

    python examples/heterogeneous_batch.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import numericalnim_amd as nn

dev = torch.device("cuda:0")
n = 500_000
rng = np.random.default_rng(1)
mu = torch.from_numpy(rng.uniform(0.1, 20.0, n)).to(dev)                       # every IVP its own ctx["mu"]
y0 = torch.stack([torch.full((n,), 2.0, dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.float64, device=dev)])
opt = nn.newODEoptions(absTol=1e-6, relTol=1e-6, dtMin=1e-9, dtMax=1.0)
f = nn.Rhs.vanderpol()


def timed(fn):
    fn(); torch.cuda.synchronize(); c0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return r, (time.perf_counter() - c0) * 1e3


(t, y), ms0 = timed(lambda: nn.solveODE(f, y0, [0.0, 10.0], opt, integrator="dopri54", sweep=mu[None, :]))
(_, ys), ms1 = timed(lambda: nn.solveODE(f, y0, [0.0, 10.0], opt, integrator="dopri54", sweep=mu[None, :], sort_by=-mu))   # the stiffest (most steps) first
(_, ya), ms2 = timed(lambda: nn.solveODE(f, y0, [0.0, 10.0], opt, integrator="dopri54", sweep=mu[None, :], sort_by="auto"))
print(f"as handed over {ms0:.2f} ms | binned by -mu {ms1:.2f} ms | automatic probe {ms2:.2f} ms | identical: {torch.equal(y, ys) and torch.equal(y, ya)}")

# every IVP its own tspan end and its own tolerances (each reference call owns its tspan and ODEoptions)
t_end = torch.from_numpy(rng.uniform(-2.0, 10.0, n)).to(dev)                    # some integrate backwards
tol = torch.from_numpy(10 ** rng.uniform(-9, -4, n)).to(dev)
yc, cnt = nn.solveODEPerIvpEnd(f, y0, t_end, opt, integrator="tsit54", sweep=mu[None, :], absTol=tol, relTol=tol)
fwd = t_end > 0
print("per-IVP calls: rows hold (y0, y(tEnd)) forward and (y(tEnd), y0) backward;",
      f"{int(fwd.sum())} forward / {int((~fwd).sum())} backward IVPs, accepted steps {int(cnt['steps'].min())}..{int(cnt['steps'].max())}")

# the whole ODESolver — both directions, dense Hermite rows — with the state resident in HBM between stepper calls
ts = np.linspace(-1.0, 3.0, 41)
small = slice(0, 20_000)
tt, yd, ny, launches = nn.adaptiveStreamSolve(nn.Rhs.vanderpol(2.0), y0[:, small].contiguous(), ts, opt, integrator="dopri54")
tf, yf = nn.solveODE(nn.Rhs.vanderpol(2.0), y0[:, small].contiguous(), ts, opt, integrator="dopri54")
print(f"stream-solve through the IntegratorProc seam: {launches} launches, rows {tuple(yd.shape)}, identical to the fused solve: {torch.equal(yd, yf)}")

# every IVP its own output grid as well: 8 requested times per IVP, unsorted, on both sides of tStart
grids = torch.from_numpy(rng.uniform(-1.0, 4.0, (n, 8))).to(dev)
tg, yg, cg = nn.solveODEPerIvpTspan(f, y0, grids, opt, integrator="dopri54", sweep=mu[None, :])
print(f"per-IVP grids: t {tuple(tg.shape)} (row i = IVP i's sorted times), y {tuple(yg.shape)}, rows returned per IVP {int(cg['ny'].min())}..{int(cg['ny'].max())}")
