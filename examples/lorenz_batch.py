#!/usr/bin/env python3
"""One million Lorenz trajectories with DOPRI54 on an MI355X, then the README's Hermite-spline recipe on the result.

    python examples/lorenz_batch.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import numericalnim_amd as nn

dev = torch.device("cuda:0")
n = 1_000_000
y0 = torch.stack([1.0 + torch.rand(n, dtype=torch.float64, device=dev) * 1e-3,       # [dim, N] (SoA): coalesced for thread-per-IVP kernels
                  torch.ones(n, dtype=torch.float64, device=dev), torch.ones(n, dtype=torch.float64, device=dev)])
ctx = nn.newNumContext()
ctx.setF("sigma", 10.0); ctx.setF("rho", 28.0); ctx.setF("beta", 8.0 / 3.0)           # parameters travel in ctx.fValues, as in numericalnim
tspan = np.linspace(0.0, 2.0, 41)
opt = nn.newODEoptions(absTol=1e-8, relTol=1e-8, dtMax=0.1, dtMin=1e-7)
t, y, counts = nn.solveODE(nn.Rhs.lorenz(), y0, tspan, opt, ctx, integrator="dopri54", return_counts=True)
print("t:", t.shape, " y:", tuple(y.shape), " mean accepted steps:", float(counts["steps"].double().mean()))

# (t, y, dy) -> cubic Hermite spline through every trajectory, evaluated between the knots
dy = torch.stack([nn.rhsBatch(nn.Rhs.lorenz(), t[j], y[j], ctx) for j in range(len(t))])
spline = nn.newHermiteSpline(t, y, dy)
print("x(1.2345) of the first 3 trajectories:", spline.eval(1.2345)[0, :3].tolist())

# a right-hand side that is not compiled in: hand its source over (run-time compiled with hiprtc)
duffing = nn.Rhs.custom(2, "dy[0] = y[1]; dy[1] = -p[0]*y[1] + y[0] - y[0]*y[0]*y[0] + p[1]*t;", keys=("delta", "gamma"),
                        defaults=dict(delta=0.3, gamma=0.1))
t2, y2 = nn.solveODE(duffing, torch.rand(2, 100_000, dtype=torch.float64, device=dev), [0.0, 5.0], integrator="tsit54")
print("Duffing end state of IVP 0:", y2[-1][:, 0].tolist())

# a parameter sweep: every trajectory gets its own rho (the other parameters stay batch-wide)
rho = torch.linspace(20.0, 35.0, n, dtype=torch.float64, device=dev)
sigma = torch.full_like(rho, 10.0)
t3, y3 = nn.solveODE(nn.Rhs.lorenz(), y0, [0.0, 1.0], opt, integrator="tsit54", sweep=torch.stack([sigma, rho]))
print("z(1) across the rho sweep (first / last IVP):", float(y3[-1][2, 0]), float(y3[-1][2, -1]))
