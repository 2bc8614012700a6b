#!/usr/bin/env python3
"""A user right-hand side from source: method-of-lines heat equation, 64 unknowns per rod, 2e5 rods with their own conductivity.
The body is written per component (it returns dy_c) and declares the halo it reads (c - 1 .. c + 1), so the lanes-per-system kernels take
the neighbours from the adjacent lanes instead of the LDS stage vector; same bits as without the declaration.

    python examples/heat_stencil.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import numericalnim_amd as nn

dev = torch.device("cuda:0")
dim, n = 64, 200_000
BODY = ("const double l = c > 0 ? y[c - 1] : 0.0;            // Dirichlet ends\n"
        "const double r = c + 1 < dim ? y[c + 1] : 0.0;\n"
        "return p[0] * ((l - 2.0 * y[c]) + r);")
rng = np.random.default_rng(0)
y0 = torch.from_numpy(np.sin(np.pi * (np.arange(dim) + 1) / (dim + 1))[None, :] * rng.uniform(0.5, 1.5, (n, 1))).to(dev)   # AoS: one rod contiguous
kappa = torch.from_numpy(rng.uniform(0.2, 2.0, n)).to(dev)                                                                 # every rod its own ctx["kappa"]
opt = nn.newODEoptions(absTol=1e-8, relTol=1e-8, dtMin=1e-8, dtMax=0.25)
plain = nn.Rhs.custom(dim, BODY, keys=("kappa",), defaults={"kappa": 1.0}, name="heat_plain", per_component=True)
banded = nn.Rhs.custom(dim, BODY, keys=("kappa",), defaults={"kappa": 1.0}, name="heat_banded", per_component=True, halo=(1, 1))


def timed(fn):
    fn(); torch.cuda.synchronize(); c0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return r, (time.perf_counter() - c0) * 1e3


(t, ya), ms_a = timed(lambda: nn.solveODE(plain, y0, [0.0, 1.0], opt, integrator="tsit54", layout=nn.LAYOUT_AOS, sweep=kappa[None, :]))
(_, yb), ms_b = timed(lambda: nn.solveODE(banded, y0, [0.0, 1.0], opt, integrator="tsit54", layout=nn.LAYOUT_AOS, sweep=kappa[None, :]))
print(f"fused solve: plain {ms_a:.2f} ms | halo declared {ms_b:.2f} ms | identical: {torch.equal(ya, yb)}")
# the slowest mode of rod i decays like exp(-kappa_i * lambda_1 * t), lambda_1 = 2 - 2 cos(pi / (dim + 1))
lam = 2.0 - 2.0 * np.cos(np.pi / (dim + 1))
ratio = (yb[-1][:, dim // 2] / y0[:, dim // 2]).cpu().numpy()
print("decay of the fundamental mode vs exp(-kappa * lambda_1): max abs deviation %.2e" % np.abs(ratio - np.exp(-kappa.cpu().numpy() * lam)).max())
