#!/usr/bin/env python3
"""Wide systems on the lanes-per-system kernels instantiated at run time: a method-of-lines heat equation with `dim` unknowns per
system (per-component source), N systems, RK4 (100 steps) and DOPRI54 (default tolerances tightened to 1e-8).  One JSON line."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import numericalnim_amd as nn  # noqa: E402

HEAT = ("const double l = c > 0 ? y[c - 1] : 0.0; const double r = c + 1 < dim ? y[c + 1] : 0.0; "
        "return p[0] * ((l - 2.0 * y[c]) + r);")
dev = torch.device("cuda", 0)
out = {}
for dim, n in ((64, 262144), (100, 131072), (256, 65536)):
    f = nn.Rhs.custom(dim, HEAT, keys=("kappa",), defaults={"kappa": 0.4}, name=f"heat{dim}", per_component=True)
    y0 = torch.rand(n, dim, dtype=torch.float64, device=dev)  # AoS: one system contiguous
    for integ, opt, steps_hint in (("rk4", nn.newODEoptions(dt=1e-2), 100), ("dopri54", nn.newODEoptions(absTol=1e-8, relTol=1e-8, dtMax=0.1, dtMin=1e-7), None)):
        for _ in range(2):
            t, y, cnt = nn.solveODE(f, y0, [0.0, 1.0], opt, integrator=integ, layout=nn.LAYOUT_AOS, return_counts=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            t, y, cnt = nn.solveODE(f, y0, [0.0, 1.0], opt, integrator=integ, layout=nn.LAYOUT_AOS, return_counts=True)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        att = int(cnt["steps"].sum()) + int(cnt["rejected"].sum())
        out[f"heat{dim}_{integ}_N{n}"] = dict(ms=round(ms, 3), attempted_system_steps=att, component_steps_per_s=att * dim / (ms * 1e-3))
print(json.dumps(out))
