#!/usr/bin/env python3
"""For rocprofv3 --kernel-trace: the 16-component ring through the adaptive loop and the fused solve, compiled-in and from source (with and without halo)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import numericalnim_amd as nn
dev = torch.device("cuda:0")
n = 1_000_000
RING = "return -((double)(c + 1) / (double)dim) * y[c] + p[0] * y[(c + 1) % dim];"
y16 = torch.from_numpy(1.0 + np.arange(16)[None, :] / 16 + ((np.arange(n) % 1024) * 2.0 ** -20)[:, None]).to(dev)
for f in (nn.Rhs.ring(0.1), nn.Rhs.custom(16, RING, keys=("c",), defaults={"c": 0.1}, name="ring16_src", per_component=True),
          nn.Rhs.custom(16, RING, keys=("c",), defaults={"c": 0.1}, name="ring16_src_halo", per_component=True, halo=(0, 1))):
    for _ in range(2):
        nn.adaptiveStream(f, y16.clone(), 0.0, 1.0, nn.newODEoptions(), integrator="tsit54", layout=1)
        nn.solveODE(f, y16, [0.0, 1.0], nn.newODEoptions(), integrator="tsit54", layout=1)
    torch.cuda.synchronize()
