#!/usr/bin/env python3
"""One streamed C4 solve per adaptive method (for rocprofv3 --kernel-trace --stats): which kernels run, how long."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import numericalnim_amd as nn
dev = torch.device("cuda:0")
n = 1_000_000
y16 = torch.from_numpy(1.0 + np.arange(16)[None, :] / 16 + ((np.arange(n) % 1024) * 2.0 ** -20)[:, None]).to(dev)
for integ in sys.argv[1:] or ("rk21", "bs32", "dopri54", "tsit54", "vern65"):
    opt = nn.newODEoptions(dtMax=1e-2, dtMin=1e-4)
    for _ in range(2):
        nn.adaptiveStream(nn.Rhs.ring(0.1), y16.clone(), 0.0, 1.0, opt, integrator=integ, layout=1, check_every=8)
    torch.cuda.synchronize()
