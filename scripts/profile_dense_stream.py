"""Workload for rocprofv3 --kernel-trace --stats: the adaptive loop without dense output and the dense streaming driver (11 and 101 requested
times) on the C3 shape, three rounds each.  usage: rocprofv3 --kernel-trace --stats -d OUT -- python scripts/profile_dense_stream.py [N]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numericalnim_amd as nn
dev = torch.device("cuda:0")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
y0 = torch.from_numpy(np.stack([1.0 + (np.arange(n) % 1024) * 2.0 ** -20, np.ones(n), np.ones(n)])).to(dev)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3):
        y = y0.clone()
        nn.adaptiveStream(nn.Rhs.lorenz(), y, 0.0, 1.0, nn.newODEoptions(), integrator="dopri54")
        for n_t in (2, 11, 101):
            ts = np.linspace(0.0, 1.0, n_t)
            nn.adaptiveStreamSolve(nn.Rhs.lorenz(), y0, ts, nn.newODEoptions(), integrator="dopri54")
    side.synchronize()
