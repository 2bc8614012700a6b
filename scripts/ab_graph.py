#!/usr/bin/env python3
"""A/B: eager launches vs hipGraph replay of the fixed-step streaming loop across batch sizes."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import numericalnim_amd as nn
from numericalnim_amd import distributed as nd
L = nn._lib.lib()
dev = torch.device("cuda:0")
dt, nsteps = 2.0 ** -10, 1000
opt = nn.newODEoptions(dt=dt)
res = {}
side = torch.cuda.Stream()
torch.cuda.set_stream(side)  # hipGraph capture needs a non-default stream
for n in (1024, 10_000, 100_000, 1_000_000, 10_000_000):
    y0 = nd.c2_y0_torch(0, n, dev)
    y, sc = y0.clone(), torch.empty_like(y0)
    outs = []
    for g in (0, 1, 2):
        L.nnhip_tune_set(b"stream_graph", g)
        ts = []
        for r in range(6):
            y.copy_(y0); torch.cuda.synchronize()
            c0 = time.perf_counter()
            yf, k = nn.fixedStream(nn.Rhs.neg_y(), y, 0.0, nsteps * dt, opt, integrator="rk4", scratch=sc)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - c0)
        outs.append(yf.clone())
        res[f"N{n}_{('eager', 'graph', 'auto')[g]}_ms"] = sorted(ts[2:])[1] * 1e3
    res[f"N{n}_equal"] = bool(torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]))
L.nnhip_tune_set(b"stream_graph", 2)
print(json.dumps(res, indent=1))
