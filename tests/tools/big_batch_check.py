#!/usr/bin/env python3
"""Large-batch check on one MI355X (288 GB HBM): N = 2^31 + 4096 scalar IVPs (> 2^31 elements per array, 17 GB per state
buffer) through the fused and the step-streaming RK4 paths; verifies 64-bit indexing end to end against the oracle on
samples from the head, the 2^31 boundary and the tail."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import numericalnim_amd as nn
from oracle import oracle as O
dev = torch.device("cuda:0")
n = (1 << 31) + 4096
dt, nsteps = 2.0 ** -6, 8
y0 = torch.empty(n, dtype=torch.float64, device=dev)
blk = 1 << 27
for lo in range(0, n, blk):   # y0[i] = 1 + (i mod 2^20) * 2^-20 + (i // 2^20) * 2^-40, built in slices to bound temporaries
    hi = min(n, lo + blk)
    i = torch.arange(lo, hi, dtype=torch.int64, device=dev)
    y0[lo:hi] = 1.0 + (i % (1 << 20)).double() * 2.0 ** -20 + (i >> 20).double() * 2.0 ** -40
    del i
opt = nn.newODEoptions(dt=dt)
torch.cuda.synchronize(); c0 = time.perf_counter()
t, y = nn.solveODE(nn.Rhs.neg_y(), y0, [0.0, nsteps * dt], opt, integrator="rk4")
torch.cuda.synchronize(); c1 = time.perf_counter()
ys = y0.clone()
yf, k = nn.fixedStream(nn.Rhs.neg_y(), ys, 0.0, nsteps * dt, opt, integrator="rk4")
torch.cuda.synchronize(); c2 = time.perf_counter()
idx = np.concatenate([np.arange(0, 2048), np.arange((1 << 31) - 1024, (1 << 31) + 1024), np.arange(n - 2048, n)]).astype(np.int64)
it = torch.from_numpy(idx).to(dev)
y0s = y0[it].cpu().numpy()
ref = O.solve_ode_batch(O.RHS_NEG_Y, [], y0s, len(idx), 0, [0.0, nsteps * dt], O.new_options(dt=dt), "rk4")["y"][-1, 0]
res = dict(N=n, bytes_per_buffer=8 * n, fused_ms=(c1 - c0) * 1e3, stream_ms=(c2 - c1) * 1e3, steps=k,
           fused_equal_oracle=bool(np.array_equal(y[-1][it].cpu().numpy(), ref)), stream_equal_oracle=bool(np.array_equal(yf[it].cpu().numpy(), ref)),
           row0_is_y0=bool(torch.equal(y[0][it], y0[it])), stream_GBps=16.0 * n * k / (c2 - c1) / 1e9,
           mem_allocated_GB=torch.cuda.max_memory_allocated() / 1e9)
print(json.dumps(res, indent=1))
assert res["fused_equal_oracle"] and res["stream_equal_oracle"] and res["row0_is_y0"]
