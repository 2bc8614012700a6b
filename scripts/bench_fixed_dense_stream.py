#!/usr/bin/env python3
"""nnhip_ode_fixed_stream_dense_f64_dev (the whole ODESolver through the IntegratorProc seam, fixed-step methods): time per solve and per emitted
row; C2-shaped scalar batch and Lorenz.  The rows are interpolated by one fused kernel per batch of due rows (dense_rows_kernel) for the
compiled-in right-hand sides."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import numericalnim_amd as nn
dev = torch.device("cuda:0")
def timed(fn, reps=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
out = {}
n = 10_000_000
y1 = torch.from_numpy(1.0 + (np.arange(n) % 2 ** 20) * 2.0 ** -20).to(dev)
o = nn.newODEoptions(dt=2.0 ** -10)
for nt in (2, 9, 33, 129):
    ts = np.linspace(0.0, 0.25, nt)
    out[f"scalar_1e7_rk4_256steps_nt{nt}_ms"] = timed(lambda: nn.fixedStreamSolve(nn.Rhs.neg_y(), y1, ts, o, integrator="rk4"))
n = 1_000_000
y3 = torch.from_numpy(np.stack([1.0 + (np.arange(n) % 1024) * 2.0 ** -20, np.ones(n), np.ones(n)])).to(dev)
o3 = nn.newODEoptions(dt=1e-3)
for nt in (2, 11, 101):
    ts = np.linspace(-0.1, 0.2, nt)
    out[f"lorenz_1e6_rk4_300steps_nt{nt}_ms"] = timed(lambda: nn.fixedStreamSolve(nn.Rhs.lorenz(), y3, ts, o3, integrator="rk4"))
print(json.dumps(out, indent=1))
