#!/usr/bin/env python3
"""Polling group size of the adaptive streaming loop (check_every launches per "anyone left?" poll): C3 1e6 / 1e7 and C4, interleaved."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import numericalnim_amd as nn
dev = torch.device("cuda:0")
side = torch.cuda.Stream()
res = {}
cases = []
for n in (1_000_000, 10_000_000):
    cases.append((f"C3_N{n:.0e}", nn.Rhs.lorenz(), torch.from_numpy(np.stack([1.0 + (np.arange(n) % 1024) * 2.0 ** -20, np.ones(n), np.ones(n)])).to(dev), 0, "dopri54"))
n = 1_000_000
cases.append(("C4", nn.Rhs.ring(0.1), torch.from_numpy(1.0 + np.arange(16)[None, :] / 16 + ((np.arange(n) % 1024) * 2.0 ** -20)[:, None]).to(dev), 1, "tsit54"))
for name, f, y0, layout, integ in cases:
    iters = int(nn.solveODE(f, y0, [0.0, 1.0], nn.newODEoptions(), integrator=integ, layout=layout, return_counts=True)[2]["steps"].max())
    tt = {}
    with torch.cuda.stream(side):
        for r in range(9):
            for ce in (2, 4, 8, 16, 32):
                y = y0.clone(); side.synchronize(); c0 = time.perf_counter()
                ys, launches = nn.adaptiveStream(f, y, 0.0, 1.0, nn.newODEoptions(), integrator=integ, layout=layout, check_every=ce)
                side.synchronize(); tt.setdefault(ce, []).append((time.perf_counter() - c0, launches))
    res[name] = {f"check_every_{ce}": dict(us_per_iteration=round(sorted(x[0] for x in v[2:])[len(v[2:]) // 2] * 1e6 / iters, 2), launches=v[-1][1]) for ce, v in tt.items()}
print(json.dumps(res, indent=1))
