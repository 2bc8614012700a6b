#!/usr/bin/env python3
"""Where does replaying the adaptive streaming loop's polling groups from a hipGraph stop paying?  Lorenz (DOPRI54) and the 16-component ring
(Tsit54) at batch sizes from 1e4 to 3e6: whole-loop wall clock per loop iteration, graph replay (stream_graph 1) against eager launches (0)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import numericalnim_amd as nn

dev = torch.device("cuda:0")
L = nn._lib.lib()
side = torch.cuda.Stream()
res = {}
for name, f, dim, layout, integ in (("lorenz", nn.Rhs.lorenz(), 3, 0, "dopri54"), ("ring16", nn.Rhs.ring(0.1), 16, 1, "tsit54")):
    for n in (10_000, 30_000, 100_000, 300_000, 1_000_000, 3_000_000):
        if dim == 3:
            y0 = torch.from_numpy(np.stack([1.0 + (np.arange(n) % 1024) * 2.0 ** -20, np.ones(n), np.ones(n)])).to(dev)
        else:
            y0 = torch.from_numpy(1.0 + np.arange(16)[None, :] / 16 + ((np.arange(n) % 1024) * 2.0 ** -20)[:, None]).to(dev)
        iters = int(nn.solveODE(f, y0, [0.0, 1.0], nn.newODEoptions(), integrator=integ, layout=layout, return_counts=True)[2]["steps"].max())
        row = {}
        for mode, knob in (("graph", 1), ("eager", 0), ("auto", 2)):
            L.nnhip_tune_set(b"stream_graph", knob)
            best = None
            with torch.cuda.stream(side):
                for _ in range(6):
                    y = y0.clone()
                    side.synchronize()
                    c0 = time.perf_counter()
                    nn.adaptiveStream(f, y, 0.0, 1.0, nn.newODEoptions(), integrator=integ, layout=layout)
                    side.synchronize()
                    d = time.perf_counter() - c0
                    best = d if best is None or d < best else best
            row[mode + "_us_per_iteration"] = best * 1e6 / iters
        res[f"{name}_N{n:.0e}"] = row
L.nnhip_tune_set(b"stream_graph", 2)
print(json.dumps(res, indent=1))
