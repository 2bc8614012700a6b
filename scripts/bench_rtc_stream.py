#!/usr/bin/env python3
"""Run-time compiled per-component right-hand sides through the HBM-resident adaptive loop: the 16-component ring written as source
(against the compiled-in RhsRing<16>: the same kernel template, 4 components per lane) and method-of-lines heat systems of 24, 64 and 100
unknowns.  Whole loop per iteration."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import numericalnim_amd as nn
dev = torch.device("cuda:0")
side = torch.cuda.Stream()
RING = "return -((double)(c + 1) / (double)dim) * y[c] + p[0] * y[(c + 1) % dim];"  # the compiled-in RhsRing, as source
HEAT = "const double l = c > 0 ? y[c - 1] : 0.0; const double r = c + 1 < dim ? y[c + 1] : 0.0; return p[0] * ((l - 2.0 * y[c]) + r);"
res = {}
def run(name, f, y0, integ="tsit54"):
    opt = nn.newODEoptions()
    t, yf, cnt = nn.solveODE(f, y0, [0.0, 1.0], opt, integrator=integ, layout=1, return_counts=True)
    iters = int(cnt["steps"].max())
    best = None
    with torch.cuda.stream(side):
        for _ in range(4):
            y = y0.clone(); side.synchronize(); c0 = time.perf_counter()
            ys, launches = nn.adaptiveStream(f, y, 0.0, 1.0, opt, integrator=integ, layout=1)
            side.synchronize(); d = time.perf_counter() - c0
            best = d if best is None or d < best else best
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(3):
        nn.solveODE(f, y0, [0.0, 1.0], opt, integrator=integ, layout=1)
    e1.record(); torch.cuda.synchronize()
    res[name] = dict(us_per_iteration=round(best * 1e6 / iters, 2), iterations=iters, fused_ms=round(e0.elapsed_time(e1) / 3, 3), equal_to_fused=bool(torch.equal(ys, yf[-1])))
n = 1_000_000
y16 = torch.from_numpy(1.0 + np.arange(16)[None, :] / 16 + ((np.arange(n) % 1024) * 2.0 ** -20)[:, None]).to(dev)
run("builtin_ring16_N1e6", nn.Rhs.ring(0.1), y16)
run("source_ring16_N1e6", nn.Rhs.custom(16, RING, keys=("c",), defaults={"c": 0.1}, name="ring16_src", per_component=True), y16)
run("source_ring16_halo01_N1e6", nn.Rhs.custom(16, RING, keys=("c",), defaults={"c": 0.1}, name="ring16_src_halo", per_component=True, halo=(0, 1)), y16)
for dim, m in ((24, 500_000), (64, 262_144), (100, 131_072)):
    f = nn.Rhs.custom(dim, HEAT, keys=("kappa",), defaults={"kappa": 0.4}, name=f"heat{dim}_stream", per_component=True)
    yh = torch.rand(m, dim, dtype=torch.float64, device=dev)
    run(f"source_heat{dim}_N{m}", f, yh)
    run(f"source_heat{dim}_halo11_N{m}", nn.Rhs.custom(dim, HEAT, keys=("kappa",), defaults={"kappa": 0.4}, name=f"heat{dim}_stream_halo", per_component=True, halo=(1, 1)), yh)
print(json.dumps(res, indent=1))
