#!/usr/bin/env python3
"""Inspect individual seeds of the randomised parity sweep."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import numericalnim_amd as nn
from oracle import oracle as O
import test_gpu_fuzz_parity as T
dev = torch.device("cuda:0")
for seed in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(1000 + seed)
    kind, dim, params, integ, ts, opt, n, layout = T._draw(rng, nn)
    y0 = rng.uniform(-1.5, 1.5, (n, dim)) + (np.array([0.0, 0.0, 15.0]) if kind == 2 else 0.0)
    y0l = np.ascontiguousarray(y0 if layout == 1 else y0.T)
    if dim == 1:
        y0l, layout = y0[:, 0].copy(), 0
    f = nn.Rhs(kind, T.KEYS.get(kind, ()), dict(zip(T.KEYS.get(kind, ()), params)))
    t, y, cnt = nn.solveODE(f, torch.from_numpy(y0l).to(dev), ts, nn.newODEoptions(**opt), integrator=integ, layout=layout, return_counts=True)
    ref = O.solve_ode_batch(kind, params, y0l, n, 0 if dim == 1 else dim, ts, O.new_options(**opt), integ, layout=layout, n_threads=8)
    got = y.cpu().numpy().reshape(ref["y"].shape)
    d = np.abs(got - ref["y"]); d[np.isnan(d)] = 0
    steps, rej = cnt["steps"].cpu().numpy(), cnt["rejected"].cpu().numpy()
    ivp_axis = 2 if (layout == 0 or dim == 1) else 1
    per_ivp = d.max(axis=tuple(a for a in range(d.ndim) if a != ivp_axis)) if d.ndim == 3 else d.max(axis=0)
    worst = int(np.argmax(per_ivp))
    print(f"seed {seed}: {integ} kind={kind} dim={dim} n={n} layout={layout} ts={ts} opt={ {k: float('%.3g' % v) for k, v in opt.items()} }")
    print(f"   max|diff|={d.max():.3e}  ivps over 1e-6: {(per_ivp > 1e-6).sum()}  step mismatches: {(steps != ref['steps']).sum()}  rej mismatches: {(rej != ref['rejected']).sum()}")
    print(f"   worst ivp {worst}: steps gpu/ref {steps[worst]}/{ref['steps'][worst]} rej {rej[worst]}/{ref['rejected'][worst]}  |y| max {np.nanmax(np.abs(ref['y'])):.3g}")
    if d.max() > 1e-6 and np.isfinite(d.max()):
        sl = (slice(None), slice(None), worst) if ivp_axis == 2 else (slice(None), worst, slice(None))
        print("   t  =", t)
        print("   gpu=", got[sl].tolist(), "ny", int(cnt["ny"][worst]))
        print("   ref=", ref["y"][sl].tolist(), "ny", int(ref["ny"][worst]))
