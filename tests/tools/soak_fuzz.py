#!/usr/bin/env python3
"""Soak: the randomised parity sweep of tests/test_gpu_fuzz_parity.py over many more seeds, COUNTING instead of asserting.
Per integrator: IVPs integrated, IVPs whose step path (accepted, rejected, emitted rows) differs from the oracle's,
IVPs on the same path that are not bit-identical, IVPs outside 1e-6.  Writes the table as JSON (profiles/r02_fuzz_soak.json).

usage: python tests/tools/soak_fuzz.py [n_seeds] [out.json]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import numericalnim_amd as nn
from oracle import oracle as O
import test_gpu_fuzz_parity as T

dev = torch.device("cuda:0")
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "r02_fuzz_soak.json")
KEYS = T.KEYS
tab = {}
seed0 = int(os.environ.get("NNHIP_SOAK_SEED0", "120"))
for seed in range(seed0, seed0 + n_seeds):
    rng = np.random.default_rng(1000 + seed)
    kind, dim, params, integ, ts, opt, n, layout = T._draw(rng, nn)
    y0 = rng.uniform(-1.5, 1.5, (n, dim)) + (np.array([0.0, 0.0, 15.0]) if kind == 2 else 0.0)
    y0l = np.ascontiguousarray(y0 if layout == 1 else y0.T)
    if dim == 1:
        y0l, layout = y0[:, 0].copy(), 0
    f = nn.Rhs(kind, KEYS.get(kind, ()), dict(zip(KEYS.get(kind, ()), params)))
    t, y, cnt = nn.solveODE(f, torch.from_numpy(y0l).to(dev), ts, nn.newODEoptions(**opt), integrator=integ, layout=layout, return_counts=True)
    ref = O.solve_ode_batch(kind, params, y0l, n, 0 if dim == 1 else dim, ts, O.new_options(**opt), integ, layout=layout, n_threads=8)
    got = y.cpu().numpy().reshape(ref["y"].shape)
    steps, rej, ny = cnt["steps"].cpu().numpy(), cnt["rejected"].cpu().numpy(), cnt["ny"].cpu().numpy()
    ivp_axis = got.ndim - 1 if (layout == 0 or dim == 1) else 1
    other = tuple(a for a in range(got.ndim) if a != ivp_axis)
    same_path = (steps == ref["steps"]) & (rej == ref["rejected"]) & (ny == ref["ny"])
    with np.errstate(invalid="ignore", over="ignore"):
        neq = ~((got == ref["y"]) | (np.isnan(got) & np.isnan(ref["y"])))
        far = ~(np.abs(got - ref["y"]) <= 1e-6) & np.isfinite(ref["y"]) & np.isfinite(got) & (np.abs(ref["y"]) <= 1e8)
    neq_ivp = neq.any(axis=other) if got.ndim > 1 else neq
    far_ivp = far.any(axis=other) if got.ndim > 1 else far
    r = tab.setdefault(integ, dict(cases=0, ivps=0, attempted_steps=0, path_divergent=0, same_path_not_bit_identical=0, outside_1e6=0))
    r["cases"] += 1
    r["ivps"] += int(n)
    r["attempted_steps"] += int(ref["steps"].sum() + ref["rejected"].sum())
    r["path_divergent"] += int((~same_path).sum())
    r["same_path_not_bit_identical"] += int((neq_ivp & same_path).sum())
    r["outside_1e6"] += int(far_ivp.sum())
tot = {k: sum(r[k] for r in tab.values()) for k in next(iter(tab.values()))}
res = {"seeds": [seed0, seed0 + n_seeds], "per_integrator": dict(sorted(tab.items())), "total": tot,
       "adaptive_total": {k: sum(r[k] for i, r in tab.items() if i not in nn.fixedODE) for k in tot}}
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res["total"]), json.dumps(res["adaptive_total"]))
