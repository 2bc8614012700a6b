#!/usr/bin/env python3
"""A/B of the vectorised fixed-step streaming kernel (tuning knob fixed_vec_ipl: 0 = one IVP per lane (round-1 kernel), 2 IVPs per lane, with and
without the non-temporal hint): RK4 over 1e7 Lorenz IVPs (SoA), per-IVP (t, dt) arrays as in profiles/r01_bench_extra.json, and uniform (t, dt)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import numericalnim_amd as nn
dev = torch.device("cuda:0")
L = nn._lib.lib()
res = {}
for n in (1_000_000, 10_000_000):
    y0 = torch.from_numpy(np.stack([1.0 + (np.arange(n) % 1024) * 2.0 ** -20, np.ones(n), np.ones(n)])).to(dev)
    out = torch.empty_like(y0)
    tdev = torch.zeros(n, dtype=torch.float64, device=dev)
    dtdev = torch.full((n,), 1e-3, dtype=torch.float64, device=dev)
    opt = nn.newODEoptions(dt=1e-3)
    for ipl in (0, 2, -2, 2, -2, 0):
        L.nnhip_tune_set(b"fixed_vec_ipl", abs(ipl)); L.nnhip_tune_set(b"adv_nontemporal", 1 if ipl < 0 else 0)
        for name, tt, dd, b in (("per_ivp_t_dt", tdev, dtdev, 8 * (2 * 3 + 2)), ("uniform", 0.0, 1e-3, 8 * 2 * 3)):
            for integ in ("rk4", "kutta4"):
                ts = []
                for r in range(6):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        nn.integratorStep(nn.Rhs.lorenz(), tt, y0, None, dd, opt, integrator=integ, out=out)
                    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e-4)
                s = sorted(ts)[1]
                res.setdefault(f"N{n:.0e}_{integ}_{name}", {}).setdefault({0: "one_ivp_per_lane", 2: "vec2", -2: "vec2_nontemporal"}[ipl], []).append(round(b * n / s / 1e9))
L.nnhip_tune_set(b"fixed_vec_ipl", 2); L.nnhip_tune_set(b"adv_nontemporal", -1)
print(json.dumps(res, indent=1))
