#!/usr/bin/env python3
"""A/B: dim-16 fused solve, lanes-per-system (LDS) vs register-resident thread-per-IVP ("dim16_variant"), both layouts."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import numericalnim_amd as nn
L = nn._lib.lib()
dev = torch.device("cuda:0")
n = 1_000_000
s_idx = np.arange(n)
yaos = torch.from_numpy(1.0 + np.arange(16)[None, :] / 16 + ((s_idx % 1024) * 2.0 ** -20)[:, None]).to(dev)
ysoa = yaos.t().contiguous()
tight = dict(absTol=1e-10, relTol=1e-10, dtMin=1e-6, dtMax=1e-1)
res = {}
outs = {}
for integ in ("tsit54", "dopri54", "rk4", "vern65"):
    for oname, kw in (("default", dict(dt=1e-2)), ("tight", dict(dt=1e-2, **tight))):
        opt = nn.newODEoptions(**kw)
        for wide in (2, 0, 1):
            for lname, lay, y0 in (("aos", 1, yaos), ("soa", 0, ysoa)):
                L.nnhip_tune_set(b"dim16_variant", wide)
                ts = []
                for r in range(4):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    t, y = nn.solveODE(nn.Rhs.ring(0.1), y0, [0.0, 1.0], opt, integrator=integ, layout=lay)
                    e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                key = f"{integ}_{oname}_{ {2: 'lps1', 0: 'default', 1: 'tpi16'}[wide]}_{lname}"
                res[key] = sorted(ts[1:])[1]
                yy = y[-1] if lay == 1 else y[-1].t()
                outs.setdefault((integ, oname), []).append(yy.contiguous())
for k, v in outs.items():
    same = all(torch.equal(v[0], x) for x in v[1:])
    res[f"{k[0]}_{k[1]}_all_variants_bitwise_equal"] = same
print(json.dumps(res, indent=1))
