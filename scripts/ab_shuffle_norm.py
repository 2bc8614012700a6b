#!/usr/bin/env python3
"""A/B: error norm of the lanes-per-system kernels — ordered LDS sum (reference's left-to-right order, default) vs
wavefront-shuffle butterfly — on 1e6 16-dim systems."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import numericalnim_amd as nn
L = nn._lib.lib()
dev = torch.device("cuda:0")
n = 1_000_000
y0 = torch.from_numpy(1.0 + np.arange(16)[None, :] / 16 + ((np.arange(n) % 1024) * 2.0 ** -20)[:, None]).to(dev)
tight = dict(absTol=1e-10, relTol=1e-10, dtMin=1e-6, dtMax=1e-1)
res = {}
for integ in ("tsit54", "dopri54"):
    for oname, kw in (("default", {}), ("tight", tight)):
        outs = {}
        for name, knob in (("lds_ordered_4x4", 0), ("shuffle_4x4", 3), ("lds_ordered_16x1", 2), ("shuffle_16x1", 4)):
            L.nnhip_tune_set(b"dim16_variant", knob)
            ts = []
            for r in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); t, y = nn.solveODE(nn.Rhs.ring(0.1), y0, [0.0, 1.0], nn.newODEoptions(**kw), integrator=integ, layout=1); e1.record()
                torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
            outs[name] = y[-1].clone()
            res[f"{integ}_{oname}_{name}_ms"] = sorted(ts[1:])[1]
        res[f"{integ}_{oname}_shuffle_max_abs_diff_vs_ordered"] = float((outs["shuffle_4x4"] - outs["lds_ordered_4x4"]).abs().max())
L.nnhip_tune_set(b"dim16_variant", 0)
print(json.dumps(res, indent=1))
