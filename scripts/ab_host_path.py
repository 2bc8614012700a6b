#!/usr/bin/env python3
"""PCIe-inclusive host-pointer solve (nnhip_ode_solve_batch_f64): chunk count x page-locking A/B on C2 (fused RK4)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import numericalnim_amd as nn
from numericalnim_amd import distributed as nd
L = nn._lib.lib()
n = 10_000_000
y0 = nd.c2_y0_numpy(0, n)
opt = nn.newODEoptions(dt=2.0 ** -10)
res = {}
ref = None
for reg in (0, 1):
    for ch in (1, 2, 4):
        L.nnhip_tune_set(b"host_register", reg)
        L.nnhip_tune_set(b"host_chunks", ch)
        ts, ks = [], []
        for r in range(4):
            st = nn.ode.Stats()
            c0 = time.perf_counter()
            t, y = nn.solveODE(nn.Rhs.neg_y(), y0, [0.0, 1000 * 2.0 ** -10], opt, integrator="rk4", stats=st)
            ts.append(time.perf_counter() - c0); ks.append(st.kernel_ms)
        if ref is None:
            ref = y.copy()
        res[f"register{reg}_chunks{ch}"] = dict(wall_ms=sorted(ts[1:])[1] * 1e3, kernel_ms=sorted(ks[1:])[1], equal=bool(np.array_equal(y, ref)))
L.nnhip_tune_set(b"host_register", 0); L.nnhip_tune_set(b"host_chunks", 0)
print(json.dumps(res, indent=1))
