#!/usr/bin/env python3
"""N separate reference calls in one launch (nnhip_ode_solve_batch_calls_f64_dev: every IVP its own tEnd), spans uniform in (0, 10]: the calls in the caller's
order (knob calls_bin = 0) against the longest spans first, binned by span below the boundary (default).  1e6 Van der Pol calls, DOPRI54 and RK4."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import numericalnim_amd as nn
dev = torch.device("cuda:0"); n = 1_000_000
L = nn._lib.lib()
rng = np.random.default_rng(2)
y0 = torch.from_numpy(np.stack([rng.uniform(1.5, 2.5, n), np.zeros(n)])).to(dev)
t_end = torch.from_numpy(rng.uniform(0.05, 10.0, n)).to(dev)
t_end_sorted = torch.sort(t_end, descending=True).values
opt = nn.newODEoptions(absTol=1e-6, relTol=1e-6, dtMin=1e-9, dtMax=1.0, dt=1e-2)
res = {}
for integ in ("dopri54", "rk4"):
    cases = {"callers_order": (0, t_end), "binned_by_span": (1, t_end), "presorted_by_caller_longest_first": (0, t_end_sorted)}
    times = {k: [] for k in cases}
    outs = {}
    for r in range(9):
        for k, (knob, te) in cases.items():
            L.nnhip_tune_set(b"calls_bin", knob)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
            out = nn.solveODEPerIvpEnd(nn.Rhs.vanderpol(2.0), y0, te, opt, integrator=integ); e1.record()
            torch.cuda.synchronize(); times[k].append(e0.elapsed_time(e1)); outs[k] = out
    L.nnhip_tune_set(b"calls_bin", 1)
    st = outs["callers_order"][1]["steps"].double()
    w = st[: n // 64 * 64].reshape(-1, 64)
    res[integ] = {k: sorted(v[2:])[len(v[2:]) // 2] for k, v in times.items()}
    res[integ]["lane_utilisation_callers_order"] = float(w.mean() / w.max(dim=1).values.mean())
    res[integ]["bit_identical"] = bool(torch.equal(outs["callers_order"][0], outs["binned_by_span"][0]) and all(torch.equal(outs["callers_order"][1][k], outs["binned_by_span"][1][k]) for k in outs["callers_order"][1]))
print(json.dumps(res, indent=1))
