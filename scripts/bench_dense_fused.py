import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import numericalnim_amd as nn
dev = torch.device("cuda:0")
def timed(fn, reps=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts)//2]
out = {}
n = 1_000_000
y0 = torch.from_numpy(np.stack([1.0 + (np.arange(n) % 1024) * 2.0 ** -20, np.ones(n), np.ones(n)])).to(dev)
y16 = torch.from_numpy(1.0 + np.arange(16)[None, :] / 16 + ((np.arange(n) % 1024) * 2.0 ** -20)[:, None]).to(dev)
tight = dict(absTol=1e-10, relTol=1e-10, dtMin=1e-6, dtMax=1e-1)
for name, kw in (("default", {}), ("tight", tight)):
    opt = nn.newODEoptions(**kw)
    for integ in ("dopri54", "tsit54", "rk4"):
        o = nn.newODEoptions(dt=1e-3) if integ == "rk4" else opt
        for nt in (2, 3, 11, 101):
            ts = np.linspace(0.0, 1.0, nt)
            out[f"lorenz_{integ}_{name}_nt{nt}"] = timed(lambda: nn.solveODE(nn.Rhs.lorenz(), y0, ts, o, integrator=integ))
            if nt <= 11:
                out[f"ring16_{integ}_{name}_nt{nt}"] = timed(lambda: nn.solveODE(nn.Rhs.ring(0.1), y16, ts, o, integrator=integ, layout=1))
print(json.dumps(out, indent=1))
