#!/usr/bin/env python3
"""Times the function-argument forms of cumtrapz / cumsimpson (integrate.nim:138-175, 377-400) on the device:
N parameter sets of the integrand a x^2 + b x + c, X = linspace(0, 1, 11), grid spacing dx.  The kernels are FP64-VALU bound
(one integrand evaluation + the rule per grid point per item; nothing but the 11 requested rows reaches HBM).
Writes one JSON line."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=float, default=1e6)
    ap.add_argument("--dx", type=float, default=1e-4)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch
    import numericalnim_amd as nn
    dev = torch.device("cuda", 0)
    N = int(args.n)
    sw = torch.rand(3, N, dtype=torch.float64, device=dev) * 2 - 1
    f = nn.Rhs.custom(1, "dy[0] = ((p[0] * t + p[1]) * t) * 1.0 + p[2];", keys=("a", "b", "c"), defaults={"a": 0.0, "b": 0.0, "c": 0.0}, name="poly_bench")
    X = np.linspace(0.0, 1.0, 11)
    out = {"N": N, "dx": args.dx, "n_x": len(X)}
    xs = torch.tensor(X, device=dev)[:, None]
    exact = sw[0] / 3 * xs ** 3 + sw[1] / 2 * xs ** 2 + sw[2] * xs
    for name, fn in (("cumtrapz", nn.cumtrapz), ("cumsimpson", nn.cumsimpson)):
        got = fn(f, X, dx=args.dx, sweep=sw)  # compiles the user integrand on first use
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            got = fn(f, X, dx=args.dx, sweep=sw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        grid_points = round(1.0 / args.dx) + 1  # the march stops at the last requested x
        out[name] = {"ms": ms, "grid_points_per_item": grid_points, "integrand_evals_per_s": N * grid_points / (ms * 1e-3),
                     "max_abs_err_vs_closed_form": float((got - exact).abs().max())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
