#!/usr/bin/env python3
"""cumsimpson(f, X, ctx, dx) / cumtrapz(f, X, ctx, dx) of numericalnim (integrate.nim:138-175, 377-400) for 100 000 parameter sets
at once: the cumulative distribution function of a Gaussian for many (mu, sigma), i.e. the running integral of its density.
The integrand is HIP C++ source compiled at run time; the batch axis is the parameter sweep."""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import numericalnim_amd as nn  # noqa: E402

dev = torch.device("cuda", 0)
N = 100_000
mu = torch.linspace(-1.0, 1.0, N, dtype=torch.float64, device=dev)
sigma = torch.linspace(0.5, 2.0, N, dtype=torch.float64, device=dev)
pdf = nn.Rhs.custom(1, "const double z = (t - p[0]) / p[1]; dy[0] = exp(-0.5 * z * z) / (p[1] * 2.5066282746310002);",
                    keys=("mu", "sigma"), defaults={"mu": 0.0, "sigma": 1.0}, name="gauss_pdf")
X = np.linspace(-6.0, 6.0, 25)
cdf = nn.cumsimpson(pdf, X, dx=1e-3, sweep=torch.stack([mu, sigma]))            # [25, N]: integral from X[0] to X[j]
ref = 0.5 * (torch.erf((torch.tensor(X, device=dev)[:, None] - mu) / (sigma * math.sqrt(2.0)))
             - torch.erf((X[0] - mu) / (sigma * math.sqrt(2.0))))
print("rows x parameter sets:", tuple(cdf.shape), " max |cumsimpson - erf formula| =", float((cdf - ref).abs().max()))
trap = nn.cumtrapz(pdf, X, dx=1e-3, sweep=torch.stack([mu, sigma]))
print("cumtrapz, same grid:   max error =", float((trap - ref).abs().max()))
