#!/usr/bin/env python3
"""How often does the device controller factor equal glibc's (the reference's libm) bit for bit?"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import numericalnim_amd as nn
L = nn._lib.lib()
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
err = 10 ** rng.uniform(-3, 3, 2_000_000)
e = torch.from_numpy(err).to(dev); out = torch.empty_like(e)
for order in (2, 3, 5, 6):
    L.nnhip_ode_controller_factor_f64_dev(order, e.data_ptr(), out.data_ptr(), e.numel(), None); torch.cuda.synchronize()
    got = out.cpu().numpy()
    ref = np.minimum(4.0, np.maximum(0.125, 0.9 * np.power(1.0 / err, 1.0 / order)))   # numpy -> glibc pow, same expression order
    un = (ref != 4.0) & (ref != 0.125)
    print(f"order {order}: bitwise equal to glibc on {100.0 * (got[un] == ref[un]).mean():.3f} % of {un.sum()} unclamped points; max ulp diff {np.max(np.abs(got - ref) / np.spacing(ref)):.0f}")
