#!/usr/bin/env python3
"""How often does the device controller factor min(4, max(0.125, 0.9 * pow(1/error, 1/order))) (ode.nim:71, 537) equal the
reference's bit for bit?  The reference is Nim -> C `pow` -> glibc's libm, so the comparison is against libm's pow called
through ctypes — NOT numpy.power, whose SIMD implementation misrounds ~5 % of these arguments and is not what the
reference runs.  A subsample is also compared with the correctly rounded value (decimal, 60 digits)."""
import ctypes as C
import ctypes.util
import os
import sys
from decimal import Decimal, getcontext

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import numericalnim_amd as nn

L = nn._lib.lib()
libm = C.CDLL(ctypes.util.find_library("m"))
libm.pow.restype = C.c_double
libm.pow.argtypes = [C.c_double, C.c_double]
getcontext().prec = 60
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
err = 10 ** rng.uniform(-3, 3, 400_000)
e = torch.from_numpy(err).to(dev)
out = torch.empty_like(e)
for order in (2, 3, 5, 6):
    L.nnhip_ode_controller_factor_f64_dev(order, e.data_ptr(), out.data_ptr(), e.numel(), None)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    y = 1.0 / order
    root = np.array([libm.pow(1.0 / v, y) for v in err])
    ref = np.minimum(4.0, np.maximum(0.125, 0.9 * root))
    un = (ref != 4.0) & (ref != 0.125)
    idx = np.flatnonzero(un)[:20000]
    yd = Decimal(y)
    exact = np.array([float((Decimal(float(1.0 / err[i])).ln() * yd).exp()) for i in idx])  # correctly rounded pow(1/err, fl(1/order))
    cr = np.minimum(4.0, np.maximum(0.125, 0.9 * exact))
    print(f"order {order}: equal to glibc pow on {100.0 * (got[un] == ref[un]).mean():.3f} % of {un.sum()} unclamped points "
          f"(max {np.max(np.abs(got - ref) / np.spacing(ref)):.0f} ulp); vs correctly rounded on {len(idx)} points: "
          f"device {100.0 * (got[idx] == cr).mean():.3f} %, glibc {100.0 * (ref[idx] == cr).mean():.3f} %")
