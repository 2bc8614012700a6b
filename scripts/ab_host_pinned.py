#!/usr/bin/env python3
"""PCIe-inclusive host-pointer solve (nnhip_ode_solve_batch_f64) on C2 (fused RK4, 1e7 IVPs: 80 MB in, 160 MB out): what the
caller's buffers cost.  (a) fresh pageable output every call (what a naive host wrapper does: first-touch page faults inside the
copy), (b) pageable buffers reused, (c) page-locked buffers reused (hipHostMalloc through torch's pinned allocator)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import numericalnim_amd as nn  # noqa: E402
from numericalnim_amd import distributed as nd  # noqa: E402

L = nn._lib.lib()
n = 10_000_000
opt = nn.newODEoptions(dt=2.0 ** -10)
ts = np.array([0.0, 1000 * 2.0 ** -10])
dp = C.POINTER(C.c_double)


def call(y0, out):
    t_out = np.empty(2)
    st = nn.ode.Stats()
    rc = L.nnhip_ode_solve_batch_f64(C.byref(opt), 0, 0, None, 0, y0.ctypes.data_as(dp), n, 1, 0, ts.ctypes.data_as(dp), 2, t_out.ctypes.data_as(dp),
                                     out.ctypes.data_as(dp), None, None, None, 0, C.byref(st), 0)
    assert rc == 0, nn._lib.last_error()
    return st.kernel_ms


def timed(fn, reps=5):
    fn()
    xs = []
    for _ in range(reps):
        c0 = time.perf_counter()
        fn()
        xs.append((time.perf_counter() - c0) * 1e3)
    return sorted(xs)[len(xs) // 2]


res = {}
y0 = nd.c2_y0_numpy(0, n)
res["fresh_pageable_output_ms"] = timed(lambda: call(y0, np.empty((2, n))))
out = np.zeros((2, n))
res["reused_pageable_ms"] = timed(lambda: call(y0, out))
ref = out.copy()
y0p = torch.empty(n, dtype=torch.float64).pin_memory()
y0p.numpy()[:] = y0
outp = torch.zeros(2, n, dtype=torch.float64).pin_memory()
res["reused_pinned_ms"] = timed(lambda: call(y0p.numpy(), outp.numpy()))
res["bitwise_equal"] = bool(np.array_equal(outp.numpy(), ref))
st = call(y0p.numpy(), outp.numpy())
res["kernel_ms"] = st
for ch in (2, 4, 8):
    L.nnhip_tune_set(b"host_chunks", ch)
    res[f"reused_pageable_chunks{ch}_ms"] = timed(lambda: call(y0, out))
    res[f"reused_pinned_chunks{ch}_ms"] = timed(lambda: call(y0p.numpy(), outp.numpy()))
    assert np.array_equal(outp.numpy(), ref) and np.array_equal(out, ref)
L.nnhip_tune_set(b"host_chunks", 0)
res["reused_pinned_auto_again_ms"] = timed(lambda: call(y0p.numpy(), outp.numpy()), reps=9)
res["reused_pageable_auto_again_ms"] = timed(lambda: call(y0, out), reps=9)
res["traj_steps_per_s_pinned"] = n * 1000 / (res["reused_pinned_ms"] * 1e-3)
print(json.dumps(res))
