#!/usr/bin/env python3
"""Secondary measurements on one MI355X (run via gpurun): BASELINE configs C3 / C4 (fused adaptive kernels), the
step-streaming adaptive kernels, the fused RK4 kernel, and the PCIe-inclusive host-pointer entry.  Prints one JSON
object; the headline number stays bench.py's."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import numericalnim_amd as nn  # noqa: E402
from numericalnim_amd import distributed as nd  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, reps=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return sorted(ts)[len(ts) // 2], r


def main():
    out = {}
    tight = dict(absTol=1e-10, relTol=1e-10, dtMin=1e-6, dtMax=1e-1)
    n = 1_000_000
    # ---- C3: DOPRI54, Lorenz, 1e6 IVPs, SoA -------------------------------------------------------------
    y0 = torch.from_numpy(np.stack([1.0 + (np.arange(n) % 1024) * 2.0 ** -20, np.ones(n), np.ones(n)])).to(dev)
    for integ in ("dopri54", "tsit54"):
        for name, kw in (("default", {}), ("tight", tight)):
            opt = nn.newODEoptions(**kw)
            s, r = timed(lambda: nn.solveODE(nn.Rhs.lorenz(), y0, [0.0, 1.0], opt, integrator=integ, return_counts=True))
            steps = int(r[2]["steps"].sum()) + int(r[2]["rejected"].sum())
            out[f"C3_lorenz_{integ}_{name}"] = dict(ms=s * 1e3, ivps_per_s=n / s, attempted_steps=steps, attempted_steps_per_s=steps / s)
    # ---- C4: Tsit54, 16-dim ring, 1e6 systems, AoS, lanes-per-system kernel ------------------------------------
    s_idx = np.arange(n)
    y16 = torch.from_numpy(1.0 + np.arange(16)[None, :] / 16 + ((s_idx % 1024) * 2.0 ** -20)[:, None]).to(dev)
    for integ in ("tsit54", "dopri54"):
        for name, kw in (("default", {}), ("tight", tight)):
            opt = nn.newODEoptions(**kw)
            s, r = timed(lambda: nn.solveODE(nn.Rhs.ring(0.1), y16, [0.0, 1.0], opt, integrator=integ, layout=1, return_counts=True))
            steps = int(r[2]["steps"].sum()) + int(r[2]["rejected"].sum())
            out[f"C4_ring16_{integ}_{name}"] = dict(ms=s * 1e3, systems_per_s=n / s, attempted_steps=steps, attempted_steps_per_s=steps / s)
    # ---- step-streaming adaptive kernels: 8*(4d+5) B per attempted step ---------------------------------
    opt = nn.newODEoptions(**tight)
    fs = torch.empty_like(y0)
    L = nn._lib.lib()
    tdev = torch.zeros(n, dtype=torch.float64, device=dev)
    dtdev = torch.full((n,), 1e-3, dtype=torch.float64, device=dev)
    import ctypes as C
    p = np.array([10.0, 28.0, 8.0 / 3.0])
    L.nnhip_ode_rhs_batch_f64_dev(2, p.ctypes.data_as(C.POINTER(C.c_double)), 3, n, 3, 0, 0.0, y0.data_ptr(), fs.data_ptr(), None)
    # one IntegratorProc call = one nnhip_ode_step_batch_f64_dev call; 20 back-to-back calls through the C ABI with preallocated outputs
    # (the Python convenience wrapper allocates five tensors per call, which is host time, not the entry's)
    ynew, fsn, dtu, er = (torch.empty_like(y0), torch.empty_like(y0), torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev))
    stream = torch.cuda.current_stream().cuda_stream
    for integ in ("dopri54", "tsit54"):
        iid = nn.ode.integrator_id(integ)
        def calls(k=20):
            for _ in range(k):
                rc = L.nnhip_ode_step_batch_f64_dev(C.byref(opt), iid, 2, p.ctypes.data_as(C.POINTER(C.c_double)), 3, n, 3, 0, tdev.data_ptr(), 0.0, dtdev.data_ptr(), 0.0,
                                                    y0.data_ptr(), fs.data_ptr(), ynew.data_ptr(), fsn.data_ptr(), dtu.data_ptr(), er.data_ptr(), 0, stream)
                assert rc == 0
        s, _ = timed(calls, reps=5)
        s /= 20
        out[f"step_stream_lorenz_{integ}"] = dict(us=s * 1e6, GBps=8 * (4 * 3 + 5) * n / s / 1e9, algorithmic_bytes=8 * (4 * 3 + 5) * n, calls_per_timing=20)
        s1, _ = timed(lambda: nn.integratorStep(nn.Rhs.lorenz(), tdev, y0, fs, dtdev, opt, integrator=integ), reps=5)
        out[f"step_stream_lorenz_{integ}"]["us_single_call_through_python_wrapper"] = s1 * 1e6
    fs16 = torch.empty_like(y16)
    pr = np.array([0.1])
    L.nnhip_ode_rhs_batch_f64_dev(3, pr.ctypes.data_as(C.POINTER(C.c_double)), 1, n, 16, 1, 0.0, y16.data_ptr(), fs16.data_ptr(), None) if False else None
    # ---- fused RK4 (C2 shape): FP64-VALU bound ------------------------------------------------------------
    n2 = 10_000_000
    y2 = nd.c2_y0_torch(0, n2, dev)
    o2 = nn.newODEoptions(dt=2.0 ** -10)
    s, _ = timed(lambda: nn.solveODE(nn.Rhs.neg_y(), y2, [0.0, 1000 * 2.0 ** -10], o2, integrator="rk4"))
    out["C2_fused_rk4"] = dict(ms=s * 1e3, traj_steps_per_s=n2 * 1000 / s, useful_fp64_flop_per_s=16 * n2 * 1000 / s)
    # ---- PCIe-inclusive: host-pointer entry (alloc + H2D + kernel + D2H) -------------------------------------------
    y2h = nd.c2_y0_numpy(0, n2)
    st = nn.ode.Stats()
    nn.solveODE(nn.Rhs.neg_y(), y2h[:1000], [0.0, 1.0], o2, integrator="rk4")
    c0 = time.perf_counter()
    nn.solveODE(nn.Rhs.neg_y(), y2h, [0.0, 1000 * 2.0 ** -10], o2, integrator="rk4", stats=st)
    c1 = time.perf_counter()
    out["C2_fused_rk4_host_pointers"] = dict(wall_ms=(c1 - c0) * 1e3, kernel_ms=st.kernel_ms, traj_steps_per_s_pcie_inclusive=n2 * 1000 / (c1 - c0))
    # ---- opt-in FMA-contracted instantiations (tuning knob fp_contract; not bit-exact, inside the 1e-10 / 1e-6 tolerances) ----
    L.nnhip_tune_set(b"fp_contract", 1)
    s, _ = timed(lambda: nn.solveODE(nn.Rhs.neg_y(), y2, [0.0, 1000 * 2.0 ** -10], o2, integrator="rk4"))
    out["fp_contract_C2_fused_rk4"] = dict(ms=s * 1e3, traj_steps_per_s=n2 * 1000 / s)
    s, _ = timed(lambda: nn.solveODE(nn.Rhs.lorenz(), y0, [0.0, 1.0], nn.newODEoptions(), integrator="dopri54"))
    out["fp_contract_C3_lorenz_dopri54_default"] = dict(ms=s * 1e3)
    s, _ = timed(lambda: nn.solveODE(nn.Rhs.ring(0.1), y16, [0.0, 1.0], nn.newODEoptions(), integrator="tsit54", layout=1))
    out["fp_contract_C4_ring16_tsit54_default"] = dict(ms=s * 1e3)
    L.nnhip_tune_set(b"fp_contract", 0)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
