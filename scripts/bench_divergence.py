#!/usr/bin/env python3
"""Wavefront divergence in the fused adaptive kernels: a Van der Pol sweep over mu (step counts differ ~2x across IVPs, and lanes
disagree on accept / reject).  The batch in random order — as a caller would hand it over — integrated (a) as is, (b) through the
C ABI's divergence binning with the caller's key (nnhip_ode_solve_batch_sorted_f64_dev, sort_key = mu), (c) in its automatic
two-pass mode (probe solve + device argsort), and (d) pre-sorted by the caller (the bound: no indirection, no sort); (b) and (c) with the
order array followed inside the solve kernel (default) and with the batch physically reordered around the solve (knob sort_copy), each
with and without the per-IVP counters (rows, steps, rejected) written.
All cases are timed INTERLEAVED (one call of each per round, 11 rounds after a warm-up, median of the last 9): timed one case after the
other, the cases were ranked by their position in the file as much as by what they do (clock and power state drift by several percent)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import numericalnim_amd as nn
dev = torch.device("cuda:0"); n = 1_000_000
L = nn._lib.lib()
rng = np.random.default_rng(0)
mu = rng.uniform(0.1, 20.0, n)
y0 = torch.from_numpy(np.stack([np.full(n, 2.0), np.zeros(n)])).to(dev)
opt = nn.newODEoptions(absTol=1e-6, relTol=1e-6, dtMin=1e-9, dtMax=1.0)
sw = torch.from_numpy(mu[None, :].copy()).to(dev)
sw_sorted = torch.from_numpy(np.sort(mu)[None, :].copy()).to(dev)
mu2 = 100.0 + rng.random(n)   # a narrow-range key (float32 image resolves it; round 2's 16-bit keys put the whole batch into two bins)
sw2 = torch.from_numpy(mu2[None, :].copy()).to(dev)
sw_desc = torch.from_numpy(np.sort(mu)[None, ::-1].copy()).to(dev)
neg_mu = (-sw[0]).contiguous()


# a batch whose step sizes change LATE: relaxation oscillations (mu in [5, 50]) all starting on the slow branch — the fast jump, where the steps are spent,
# comes at t ~ 0.8 mu, long after the 8 steps of the probe
mu3 = rng.uniform(5.0, 50.0, n)
sw3 = torch.from_numpy(mu3[None, :].copy()).to(dev)


def solve(sweep, sort_by=None, counts=False, copy=0, tend=10.0, resume=0, auto_key=1, rebin=0):
    L.nnhip_tune_set(b"sort_auto_key", auto_key)
    L.nnhip_tune_set(b"sort_rebin_steps", rebin)
    L.nnhip_tune_set(b"sort_copy", copy)
    L.nnhip_tune_set(b"sort_resume", resume)
    return nn.solveODE(nn.Rhs.vanderpol(), y0, [0.0, tend], opt, integrator="dopri54", sweep=sweep, sort_by=sort_by, return_counts=counts)


cases = {
    "random_order": lambda: solve(sw, None, True),
    "random_order_sort_key_mu": lambda: solve(sw, sw[0], True),
    "random_order_auto_probe": lambda: solve(sw, "auto", True),
    "random_order_auto_probe_resuming": lambda: solve(sw, "auto", True, 0, 10.0, 1),   # knob sort_resume: the sorted pass continues from the probe's (t, dt, y)
    "random_order_auto_probe_by_progress": lambda: solve(sw, "auto", True, 0, 10.0, 0, 0),     # knob sort_auto_key 0: round 3's key
    "random_order_auto_probe_rebin_after_32": lambda: solve(sw, "auto", True, 0, 10.0, 1, 1, 32),
    "random_order_auto_probe_rebin_after_64": lambda: solve(sw, "auto", True, 0, 10.0, 1, 1, 64),
    "presorted_by_caller": lambda: solve(sw_sorted, None, True),
    "presorted_by_caller_most_work_first": lambda: solve(sw_desc, None, True),
    "random_order_sort_key_neg_mu": lambda: solve(sw, neg_mu, True),
    "sort_key_mu_perm_in_kernel": lambda: solve(sw, sw[0], False),
    "auto_probe_perm_in_kernel": lambda: solve(sw, "auto", False),
    "sort_key_mu_physical_reorder": lambda: solve(sw, sw[0], False, 1),
    "sort_key_mu_physical_reorder_with_counters": lambda: solve(sw, sw[0], True, 1),
    "auto_probe_physical_reorder_with_counters": lambda: solve(sw, "auto", True, 1),
    "late_random_order": lambda: solve(sw3, None, True, 0, 30.0),
    "late_sort_key_neg_mu": lambda: solve(sw3, neg_mu3, True, 0, 30.0),
    "late_auto_probe": lambda: solve(sw3, "auto", True, 0, 30.0),
    "late_auto_probe_resuming": lambda: solve(sw3, "auto", True, 0, 30.0, 1),
    "late_auto_probe_rebin_after_32": lambda: solve(sw3, "auto", True, 0, 30.0, 1, 1, 32),
    "late_auto_probe_rebin_after_64": lambda: solve(sw3, "auto", True, 0, 30.0, 1, 1, 64),
    "late_auto_probe_rebin_after_128": lambda: solve(sw3, "auto", True, 0, 30.0, 1, 1, 128),
    "late_sort_key_true_step_count": lambda: solve(sw3, true_steps_key, True, 0, 30.0),
    "narrow_key_random_order": lambda: solve(sw2, None, False, 0, 0.5),
    "narrow_key_sort_key_mu": lambda: solve(sw2, sw2[0], False, 0, 0.5),
}
neg_mu3 = (-sw3[0]).contiguous()
_c = solve(sw3, None, True, 0, 30.0)[2]
true_steps_key = (-(_c["steps"] + _c["rejected"]).double()).contiguous()   # the bound for any key: the attempts each IVP really takes, most first
for _ in range(150):   # sustained clocks before anything is timed
    cases["random_order"]()
torch.cuda.synchronize()
times = {k: [] for k in cases}
outs = {}
for r in range(11):
    for k, fn in cases.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
        out = fn(); e1.record()
        torch.cuda.synchronize(); times[k].append(e0.elapsed_time(e1))
        if r == 0:
            outs[k] = out
L.nnhip_tune_set(b"sort_copy", 0)
L.nnhip_tune_set(b"sort_resume", 0)
L.nnhip_tune_set(b"sort_auto_key", 1)
L.nnhip_tune_set(b"sort_rebin_steps", 0)
res = {}
ref = outs["random_order"][1]
ref3 = outs["late_random_order"][1]
for k, tt in times.items():
    res[k] = dict(ms=sorted(tt[2:])[len(tt[2:]) // 2])
    if ("sort_key" in k or "auto_probe" in k) and "narrow" not in k:
        r = ref3 if k.startswith("late_") else ref
        res[k]["bit_identical_to_unsorted"] = bool(torch.equal(torch.nan_to_num(outs[k][1], nan=-1.0), torch.nan_to_num(r, nan=-1.0)))
for k in ("random_order", "presorted_by_caller", "late_random_order"):
    cnt = outs[k][2]
    st = (cnt["steps"] + cnt["rejected"]).double()
    w = st[: n // 64 * 64].reshape(-1, 64)
    res[k].update(attempted_steps_mean=float(st.mean()), attempted_steps_max=float(st.max()), lane_utilisation=float(w.mean() / w.max(dim=1).values.mean()))
print(json.dumps(res, indent=1))
