#!/usr/bin/env python3
"""HBM-resident adaptive streaming (advance kernels) vs the fused solve:
C3 shape (Lorenz, thread-per-IVP) at 1e6 and 1e7 IVPs and C4 shape (16-component ring, lanes-per-system) at 1e6 systems.
Algorithmic bytes per step and IVP (a rejected attempt is retried inside the launch, so the state moves once per ACCEPTED step):
  FSAL carried through HBM (knob adv_recompute_fsal = 0: the IntegratorProc signature as the reference passes it)   8*(4d+4)
  FSAL re-evaluated per launch (default for DOPRI54 / Tsit54 since round 3: the same bits, one more evaluation of f) 8*(2d+4)
(rounds 1-2 also stored the error estimate: 8*(4d+5); `frac_of_8TBps_round2_bytes` prices the time with that figure for continuity.)
Each config runs on a side stream: eager launches (default since round 3) and hipGraph-replayed polling groups (stream_graph=1), non-temporal hint
forced off / on, FSAL carried, K = 2 / 5 loop iterations per launch.
Wall clock of the whole loop incl. host polling; `us_per_iteration` divides by the loop iterations that do work
(= max accepted steps over the batch), speculative tail launches are overhead, not work."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import numericalnim_amd as nn

dev = torch.device("cuda:0")
L = nn._lib.lib()
res = {}
side = torch.cuda.Stream()


def run(f, y0, integ, layout, check_every, reps=3):
    best = None
    with torch.cuda.stream(side):
        for _ in range(reps + 1):  # first call: capture (or warm-up)
            y = y0.clone()
            side.synchronize()
            c0 = time.perf_counter()
            ys, launches = nn.adaptiveStream(f, y, 0.0, 1.0, nn.newODEoptions(), integrator=integ, layout=layout, check_every=check_every)
            side.synchronize()
            c1 = time.perf_counter()
            if best is None or c1 - c0 < best[0]:
                best = (c1 - c0, launches, ys)
    return best


if os.environ.get("ADV_BLOCK"):   # workgroup size of the thread-per-IVP advance kernel (tuning knob "adv_block")
    assert L.nnhip_tune_set(b"adv_block", int(os.environ["ADV_BLOCK"])) == 0
cases = []
ONLY = os.environ.get("ADV_BENCH_ONLY", "")   # e.g. "C3_lorenz_N1e+07" to profile one config
MODES = [m for m in os.environ.get("ADV_BENCH_MODES", "").split(",") if m]     # e.g. "default,fsal_carried"
INTEGS = [m for m in os.environ.get("ADV_BENCH_INTEGRATORS", "").split(",") if m] or ["dopri54", "tsit54"]
NOVR = int(os.environ.get("ADV_BENCH_N", "0"))   # dry runs of this script (e.g. on the ISA-backed fake node): every batch this size; the names keep BASELINE's sizes
for n in (1_000_000, 10_000_000):
    label, n = f"C3_lorenz_N{n:.0e}", NOVR or n
    y0 = torch.from_numpy(np.stack([1.0 + (np.arange(n) % 1024) * 2.0 ** -20, np.ones(n), np.ones(n)])).to(dev)
    cases.append((label, nn.Rhs.lorenz(), y0, 0, 3, n))
n = NOVR or 1_000_000
y16 = torch.from_numpy(1.0 + np.arange(16)[None, :] / 16 + ((np.arange(n) % 1024) * 2.0 ** -20)[:, None]).to(dev)
cases.append(("C4_ring16_N1e+06", nn.Rhs.ring(0.1), y16, 1, 16, n))
for name, f, y0, layout, d, n in cases:
    if ONLY and ONLY not in name:
        continue
    for integ in INTEGS:
        t, yf, cnt = nn.solveODE(f, y0, [0.0, 1.0], nn.newODEoptions(), integrator=integ, layout=layout, return_counts=True)
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        nn.solveODE(f, y0, [0.0, 1.0], nn.newODEoptions(), integrator=integ, layout=layout)
        torch.cuda.synchronize()
        fused_ms = (time.perf_counter() - c0) * 1e3
        iters = int(cnt["steps"].max())                       # loop iterations until the slowest IVP is done
        accepted = int(cnt["steps"].sum())
        # default = eager launches, non-temporal hint chosen from the working-set size, FSAL re-evaluated; nt0 / nt1 force the hint off / on;
        # graph = polling groups replayed from a hipGraph (knob stream_graph 1); fsal_carried = knob adv_recompute_fsal 0
        # K > 1 (knob "adv_steps_per_launch"): K loop iterations per IVP and launch with the state kept in registers in between — ITS OWN
        # traffic model (1/K of the bytes per step), reported beside the one-iteration-per-launch figures, never mixed with them
        # round 6: default = the library's defaults = the configuration with a hardware record (general kernels, polling groups of 8: round 4's);
        # lean = knob adv_lean 1 (same bits); lean_auto_poll = + the library's own polling schedule (knob adv_auto_poll 1, check_every 0): round 5's pair;
        # general_auto_poll = the schedule alone; lean_auto_poll_fp_contract = + the FMA-contracted lean kernels (within 1e-6, not bit-equal)
        #       mode                          graph nt  K refsal lean ce auto contract
        for mode, knob, nt, K, refsal, lean, ce, auto, contract in (
                ("default", 2, -1, 1, -1, 0, 0, 0, 0), ("lean", 2, -1, 1, -1, 1, 0, 0, 0), ("lean_auto_poll", 2, -1, 1, -1, 1, 0, 1, 0), ("general_auto_poll", 2, -1, 1, -1, 0, 0, 1, 0),
                ("lean_auto_poll_fp_contract", 2, -1, 1, -1, 1, 0, 1, 1), ("nt0", 2, 0, 1, -1, 0, 8, 0, 0), ("nt1", 2, 1, 1, -1, 0, 8, 0, 0),
                ("graph", 1, -1, 1, -1, 0, 8, 0, 0), ("fsal_carried", 2, -1, 1, 0, 0, 8, 0, 0), ("graph_fsal_carried", 1, -1, 1, 0, 0, 8, 0, 0),
                ("K2", 2, -1, 2, -1, 0, 8, 0, 0), ("K5", 2, -1, 5, -1, 0, 8, 0, 0), ("K5_fsal_carried", 2, -1, 5, 0, 0, 8, 0, 0)):
            if MODES and mode not in MODES:
                continue
            L.nnhip_tune_set(b"stream_graph", knob)
            L.nnhip_tune_set(b"adv_nontemporal", nt)
            L.nnhip_tune_set(b"adv_steps_per_launch", K)
            L.nnhip_tune_set(b"adv_recompute_fsal", refsal)
            L.nnhip_tune_set(b"adv_lean", lean)
            L.nnhip_tune_set(b"adv_auto_poll", auto)
            L.nnhip_tune_set(b"fp_contract", contract)
            dt, launches, ys = run(f, y0, integ, layout, ce)
            per_step = 8 * (4 * d + 4) if refsal == 0 else 8 * (2 * d + 4)
            nb = per_step * accepted / K
            res[f"{name}_{integ}_{mode}"] = dict(stream_ms=dt * 1e3, launches=launches, iterations=iters, us_per_iteration=dt * 1e6 / iters, steps_per_launch=K,
                                                 fsal="carried through HBM" if refsal == 0 else "re-evaluated per launch", bytes_per_step=per_step / K,
                                                 GBps=nb / dt / 1e9, frac_of_8TBps=nb / dt / 8e12,
                                                 frac_of_8TBps_round2_bytes=8 * (4 * d + 5) * accepted / K / dt / 8e12,
                                                 fused_ms=fused_ms, equal_to_fused=bool(torch.equal(ys, yf[-1])), max_abs_deviation_from_fused=float((ys - yf[-1]).abs().max()))
        L.nnhip_tune_set(b"adv_recompute_fsal", -1)
        L.nnhip_tune_set(b"adv_lean", 0)
        L.nnhip_tune_set(b"adv_auto_poll", 0)
        L.nnhip_tune_set(b"fp_contract", 0)
        L.nnhip_tune_set(b"stream_graph", 2)
        L.nnhip_tune_set(b"adv_nontemporal", -1)
        L.nnhip_tune_set(b"adv_steps_per_launch", 1)
print(json.dumps(res, indent=1))
