#!/usr/bin/env python3
"""Soak of the entries AROUND the fused solve: for the random cases of tests/test_gpu_fuzz_parity.py (any integrator, right-hand side,
size, layout, tspan incl. both directions / duplicates / tStart inside or outside, options) the results of
  * the whole ODESolver through the IntegratorProc seam (nnhip_ode_fixed_stream_dense_f64_dev / nnhip_ode_adaptive_stream_dense_f64_dev),
  * the divergence-binned solve (caller's key and automatic probe),
  * the per-call solves (every IVP its own 2-point or n_t-point tspan and option fields; device tables and the host forms with option objects)
must be the bits of the plain fused solve (which tests/tools/soak_fuzz.py compares with the oracle).  COUNTS mismatching cases instead
of asserting; writes JSON.   usage: python tests/tools/soak_paths.py [n_seeds] [out.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import numericalnim_amd as nn
import test_gpu_fuzz_parity as T

dev = torch.device("cuda:0")
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 600
out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "r03_paths_soak.json")
KEYS = T.KEYS
# optional: the streaming loops with FSAL carried through HBM (knob adv_recompute_fsal = 0) / polling groups replayed from hipGraphs (stream_graph = 1)
for _knob, _env in ((b"adv_recompute_fsal", "NNHIP_SOAK_RECOMPUTE_FSAL"), (b"stream_graph", "NNHIP_SOAK_STREAM_GRAPH")):
    if os.environ.get(_env, "") != "":
        assert nn._lib.lib().nnhip_tune_set(_knob, int(os.environ[_env])) == 0


def same(a, b):
    return a.shape == b.shape and bool(torch.equal(torch.nan_to_num(a, nan=-7.25), torch.nan_to_num(b, nan=-7.25)))


tab = {k: dict(cases=0, ivps=0, mismatching_cases=0, errors=0) for k in ("stream_solve", "sorted_key", "sorted_probe", "calls_dev", "calls_host", "host_solve",
                                                                         "sweep_host", "sweep_sorted", "sweep_calls", "stream_final", "stream_final_k", "tspans_dev", "tspans_host")}
bad = []
seed0 = int(os.environ.get("NNHIP_SOAK_SEED0", "5000"))
for seed in range(seed0, seed0 + n_seeds):
    rng = np.random.default_rng(seed)
    kind, dim, params, integ, ts, opt, n, layout = T._draw(rng, nn)
    y0 = rng.uniform(-1.5, 1.5, (n, dim)) + (np.array([0.0, 0.0, 15.0]) if kind == 2 else 0.0)
    y0l = np.ascontiguousarray(y0 if layout == 1 else y0.T)
    if dim == 1:
        y0l, layout = y0[:, 0].copy(), 0
    f = nn.Rhs(kind, KEYS.get(kind, ()), dict(zip(KEYS.get(kind, ()), params)))
    o = nn.newODEoptions(**opt)
    yt = torch.from_numpy(y0l).to(dev)
    t, y, cnt = nn.solveODE(f, yt, ts, o, integrator=integ, layout=layout, return_counts=True)
    fixed = integ in nn.fixedODE

    def note(name, ok):
        r = tab[name]
        r["cases"] += 1
        r["ivps"] += n
        if not ok:
            r["mismatching_cases"] += 1
            bad.append(dict(path=name, seed=seed, integ=integ, kind=kind, dim=dim, layout=layout, n=n, ts=list(map(float, ts)), opt=opt))

    try:
        if fixed:
            t2, y2, ny2, _ = nn.fixedStreamSolve(f, yt, ts, o, integrator=integ, layout=layout)
            ok = np.array_equal(t, t2) and same(y, y2) and bool((cnt["ny"] == int(ny2)).all())
        else:
            t2, y2, ny2, _ = nn.adaptiveStreamSolve(f, yt, ts, o, integrator=integ, layout=layout, check_every=int(rng.choice([1, 3, 8])))
            ok = np.array_equal(t, t2) and same(y, y2) and bool(torch.equal(cnt["ny"], ny2))
        note("stream_solve", ok)
    except Exception as e:  # an entry that refuses a case the fused solve accepts is a finding as well
        tab["stream_solve"]["errors"] += 1
        bad.append(dict(path="stream_solve", seed=seed, error=str(e)[:200], integ=integ, kind=kind, dim=dim))
    try:
        key = torch.from_numpy(rng.uniform(0, 1, n)).to(dev)
        t3, y3, c3 = nn.solveODE(f, yt, ts, o, integrator=integ, layout=layout, return_counts=True, sort_by=key)
        note("sorted_key", np.array_equal(t, t3) and same(y, y3) and all(bool(torch.equal(cnt[k], c3[k])) for k in ("ny", "steps", "rejected")))
        t4, y4, c4 = nn.solveODE(f, yt, ts, o, integrator=integ, layout=layout, return_counts=True, sort_by="auto")
        note("sorted_probe", np.array_equal(t, t4) and same(y, y4) and all(bool(torch.equal(cnt[k], c4[k])) for k in ("ny", "steps", "rejected")))
    except Exception as e:
        tab["sorted_key"]["errors"] += 1
        bad.append(dict(path="sorted", seed=seed, error=str(e)[:200], integ=integ, kind=kind, dim=dim))
    # per-call: IVP i integrates [tStart, te_i]; compared with 2-point fused solves of the distinct ends (4 distinct values)
    try:
        ends = np.round(opt["tStart"] + rng.uniform(-0.4, 0.4, 4), 3)
        ends[0] = opt["tStart"]
        te = ends[rng.integers(0, 4, n)]
        yc, cc = nn.solveODEPerIvpEnd(f, yt, torch.from_numpy(te).to(dev), o, integrator=integ, layout=layout)
        ok = True
        for e in ends:
            tt, ye, ce = nn.solveODE(f, yt, [opt["tStart"], float(e)], o, integrator=integ, layout=layout, return_counts=True)
            m = torch.from_numpy(te == e).to(dev)
            if dim == 1:
                a, b = yc[:, m], ye[:, m]
            elif layout == 0:
                a, b = yc[:, :, m], ye[:, :, m]
            else:
                a, b = yc[:, m, :], ye[:, m, :]
            ok = ok and same(a, b) and bool(torch.equal(cc["ny"][m], ce["ny"][m])) and bool(torch.equal(cc["steps"][m], ce["steps"][m]))
        note("calls_dev", ok)
        yh, ch = nn.solveODECalls(f, y0l, te, o, integrator=integ, layout=layout)
        note("calls_host", same(torch.from_numpy(yh), yc.cpu()) and bool(np.array_equal(ch["ny"], cc["ny"].cpu().numpy())))
    except Exception as e:
        tab["calls_dev"]["errors"] += 1
        bad.append(dict(path="calls", seed=seed, error=str(e)[:200], integ=integ, kind=kind, dim=dim))
    # host-pointer entry (chunked pinned pipeline) vs device entry; per-IVP parameter sweeps through host / binned / per-call entries
    try:
        th, yh2, ch2 = nn.solveODE(f, y0l, ts, o, integrator=integ, layout=layout, return_counts=True)
        note("host_solve", np.array_equal(t, th) and same(torch.from_numpy(yh2), y.cpu()) and np.array_equal(ch2["steps"], cnt["steps"].cpu().numpy()))
        if len(params) >= 1:
            sw = np.ascontiguousarray(params[0] * rng.uniform(0.5, 1.5, (1, n)))
            swt = torch.from_numpy(sw).to(dev)
            ts_, ys_, cs_ = nn.solveODE(f, yt, ts, o, integrator=integ, layout=layout, return_counts=True, sweep=swt)
            tsh, ysh, csh = nn.solveODE(f, y0l, ts, o, integrator=integ, layout=layout, return_counts=True, sweep=sw)
            note("sweep_host", same(torch.from_numpy(ysh), ys_.cpu()) and np.array_equal(csh["steps"], cs_["steps"].cpu().numpy()))
            tso, yso, cso = nn.solveODE(f, yt, ts, o, integrator=integ, layout=layout, return_counts=True, sweep=swt, sort_by="auto")
            note("sweep_sorted", same(yso, ys_) and bool(torch.equal(cso["steps"], cs_["steps"])))
            e = float(np.round(opt["tStart"] + rng.uniform(0.05, 0.4), 3))
            t2p, y2p, c2p = nn.solveODE(f, yt, [opt["tStart"], e], o, integrator=integ, layout=layout, return_counts=True, sweep=swt)
            ycs, ccs = nn.solveODEPerIvpEnd(f, yt, torch.full((n,), e, dtype=torch.float64, device=dev), o, integrator=integ, layout=layout, sweep=swt)
            note("sweep_calls", same(ycs, y2p) and bool(torch.equal(ccs["steps"], c2p["steps"])))
    except Exception as e:
        tab["host_solve"]["errors"] += 1
        bad.append(dict(path="host/sweep", seed=seed, error=str(e)[:200], integ=integ, kind=kind, dim=dim))
    # the streaming loops without dense output: final state of a forward 2-point span
    try:
        e = float(np.round(opt["tStart"] + rng.uniform(0.05, 0.4), 3))
        t2p, y2p = nn.solveODE(f, yt, [opt["tStart"], e], o, integrator=integ, layout=layout)
        if fixed:
            yfin, _ = nn.fixedStream(f, yt.clone(), opt["tStart"], e, o, integrator=integ, layout=layout)
        else:
            yfin, _ = nn.adaptiveStream(f, yt.clone(), opt["tStart"], e, o, integrator=integ, layout=layout, check_every=int(rng.choice([1, 4, 8])))
        torch.cuda.synchronize()
        note("stream_final", same(yfin, y2p[-1]))
        if not fixed:  # round 3: K loop iterations per launch (own kernel instantiation) must give the same bits
            K = int(rng.choice([2, 3, 5, 16]))
            try:
                yk, _ = nn.adaptiveStream(f, yt.clone(), opt["tStart"], e, o, integrator=integ, layout=layout, check_every=int(rng.choice([1, 4])), steps_per_launch=K)
                torch.cuda.synchronize()
                note("stream_final_k", same(yk, y2p[-1]))
            finally:
                nn._lib.lib().nnhip_tune_set(b"adv_steps_per_launch", 1)
    except Exception as e:
        tab["stream_final"]["errors"] += 1
        bad.append(dict(path="stream_final", seed=seed, error=str(e)[:200], integ=integ, kind=kind, dim=dim))
    # per-IVP n_t-point tspans: every row a permutation of the batch tspan -> the plain solve's bits, times in sorted order
    try:
        if len(ts) > 0:
            rows = np.stack([rng.permutation(ts) for _ in range(n)])
            tq, yq, cq = nn.solveODEPerIvpTspan(f, yt, torch.from_numpy(rows).to(dev), o, integrator=integ, layout=layout)
            tq = tq.cpu().numpy()
            ok = all(np.array_equal(tq[i, :len(t)], t) and np.isnan(tq[i, len(t):]).all() for i in range(0, n, max(1, n // 7)))
            note("tspans_dev", ok and same(yq, y) and all(bool(torch.equal(cnt[k], cq[k])) for k in ("ny", "steps", "rejected")))
            th, yh3, ch3 = nn.solveODECallsTspan(f, y0l, rows, o, integrator=integ, layout=layout)
            note("tspans_host", same(torch.from_numpy(yh3), yq.cpu()) and np.array_equal(ch3["steps"], cq["steps"].cpu().numpy()))
    except Exception as e:
        tab["tspans_dev"]["errors"] += 1
        bad.append(dict(path="tspans", seed=seed, error=str(e)[:200], integ=integ, kind=kind, dim=dim))
res = {"seeds": [seed0, seed0 + n_seeds], "per_path": tab, "findings": bad[:40]}
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(tab))
print(json.dumps(bad[:8]))
