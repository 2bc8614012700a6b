#!/usr/bin/env python3
"""bench.py — BASELINE.json headline: RK4 trajectory-steps/s on 1e7 float64 IVPs per GPU, with the achieved
HBM bandwidth of the step-streaming kernel against the 8 TB/s roofline.

One bench "step" = one pass of the hot path over the batch: the fixed-step ODESolver loop
(ode.nim:511-532) for config C2 — dy/dt = -y, N = 1e7 scalar float64 IVPs per GPU, dt = 2^-10,
tspan = [0, 1000*dt] -> exactly 1000 RK4_step (ode.nim:180-189) launches of the step-streaming kernel,
state resident in HBM between launches (16 algorithmic bytes per trajectory-step).
Multi-GPU (weak scaling, config C5): every rank owns a contiguous shard of the global IVP index
range; the only collective is one all-gather of the final states per solve (RCCL over xGMI).

Usage: python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n-ivp", type=float, default=1e7, help="IVPs per GPU (config C2: 1e7)")
    ap.add_argument("--rk4-steps", type=int, default=1000, help="RK4 time steps per solve (C2: 1000)")
    ap.add_argument("--pingpong", type=int, default=1, help="1: ping-pong between two state buffers (default), 0: update in place")
    ap.add_argument("--no-gather", action="store_true", help="skip the final-state all-gather (N>1)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="issue the final-state all-gather on the compute stream instead of overlapping it with the next solve")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL, the default and the only one measured; gloo lets several ranks share "
                         "one GPU to exercise the multi-rank code path on a single-GPU box)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (nccl=RCCL) and run the all-gather even at world size 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=float, default=2e5, help="IVPs in the CPU-baseline sample (2e5 x 1000 steps = ~3 s on one core; the rate extrapolates linearly: IVPs are independent)")
    ap.add_argument("--no-fused", action="store_true", help="skip the informational fused-solve measurement")
    ap.add_argument("--verify-gathers", action="store_true",
                    help="debug: after every overlapped all-gather completes, compare this rank's slice of the gathered tensor with the "
                         "solve result it was issued for (checks the buffer rotation; adds a device comparison per solve)")
    ap.add_argument("--no-check", action="store_true", help="skip the parity comparison inside the cpu_baseline leg and the all-gather placement check")
    return ap.parse_args()


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    import numericalnim_amd as nn
    from numericalnim_amd import distributed as nd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    dev_index = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    n = int(args.n_ivp)
    nsteps = int(args.rk4_steps)
    dt = 2.0 ** -10
    t_end = nsteps * dt  # exact in binary: t accumulates without rounding -> exactly nsteps launches
    opt = nn.newODEoptions(dt=dt)
    f = nn.Rhs.neg_y()
    lo, hi = nd.shard_range(n * world, rank, world)
    y0 = nd.c2_y0_torch(lo, hi, dev)
    # Two buffer sets: with the all-gather of solve k overlapped with solve k+1 (second HIP stream), solve k+1 must not
    # overwrite the states that are still being gathered.
    overlap = use_dist and not args.no_gather and not args.no_overlap
    # Overlapped gather + ping-pong: three state buffers rotate (solve k uses B[k%3] and B[(k+1)%3]; its result lands back in
    # B[k%3] after an even number of steps, so solve k+1 can start in B[(k+1)%3] / B[(k+2)%3] while B[k%3] is being gathered).
    rotate3 = overlap and bool(args.pingpong) and nsteps % 2 == 0
    nset = 3 if rotate3 else (2 if overlap else 1)
    ys = [torch.empty_like(y0) for _ in range(nset)]
    if rotate3:
        scratches = [ys[(s + 1) % 3] for s in range(3)]
    else:
        scratches = [torch.empty_like(y0) if args.pingpong else None for _ in range(nset)]
    gathered = torch.empty(n * world, dtype=torch.float64, device=dev) if (use_dist and not args.no_gather) else None
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    comm_stream = torch.cuda.Stream() if overlap else None
    pending = [None] * nset  # outstanding all-gather reading buffer set s
    state = {"i": 0, "expected": None, "verified": 0}

    def finish(q):  # wait for the all-gather that reads buffer set q
        pending[q].wait()
        pending[q] = None
        if args.verify_gathers and state["expected"] is not None:
            assert torch.equal(gathered[lo:hi], state["expected"]), "an overlapped all-gather read a buffer that was being overwritten"
            state["verified"] += 1

    def one_solve(k=None):
        s = state["i"] % nset
        state["i"] += 1
        for q in ((s, (s + 1) % 3) if rotate3 else (s,)):  # buffers about to be overwritten must have been gathered
            if pending[q] is not None:
                finish(q)
        y = ys[s]
        y.copy_(y0)  # solveODE starts from y0 (y0.clone(), ode.nim:482)
        if k is not None:
            ev[k][0].record()
        yf, ns = nn.fixedStream(f, y, 0.0, t_end, opt, integrator="rk4", scratch=scratches[s])
        if k is not None:
            ev[k][1].record()
        assert ns == nsteps, (ns, nsteps)
        if args.verify_gathers and state["expected"] is None:
            state["expected"] = yf.clone()  # every solve starts from the same y0, so every result equals the first one
        if gathered is not None:
            if overlap:
                done = torch.cuda.Event()
                done.record()
                with torch.cuda.stream(comm_stream):
                    comm_stream.wait_event(done)
                    pending[s] = dist.all_gather_into_tensor(gathered, yf, async_op=True)  # RCCL over xGMI, off the compute stream
            else:
                dist.all_gather_into_tensor(gathered, yf)
        return yf

    def drain():
        for s in range(nset):
            if pending[s] is not None:
                finish(s)

    def sync_all():
        drain()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_solve()
    sync_all()
    t0 = time.perf_counter()
    for k in range(args.steps):
        yf = one_solve(k)
    sync_all()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    total_traj_steps = float(n) * world * nsteps * args.steps
    value = total_traj_steps / elapsed
    kern_ms = sum(a.elapsed_time(b) for a, b in ev)  # device time of the 1000*K step launches on this rank
    launch_s = kern_ms * 1e-3 / (args.steps * nsteps)
    algo_bytes = 16.0 * n  # 8 B read + 8 B written per trajectory-step, n trajectory-steps per launch
    achieved = algo_bytes / launch_s / 1e9
    gather_ms = None
    if gathered is not None:  # the collective alone, timed after the run (it is overlapped inside the timed region)
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.all_gather_into_tensor(gathered, yf)
        torch.cuda.synchronize()
        g0.record()
        for _ in range(3):
            dist.all_gather_into_tensor(gathered, yf)
        g1.record()
        torch.cuda.synchronize()
        gather_ms = g0.elapsed_time(g1) / 3.0

    check = None  # filled by the cpu_baseline leg below (the only place bench.py touches the oracle)

    if gathered is not None and not args.no_check:
        # the gathered tensor must hold every rank's final states in rank order
        assert torch.equal(gathered[lo:hi], yf), "all-gather misplaced this rank's shard"
    if rank != 0:
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- PMC traffic from the committed profile of this same command, if present ------------------------
    # (PMC counters cannot be read from inside the process being timed: `traffic` is the STATIC figure of the committed profile,
    # collected by scripts/profile_gpu.sh on an earlier run of this same command; roofline.traffic_source says so.)
    traffic = None
    traffic_src = None
    pj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pj) and n == 10_000_000:
        try:
            pm = json.load(open(pj))
            traffic = pm.get("rk4_stream", {}).get("hbm_bytes_per_launch")
            traffic_src = "profiles/pmc_traffic.json (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of round %s, scripts/profile_gpu.sh; not measured in this run)" % pm.get("round")
        except Exception:
            traffic = None

    out = {
        "metric": "RK4 trajectory-steps/sec on 1e7 float64 IVPs",
        "value": value,
        "unit": "trajectory-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "C2: RK4 fixed-step step-streaming, dy/dt=-y, %d scalar float64 IVPs per GPU x %d steps (dt=2^-10), "
                        "one RK4_step kernel launch per time step, state in HBM between launches" % (n, nsteps),
            "ivps_per_gpu": n, "rk4_steps": nsteps, "state_update": "pingpong" if args.pingpong else "in-place",
            "final_state_allgather": bool(gathered is not None), "allgather_overlapped_with_next_solve": bool(overlap),
            **({"backend": args.backend} if args.backend != "nccl" else {}),
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic,
            "traffic_source": traffic_src,
            "kernel": "rk4_stream_vec_kernel", "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_us": launch_s * 1e6,
            # the headline batch (2 x 80 MB of ping-pong state) lives in the 256 MiB Infinity Cache, and gfx950's FETCH_SIZE counts
            # Infinity-Cache hits: `frac` is the kernel's streaming rate, `frac_hbm_only` (filled from the 6.4e7-IVP leg below,
            # 1 GB working set) is the part that is certainly HBM
            "frac_hbm_only": None, "achieved_hbm_only": None,
        },
        "parity_max_abs_err_vs_oracle": check,  # set by the cpu_baseline leg
    }
    if gather_ms is not None:
        out["allgather_ms_per_solve"] = gather_ms
    if args.verify_gathers:
        out["gathers_verified"] = state["verified"]

    # ---- informational: the fused whole-solve kernel (FP64-VALU bound; the HBM roofline does not apply) ----
    if not args.no_fused:
        for _ in range(2):
            nn.solveODE(f, y0, [0.0, t_end], opt, integrator="rk4")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 3
        for _ in range(reps):
            _, yfu = nn.solveODE(f, y0, [0.0, t_end], opt, integrator="rk4")
        e1.record()
        torch.cuda.synchronize()
        fs = e0.elapsed_time(e1) * 1e-3 / reps
        out["fused_solve"] = {"value": float(n) * nsteps / fs, "unit": "trajectory-steps/s", "ms_per_solve": fs * 1e3,
                              "bound": "fp64-valu", "model_fp64_flop_per_s": 16.0 * n * nsteps / fs,  # SURVEY.md §8d model: 16 flop per step (≈12 VALU instructions after sign folding)
                              "bitwise_equal_to_stream": bool(torch.equal(yfu[-1], yf))}

    # ---- informational: the same kernel on a batch that cannot live in the 256 MiB Infinity Cache (1 GB of ping-pong state) ----
    if n > 20_000_000:  # the batch itself is beyond the Infinity Cache
        out["roofline"]["achieved_hbm_only"], out["roofline"]["frac_hbm_only"] = achieved, achieved / 8000.0
    if not args.no_fused and world == 1 and n <= 20_000_000:
        nb = 64_000_000
        yb = nd.c2_y0_torch(0, nb, dev)
        sb = torch.empty_like(yb)
        tb_end = 100 * dt
        nn.fixedStream(f, yb, 0.0, tb_end, opt, integrator="rk4", scratch=sb)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _, nsb = nn.fixedStream(f, yb, 0.0, tb_end, opt, integrator="rk4", scratch=sb)
        e1.record()
        torch.cuda.synchronize()
        lb = e0.elapsed_time(e1) * 1e-3 / nsb
        out["roofline"]["achieved_hbm_only"] = 16.0 * nb / lb / 1e9
        out["roofline"]["frac_hbm_only"] = 16.0 * nb / lb / 1e9 / 8000.0
        out["beyond_infinity_cache"] = {"ivps": nb, "launches": int(nsb), "avg_launch_us": lb * 1e6, "achieved": 16.0 * nb / lb / 1e9, "unit": "GB/s",
                                        "frac": 16.0 * nb / lb / 1e9 / 8000.0,
                                        "note": "same kernel family, 1 GB working set: the unambiguous HBM figure (the headline batch's 160 MB fit the Infinity Cache)"}
        del yb, sb

    # ---- informational: BASELINE.json's adaptive configs C3 / C4 (1e6 IVPs / systems), fused and through the HBM-resident loop ----
    if not args.no_fused and world == 1:
        cfg = {}
        n6 = 1_000_000
        y3 = torch.from_numpy(np.stack([1.0 + (np.arange(n6) % 1024) * 2.0 ** -20, np.ones(n6), np.ones(n6)])).to(dev)
        y16 = torch.from_numpy(1.0 + np.arange(16)[None, :] / 16 + ((np.arange(n6) % 1024) * 2.0 ** -20)[:, None]).to(dev)
        side = torch.cuda.Stream()
        for name, fr, yy, layout, integ, d in (("C3_dopri54_lorenz_1e6", nn.Rhs.lorenz(), y3, 0, "dopri54", 3), ("C4_tsit54_ring16_1e6", nn.Rhs.ring(0.1), y16, 1, "tsit54", 16)):
            _, yfu, cnt = nn.solveODE(fr, yy, [0.0, 1.0], nn.newODEoptions(), integrator=integ, layout=layout, return_counts=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                nn.solveODE(fr, yy, [0.0, 1.0], nn.newODEoptions(), integrator=integ, layout=layout)
            e1.record()
            torch.cuda.synchronize()
            iters = int(cnt["steps"].max())
            best, ys = None, None
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    yw = yy.clone()
                    side.synchronize()
                    c0 = time.perf_counter()
                    ys, _l = nn.adaptiveStream(fr, yw, 0.0, 1.0, nn.newODEoptions(), integrator=integ, layout=layout)
                    side.synchronize()
                    dtw = time.perf_counter() - c0
                    best = dtw if best is None or dtw < best else best
            per_step = 8 * (2 * d + 4)  # y in / out and (t, dt) in / out; FSAL is re-evaluated per launch
            cfg[name] = {"fused_ms": e0.elapsed_time(e1) / 3, "streamed_ms": best * 1e3, "loop_iterations": iters, "streamed_us_per_iteration": best * 1e6 / iters,
                         "streamed_bytes_per_step": per_step, "streamed_GBps": per_step * float(cnt["steps"].sum()) / best / 1e9,
                         "streamed_bitwise_equal_to_fused": bool(torch.equal(ys, yfu[-1]))}
        out["adaptive_configs"] = cfg
        del y3, y16

    # ---- informational: batches whose members take different step sequences (every reference call is its own, ode.nim:589-591) ----
    # 1e6 Van der Pol IVPs with their own stiffness in random order: as handed over / binned below the boundary (automatic probe; the caller's key), and
    # 1e6 separate calls with their own tEnd: in the caller's order / longest span first.  All must equal the plain solves bit for bit.
    if not args.no_fused and world == 1:
        n6 = 1_000_000
        rng = np.random.default_rng(0)
        mu = torch.from_numpy(rng.uniform(0.1, 20.0, n6)[None, :].copy()).to(dev)
        yv = torch.from_numpy(np.stack([np.full(n6, 2.0), np.zeros(n6)])).to(dev)
        ov = nn.newODEoptions(absTol=1e-6, relTol=1e-6, dtMin=1e-9, dtMax=1.0)
        L = nn._lib.lib()

        def med_ms(fn, reps=5):
            fn(); torch.cuda.synchronize()
            tt = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); r = fn(); e1.record(); torch.cuda.synchronize(); tt.append(e0.elapsed_time(e1))
            return sorted(tt)[len(tt) // 2], r

        het = {}
        het["sweep_as_handed_over_ms"], ref = med_ms(lambda: nn.solveODE(nn.Rhs.vanderpol(), yv, [0.0, 10.0], ov, integrator="dopri54", sweep=mu))
        het["sweep_binned_automatic_probe_ms"], ra = med_ms(lambda: nn.solveODE(nn.Rhs.vanderpol(), yv, [0.0, 10.0], ov, integrator="dopri54", sweep=mu, sort_by="auto"))
        key = (-mu[0]).contiguous()   # the stiffest first
        het["sweep_binned_by_callers_key_ms"], rk = med_ms(lambda: nn.solveODE(nn.Rhs.vanderpol(), yv, [0.0, 10.0], ov, integrator="dopri54", sweep=mu, sort_by=key))
        het["sweep_bitwise_equal"] = bool(torch.equal(ref[1], ra[1]) and torch.equal(ref[1], rk[1]))
        te = torch.from_numpy(rng.uniform(0.05, 10.0, n6)).to(dev)
        try:
            L.nnhip_tune_set(b"calls_bin", 0)
            het["calls_in_callers_order_ms"], c0 = med_ms(lambda: nn.solveODEPerIvpEnd(nn.Rhs.vanderpol(2.0), yv, te, ov, integrator="dopri54"))
        finally:
            L.nnhip_tune_set(b"calls_bin", 1)
        het["calls_longest_span_first_ms"], c1 = med_ms(lambda: nn.solveODEPerIvpEnd(nn.Rhs.vanderpol(2.0), yv, te, ov, integrator="dopri54"))
        het["calls_bitwise_equal"] = bool(torch.equal(c0[0], c1[0]) and all(torch.equal(c0[1][k], c1[1][k]) for k in c0[1]))
        out["heterogeneous_batches"] = het
        del mu, yv, te

    # ---- CPU baseline: the oracle (C++ restatement of the reference) on this box's host cores -------------
    if not args.no_cpu_baseline and world == 1:
        from oracle import oracle as O
        ns = int(args.cpu_sample)
        y0s = nd.c2_y0_numpy(0, ns)
        oo = O.new_options(dt=dt)
        O.solve_ode_batch(O.RHS_NEG_Y, [], y0s[:1000], 1000, 0, [0.0, t_end], oo, "rk4")  # warm
        c0 = time.perf_counter()
        cpu = O.solve_ode_batch(O.RHS_NEG_Y, [], y0s, ns, 0, [0.0, t_end], oo, "rk4", n_threads=1)
        c1 = time.perf_counter()
        if not args.no_check:  # the CPU port just integrated the first `ns` IVPs of the timed batch: compare the GPU's result
            k = min(ns, n)
            check = float(np.abs(yf[:k].cpu().numpy() - cpu["y"][-1, 0][:k]).max())
            assert check <= 1e-10, f"parity failure vs oracle: max abs err {check}"
            out["parity_max_abs_err_vs_oracle"] = check
            out["parity_checked_ivps"] = k
        ncores = os.cpu_count() or 1
        O.solve_ode_batch(O.RHS_NEG_Y, [], y0s, ns, 0, [0.0, t_end], oo, "rk4", n_threads=ncores)
        c2 = time.perf_counter()
        out["cpu_baseline"] = {
            "value": ns * nsteps / (c1 - c0), "unit": "trajectory-steps/s", "cores": 1, "kind": "port",
            "sample": "first %d IVPs of the C2 batch x %d RK4 steps through the oracle's solveODE (closure-style RHS call), "
                      "1 thread = the single-threaded reference; IVPs are independent so the rate extrapolates linearly" % (ns, nsteps),
            "all_cores": {"value": ns * nsteps / (c2 - c1), "cores": ncores},
        }
    # RCCL prints a banner ("Librccl path : ...") through C stdio, which would otherwise be flushed AFTER this line at
    # exit; flush C stdio first so that the JSON line is the last thing on stdout.
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
