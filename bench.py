#!/usr/bin/env python3
"""bench.py — BASELINE.json headline: RK4 trajectory-steps/s on 1e7 float64 IVPs per GPU, with the achieved
HBM bandwidth of the step-streaming kernel against the 8 TB/s roofline.

One bench "step" = one pass of the hot path over the batch: the fixed-step ODESolver loop
(ode.nim:511-532) for config C2 — dy/dt = -y, N = 1e7 scalar float64 IVPs per GPU, dt = 2^-10,
tspan = [0, 1000*dt] -> exactly 1000 RK4_step (ode.nim:180-189) launches of the step-streaming kernel,
state resident in HBM between launches (16 algorithmic bytes per trajectory-step).
Multi-GPU (weak scaling, config C5): every rank owns a contiguous shard of the global IVP index
range; the only collective is one all-gather of the final states per solve (RCCL over xGMI).

Usage: python bench.py --gpus N --steps K --warmup W
  N>1 either way: under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (the launcher's RANK /
  LOCAL_RANK / WORLD_SIZE are used), or plain `python bench.py --gpus N ...` — with WORLD_SIZE unset bench.py starts its own N ranks
  through torch.distributed.run on 127.0.0.1 (a free port) and rank 0's line is the output.  --gpus != WORLD_SIZE is an error.
Prints ONE JSON line on rank 0.  (Informational legs that go through settings without a hardware record — the opt-in lean / auto-poll / FMA-contracted
streamed kernels — are measured in a child process, `bench.py --child-leg streamed_opt_in`, under a timeout: see streamed_opt_in_parent.)
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
    ap.add_argument("--cpu-ivps-per-thread", type=float, default=2e4, help="all-cores CPU leg: IVPs per thread (C2)")
    ap.add_argument("--cpu-adaptive-sample", type=float, default=1e4, help="IVPs in the 1-core CPU sample of C3 / C4 (all cores: an eighth of it per thread)")
    ap.add_argument("--no-fused", action="store_true", help="skip the informational fused-solve measurement")
    ap.add_argument("--verify-gathers", action="store_true",
                    help="debug: after every overlapped all-gather completes, compare this rank's slice of the gathered tensor with the "
                         "solve result it was issued for (checks the buffer rotation; adds a device comparison per solve)")
    ap.add_argument("--adaptive-n", type=float, default=1e6, help="IVPs / systems of the informational C3 / C4 / heterogeneous-batch legs (BASELINE.json: 1e6; smaller only to exercise the legs, 0 skips them)")
    ap.add_argument("--beyond-cache-n", type=float, default=6.4e7, help="IVPs of the informational leg whose working set cannot live in the Infinity Cache (1 GB at 6.4e7)")
    ap.add_argument("--child-leg", default="", help="internal: run ONE informational leg in this (child) process and print its JSON object (bench.py starts it for the "
                                                     "settings that have no hardware record, so that a fault there cannot cost the parent its line)")
    ap.add_argument("--no-check", action="store_true", help="skip the parity comparison inside the cpu_baseline leg and the all-gather placement check")
    return ap.parse_args()


# ---- CPU baseline legs: the oracle (C++ restatement of the reference, -O3 as BASELINE.md section 2 states) on the host's cores ------------
# One core = the reference as it is (single-threaded, one solveODE call per IVP).  All cores = the same calls spread over an OpenMP team: the
# team is started before the clock, and the sample is sized so that every thread has >= 2e4 IVPs x 1000 steps (C2) of work — round 4's 12 ms
# slice per thread measured the fork, not the cores.  Plain functions of numpy arrays: tests/test_bench_cpu_legs.py runs them without a GPU.
def _all_cores(run, n_small, ncores):
    run(n_small, ncores)           # start the OpenMP team (and touch every thread's stack) before the clock
    a = time.perf_counter()
    r = run(None, ncores)
    return time.perf_counter() - a, r


def cpu_baseline_c2(O, y0_of, n, nsteps, dt, t_end, ns, ivps_per_thread, ncores, gpu_first=None):
    """-> (cpu_baseline object, max abs deviation of `gpu_first` (the GPU's final states of the first min(ns, n) IVPs) from the port or None, IVPs checked)"""
    import numpy as np
    ns = int(ns)
    y0s = y0_of(0, ns)
    oo = O.new_options(dt=dt)
    O.solve_ode_batch(O.RHS_NEG_Y, [], y0s[:1000], min(1000, ns), 0, [0.0, t_end], oo, "rk4")  # warm (page in the library)
    c0 = time.perf_counter()
    cpu = O.solve_ode_batch(O.RHS_NEG_Y, [], y0s, ns, 0, [0.0, t_end], oo, "rk4", n_threads=1)
    c1 = time.perf_counter()
    check, k = None, min(ns, n)
    if gpu_first is not None:  # the CPU port just integrated the first `ns` IVPs of the timed batch: compare the GPU's result
        check = float(np.abs(np.asarray(gpu_first)[:k] - cpu["y"][-1, 0][:k]).max())
        assert check <= 1e-10, f"parity failure vs oracle: max abs err {check}"
    one_core = ns * nsteps / (c1 - c0)
    na = int(min(n, max(ns, ivps_per_thread * ncores))) if ncores > 1 else ns
    y0a = y0_of(0, na)
    ta, _ = _all_cores(lambda m, th: O.solve_ode_batch(O.RHS_NEG_Y, [], y0a[:m] if m else y0a, m or na, 0, [0.0, t_end], oo, "rk4", n_threads=th), min(na, ncores * 8), ncores)
    return {
        "value": one_core, "unit": "trajectory-steps/s", "cores": 1, "kind": "port",
        "sample": "first %d IVPs of the C2 batch x %d RK4 steps through the oracle's solveODE (closure-style RHS call), "
                  "1 thread = the single-threaded reference; IVPs are independent so the rate extrapolates linearly" % (ns, nsteps),
        "all_cores": {"value": na * nsteps / ta, "cores": ncores, "speedup_over_1_core": na * nsteps / ta / one_core,
                      "sample": "first %d IVPs x %d steps (%d per thread), OpenMP team started before the clock" % (na, nsteps, na // ncores)},
    }, check, k


def cpu_baseline_adaptive(O, name, yh, layout, integ, d, gpu_first, n1, ncores):
    """C3 / C4 beside their GPU figures: the oracle's Vector path allocates a fresh seq per operator like the reference's (utils.nim:59-64,113-118,176-180) —
    that is where the reference's CPU time goes for vector states.  yh: the batch's initial states (host, in `layout`); gpu_first: the
    GPU's final states of its first n1 IVPs."""
    import numpy as np
    kind, par = (O.RHS_LORENZ, [10.0, 28.0, 8.0 / 3.0]) if d == 3 else (O.RHS_RING, [0.1])
    ntot = yh.shape[1] if layout == 0 else yh.shape[0]
    n1 = int(min(n1, ntot))
    nall = int(min(ntot, max(n1, (n1 // 8) * ncores)))
    sub = (lambda m: np.ascontiguousarray(yh[:, :m])) if layout == 0 else (lambda m: np.ascontiguousarray(yh[:m]))
    run = lambda m, th, tot: O.solve_ode_batch(kind, par, sub(m or tot), m or tot, d, [0.0, 1.0], O.new_options(), integ, layout=layout, n_threads=th)  # noqa: E731
    run(min(64, n1), 1, n1)
    a = time.perf_counter()
    r1 = run(None, 1, n1)
    t1 = time.perf_counter() - a
    tall, rall = _all_cores(lambda m, th: run(m, th, nall), min(nall, ncores * 2), ncores)
    devi = float(np.abs(np.asarray(gpu_first) - r1["y"][-1]).max())
    assert devi <= 1e-6, (name, devi)   # north_star's tolerance for adaptive methods (bit-equal when the host libm is the glibc the device restates)
    st1, sta = float(r1["steps"].sum()), float(rall["steps"].sum())
    return {
        "kind": "port", "unit": "IVPs/s", "value": n1 / t1, "cores": 1, "accepted_steps_per_s": st1 / t1,
        "sample": "first %d IVPs of the batch, one solveODE call each (Vector[float] path: a fresh seq per operator)" % n1,
        "all_cores": {"value": nall / tall, "cores": ncores, "accepted_steps_per_s": sta / tall, "speedup_over_1_core": (nall / tall) / (n1 / t1),
                      "sample": "first %d IVPs, OpenMP team started before the clock" % nall},
        "max_abs_dev_gpu_vs_cpu": devi,
    }


# ---- the opt-in settings of the streamed adaptive loop, in a process of their own ---------------------------------------------------------
# adv_lean / adv_auto_poll / fp_contract select kernels and a polling schedule that have never been timed — or run — on hardware.  A Python
# exception in such a leg is caught where it is raised; a device fault or a hang is not catchable from inside the process that took it.  So bench.py runs these legs in a CHILD
# (`bench.py --child-leg streamed_opt_in`) with a timeout: the parent's line — the headline, the roofline, the recorded configuration of C3 / C4 — does not depend on
# code without a record.  The child builds the same inputs, takes the fused solve (recorded code) as its reference, and prints one JSON object.
def adaptive_inputs_of(nn, torch, np, dev, n6):
    y3 = torch.from_numpy(np.stack([1.0 + (np.arange(n6) % 1024) * 2.0 ** -20, np.ones(n6), np.ones(n6)])).to(dev)
    y16 = torch.from_numpy(1.0 + np.arange(16)[None, :] / 16 + ((np.arange(n6) % 1024) * 2.0 ** -20)[:, None]).to(dev)
    return (("C3_dopri54_lorenz_1e6", nn.Rhs.lorenz(), y3, 0, "dopri54", 3), ("C4_tsit54_ring16_1e6", nn.Rhs.ring(0.1), y16, 1, "tsit54", 16))  # (names: BASELINE's sizes; "ivps" holds --adaptive-n)


OPT_IN = (("lean", dict(adv_lean=1)), ("lean_auto_poll", dict(adv_lean=1, adv_auto_poll=1)), ("lean_auto_poll_fp_contract", dict(adv_lean=1, adv_auto_poll=1, fp_contract=1)))


def streamed_opt_in_child(n6):
    import numpy as np
    import torch
    import numericalnim_amd as nn
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    res, errors = {}, {}
    side = torch.cuda.Stream()
    for name, fr, yy, layout, integ, d in adaptive_inputs_of(nn, torch, np, dev, n6):
        _, yfu, cnt = nn.solveODE(fr, yy, [0.0, 1.0], nn.newODEoptions(), integrator=integ, layout=layout, return_counts=True)
        torch.cuda.synchronize()
        iters, acc = int(cnt["steps"].max()), float(cnt["steps"].sum())
        per_step = 8 * (2 * d + 4)
        optin = {}
        side.wait_stream(torch.cuda.current_stream())
        for tag, knobs in OPT_IN:
            try:
                with nn.tuning(**knobs), torch.cuda.stream(side):
                    bo, yo, lo = None, None, 0
                    for _ in range(3):
                        yw = yy.clone()
                        side.synchronize()
                        c0 = time.perf_counter()
                        yo, lo = nn.adaptiveStream(fr, yw, 0.0, 1.0, nn.newODEoptions(), integrator=integ, layout=layout)
                        side.synchronize()
                        dtw = time.perf_counter() - c0
                        bo = dtw if bo is None or dtw < bo else bo
                dev_abs = float((yo - yfu[-1]).abs().max())
                optin[tag] = {"streamed_ms": bo * 1e3, "streamed_us_per_iteration": bo * 1e6 / iters, "streamed_launches": int(lo),
                              "streamed_GBps": per_step * acc / bo / 1e9, "streamed_frac_of_8TBps": per_step * acc / bo / 8e12,
                              "max_abs_deviation_from_fused": dev_abs,
                              "bitwise_equal_to_fused": bool(torch.equal(yo, yfu[-1])), "within_north_star_tolerance": bool(dev_abs <= 1e-6)}
            except Exception as exc:  # noqa: BLE001
                errors["streamed_opt_in:%s:%s" % (name, tag)] = repr(exc)[:500]
        res[name] = optin
    print(json.dumps({"streamed_opt_in": res, "errors": errors}), flush=True)


def streamed_opt_in_parent(n6, timeout_s):
    """-> ({config name: {tag: figures}}, {error key: text}); never raises"""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child-leg", "streamed_opt_in", "--adaptive-n", str(n6)],
                           capture_output=True, text=True, timeout=timeout_s, env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")})
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"streamed_opt_in"')]
        if not lines:
            return {}, {"streamed_opt_in:child": "rc %s, no result; stderr tail: %s" % (r.returncode, r.stderr[-400:])}
        d = json.loads(lines[-1])
        return d["streamed_opt_in"], d["errors"]
    except Exception as exc:  # noqa: BLE001  (timeout: the child is killed by subprocess.run)
        return {}, {"streamed_opt_in:child": repr(exc)[:500]}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run the same command line under torch.distributed.run, one rank per GPU, on this
    node (127.0.0.1, a port the kernel just handed out).  The children inherit stdout / stderr: rank 0's JSON line is this process's."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # this platform's driver shares device memory across processes by dmabuf only
    env.setdefault("OMP_NUM_THREADS", "1")             # what torchrun would set anyway (and warn about)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.child_leg:
        assert args.child_leg == "streamed_opt_in", args.child_leg
        return streamed_opt_in_child(int(args.adaptive_n))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))
    import numpy as np
    import torch
    import torch.distributed as dist
    import numericalnim_amd as nn
    from numericalnim_amd import distributed as nd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher's WORLD_SIZE is %d: start one rank per GPU (or leave WORLD_SIZE unset "
                         "and bench.py starts its own ranks)" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    if args.backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit("bench.py: %d ranks over RCCL need %d devices, this node has %d (one rank per GPU; --backend gloo lets ranks share "
                         "a device to exercise the path)" % (world, world, torch.cuda.device_count()))
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
    rank_elapsed = [elapsed]
    world_reported = 1
    if use_dist:
        per_rank = torch.zeros(world, dtype=torch.float64, device=dev)   # every rank's wall time of the timed region, in rank order
        per_rank[rank] = elapsed
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
        rank_elapsed = [float(v) for v in per_rank.cpu()]
        elapsed = max(rank_elapsed)                                       # the job is as slow as its slowest rank
        world_reported = dist.get_world_size()                            # what the communicator says, not what the command line asked for
        assert world_reported == world == args.gpus, (world_reported, world, args.gpus)

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
            # the counters belong to ONE instantiation of the kernel: attach them only if it is the one this run launched
            import ctypes, re
            vec, mode = ctypes.c_int(0), ctypes.c_int(0)
            assert nn._lib.lib().nnhip_ode_rk4_stream_variant(n, 0 if args.pingpong else 1, ctypes.byref(vec), ctypes.byref(mode)) == 0
            m = re.search(r"rk4_stream_vec_kernel<.*?, (?:false|true), (\d+), (\d+)>", pm["rk4_stream"]["kernel"])
            launched = "rk4_stream_vec_kernel<RhsNegY<1>, false, %d, %d>" % (vec.value, mode.value)
            if not m or (int(m.group(1)), int(m.group(2))) != (vec.value, mode.value):
                traffic = None
                traffic_src = "NOT ATTACHED: profiles/pmc_traffic.json was collected on %s, this run launched %s" % (pm["rk4_stream"]["kernel"], launched)
            else:
                traffic_src += "; variant checked: this run launched %s, the profiled kernel" % launched
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
        "world_size_reported_by_backend": world_reported,
        "ms_per_step_per_rank": {"min": min(rank_elapsed) * 1e3 / args.steps, "max": max(rank_elapsed) * 1e3 / args.steps},
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
        try:  # an informational leg must not cost the run its line: a failure is reported under informational_errors
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
        except Exception as exc:  # noqa: BLE001
            out.setdefault("informational_errors", {})['fused_solve'] = repr(exc)[:500]

    # ---- informational: the same kernel on a batch that cannot live in the 256 MiB Infinity Cache (1 GB of ping-pong state) ----
    if n > 20_000_000:  # the batch itself is beyond the Infinity Cache
        out["roofline"]["achieved_hbm_only"], out["roofline"]["frac_hbm_only"] = achieved, achieved / 8000.0
    if not args.no_fused and world == 1 and n <= 20_000_000:
        try:  # an informational leg must not cost the run its line: a failure is reported under informational_errors
            nb = int(args.beyond_cache_n)
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
        except Exception as exc:  # noqa: BLE001
            out.setdefault("informational_errors", {})['beyond_infinity_cache'] = repr(exc)[:500]

    # ---- informational: BASELINE.json's adaptive configs C3 / C4 (1e6 IVPs / systems), fused and through the HBM-resident loop ----
    adaptive_inputs = {}
    if not args.no_fused and world == 1 and args.adaptive_n >= 1:
        try:  # an informational leg must not cost the run its line: a failure is reported under informational_errors
            cfg = {}
            n6 = int(args.adaptive_n)
            side = torch.cuda.Stream()
            for name, fr, yy, layout, integ, d in adaptive_inputs_of(nn, torch, np, dev, n6):
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
                             "streamed_launches": int(_l), "ivps": n6,
                             "streamed_bytes_per_step": per_step, "streamed_GBps": per_step * float(cnt["steps"].sum()) / best / 1e9,
                             "streamed_frac_of_8TBps": per_step * float(cnt["steps"].sum()) / best / 8e12,   # whole loop incl. host polling and speculative launches, algorithmic bytes
                             "streamed_bitwise_equal_to_fused": bool(torch.equal(ys, yfu[-1])),
                             "streamed_kernels": "general advance kernels, polling groups of 8 (the library's defaults: the configuration with a hardware record)",
                             "accepted_steps": int(cnt["steps"].sum()), "fused_ivps_per_s": n6 / (e0.elapsed_time(e1) / 3 * 1e-3),
                             "fused_accepted_steps_per_s": float(cnt["steps"].sum()) / (e0.elapsed_time(e1) / 3 * 1e-3)}
                adaptive_inputs[name] = (fr, yy, layout, integ, d, yfu[-1])
            # the opt-in settings of the same loop (the lean kernels — same bits —, + the library's own polling schedule, + their FMA-contracted build — within north_star's
            # 1e-6, not bit-equal): no hardware record, so in a child process with a timeout (streamed_opt_in_parent above); a failure there is an informational_errors entry
            optin, oerr = streamed_opt_in_parent(n6, float(os.environ.get("NNHIP_BENCH_CHILD_TIMEOUT", "300")))
            for name in cfg:
                cfg[name]["streamed_opt_in"] = optin.get(name, {})
            if oerr:
                out.setdefault("informational_errors", {}).update(oerr)
            try:  # static companion figures (not measured in this run): what a wavefront of the streamed kernel executes, counted on the library's code object
                dyn = json.load(open(os.path.join(ROOT, "profiles", "r06_isa_dynamic_counts.json")))["kernels"]
                for name, key, ck, lanes_per_unit in (("C3_dopri54_lorenz_1e6", "streamed_c3", "c3", 1.0), ("C4_tsit54_ring16_1e6", "streamed_c4", "c4", 4.0)):
                    waves_per_simd = n6 * lanes_per_unit / 64.0 / 1024.0          # 256 CUs x 4 SIMDs
                    floor = lambda valu: valu * 4.0 * waves_per_simd / 2.4e3      # noqa: E731  4 issue cycles per wave-wide FP64 / VALU instruction
                    g, l, c = dyn[key]["valu_per_wave"]["general"], dyn[key]["valu_per_wave"]["lean"], dyn["streamed_contracted"]["valu_per_wave"][ck]
                    cfg[name]["streamed_kernel_static"] = {
                        "valu_per_wave": {"general": g, "lean": l, "lean_fp_contract": c},
                        "fp64_per_wave": {"general": dyn[key]["general"]["valu_f64"], "lean": dyn[key]["lean"]["valu_f64"], "lean_fp_contract": dyn["streamed_contracted"][ck]["valu_f64"]},
                        "valu_issue_floor_us_at_2.4GHz": {"general": floor(g), "lean": floor(l), "lean_fp_contract": floor(c)},
                        "hbm_floor_us_at_6.3TBps": cfg[name]["streamed_bytes_per_step"] * n6 / 6.3e6,
                        "source": "profiles/r06_isa_dynamic_counts.json (an unpolled launch; tools/gfx950_isa_interp.py on the library's code objects; round 4's PMC on the general C4 kernel: 587 in a polled launch, counted 591)"}
            except Exception:  # noqa: BLE001
                pass
            out["adaptive_configs"] = cfg
        except Exception as exc:  # noqa: BLE001
            out.setdefault("informational_errors", {})['adaptive_configs'] = repr(exc)[:500]

    # ---- informational: what bit parity costs the FP64-VALU-bound fused kernels.  The default build never contracts a*b+c (every
    # operation rounds once, in the reference's order: the results ARE the reference's bits); the opt-in knob "fp_contract" runs the
    # same kernels compiled with FMA contraction: within north_star's tolerance (1e-10 fixed-step / 1e-6 adaptive), not bit-equal.
    if not args.no_fused and world == 1:
        try:  # an informational leg must not cost the run its line: a failure is reported under informational_errors
            L = nn._lib.lib()

            def timed(fn, reps=3):
                fn(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    r = fn()
                e1.record(); torch.cuda.synchronize()
                return e0.elapsed_time(e1) / reps, r

            trade = {}
            legs = [("C2_rk4_neg_y", lambda: nn.solveODE(f, y0, [0.0, t_end], opt, integrator="rk4")[1][-1], 1e-10)]
            if "C4_tsit54_ring16_1e6" in adaptive_inputs:
                fr, yy, layout, integ, d, _y = adaptive_inputs["C4_tsit54_ring16_1e6"]
                legs.append(("C4_tsit54_ring16_1e6", lambda: nn.solveODE(fr, yy, [0.0, 1.0], nn.newODEoptions(), integrator=integ, layout=layout)[1][-1], 1e-6))
            for name, fn, tol in legs:
                exact_ms, ye = timed(fn)
                try:
                    L.nnhip_tune_set(b"fp_contract", 1)
                    fast_ms, yc = timed(fn)
                finally:
                    L.nnhip_tune_set(b"fp_contract", 0)
                dev_abs = float((ye - yc).abs().max())
                trade[name] = {"bit_exact_ms": exact_ms, "contracted_ms": fast_ms, "speedup": exact_ms / fast_ms, "max_abs_deviation": dev_abs,
                               "north_star_tolerance": tol, "within_tolerance": bool(dev_abs <= tol)}  # (reported, not asserted: an informational leg must not cost the line)
            out["fused_solve_fp_contract"] = trade
        except Exception as exc:  # noqa: BLE001
            out.setdefault("informational_errors", {})['fused_solve_fp_contract'] = repr(exc)[:500]

    # ---- informational: batches whose members take different step sequences (every reference call is its own, ode.nim:589-591) ----
    # 1e6 Van der Pol IVPs with their own stiffness in random order: as handed over / binned below the boundary (automatic probe; the caller's key), and
    # 1e6 separate calls with their own tEnd: in the caller's order / longest span first.  All must equal the plain solves bit for bit.
    if not args.no_fused and world == 1 and args.adaptive_n >= 1:
        try:  # an informational leg must not cost the run its line: a failure is reported under informational_errors
            n6 = int(args.adaptive_n)
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
        except Exception as exc:  # noqa: BLE001
            out.setdefault("informational_errors", {})['heterogeneous_batches'] = repr(exc)[:500]

    # ---- CPU baseline: the oracle on this box's host cores (cpu_baseline_c2 / cpu_baseline_adaptive above) -------------------------
    if not args.no_cpu_baseline and world == 1:
        from oracle import oracle as O
        ncores = os.cpu_count() or 1
        k = min(int(args.cpu_sample), n)
        cb, check, k = cpu_baseline_c2(O, nd.c2_y0_numpy, n, nsteps, dt, t_end, int(args.cpu_sample), args.cpu_ivps_per_thread, ncores,
                                       gpu_first=None if args.no_check else yf[:k].cpu().numpy())
        out["cpu_baseline"] = cb
        if check is not None:
            out["parity_max_abs_err_vs_oracle"] = check
            out["parity_checked_ivps"] = k
        for name, (fr, yy, layout, integ, d, y_gpu) in adaptive_inputs.items():
            try:
                n1 = int(args.cpu_adaptive_sample)
                out["adaptive_configs"][name]["cpu_baseline"] = cpu_baseline_adaptive(
                    O, name, yy.cpu().numpy(), layout, integ, d, (y_gpu[:, :n1] if layout == 0 else y_gpu[:n1]).cpu().numpy(), n1, ncores)
            except Exception as exc:  # noqa: BLE001
                out.setdefault("informational_errors", {})["cpu_baseline_" + name] = repr(exc)[:500]
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
