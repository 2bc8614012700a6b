#!/usr/bin/env python3
"""Sweep the variants of the headline RK4 step-streaming kernel on the GPU (interleaved rounds in ONE process,
median of per-variant event timings) and print a table.  Run via gpurun."""
import itertools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import numericalnim_amd as nn  # noqa: E402
from numericalnim_amd import distributed as nd  # noqa: E402


def main():
    L = nn._lib.lib()
    dev = torch.device("cuda:0")
    nsteps, dt = 200, 2.0 ** -10
    opt = nn.newODEoptions(dt=dt)
    f = nn.Rhs.neg_y()
    sizes = [int(float(x)) for x in (sys.argv[1:] or ["1e7", "6.4e7"])]
    res = []
    for n in sizes:
        y0 = nd.c2_y0_torch(0, n, dev)
        y = y0.clone()
        sc = torch.empty_like(y0)
        variants = list(itertools.product([1, 2, 4, 8], [0, 1, 2, 3], [False, True]))
        times = {v: [] for v in variants}
        for rnd in range(4):
            for v in variants:
                vec, mode, pp = v
                L.nnhip_tune_set(b"rk4_stream_vec", vec)
                L.nnhip_tune_set(b"rk4_stream_mode", mode)
                y.copy_(y0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                nn.fixedStream(f, y, 0.0, nsteps * dt, opt, integrator="rk4", scratch=sc if pp else None)
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    times[v].append(e0.elapsed_time(e1) * 1e3 / nsteps)
        for v in variants:
            us = sorted(times[v])[len(times[v]) // 2]
            res.append(dict(n=n, vec=v[0], mode=v[1], pingpong=v[2], us_per_launch=us, GBps=16.0 * n / us / 1e3))
    L.nnhip_tune_set(b"rk4_stream_auto", 1)
    res.sort(key=lambda r: (r["n"], -r["GBps"]))
    for r in res:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
