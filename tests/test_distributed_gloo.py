"""N>1 path on CPU: gloo at world sizes 2, 3 (ragged shards) and 8 (BASELINE config C5's rank count).  The batch shards as contiguous IVP index ranges; the only collective
is the all-gather that reassembles the final-state tensor (numericalnim_amd/distributed.py, used by bench.py).
No GPU here, so each rank integrates its shard with the ORACLE (tests may) and the gathered result must equal
the unsharded oracle run bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, dim, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from numericalnim_amd import distributed as nd
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = nd.shard_range(n_total, rank, world)
    dt = 2.0 ** -6
    if dim == 1:
        y0 = nd.c2_y0_numpy(lo, hi)
        r = O.solve_ode_batch(O.RHS_NEG_Y, [], y0, hi - lo, 0, [0.0, 1.0], O.new_options(dt=dt), "rk4")
        local = torch.from_numpy(r["y"][-1, 0].copy())
    else:
        base = nd.c2_y0_numpy(lo, hi)
        y0 = np.stack([base, np.ones_like(base), np.ones_like(base)])
        r = O.solve_ode_batch(O.RHS_LORENZ, [10.0, 28.0, 8.0 / 3.0], y0, hi - lo, 3, [0.0, 0.25], O.new_options(), "dopri54")
        local = torch.from_numpy(r["y"][-1].copy())  # [3, n_local]
    full = nd.all_gather_states(local, n_total=n_total)
    if rank == 0:
        q.put(full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 512), (3, 500), (8, 1000)], ids=["world2", "world3_ragged", "world8"])
@pytest.mark.parametrize("dim", [1, 3])
def test_shard_integrate_allgather(oracle, dim, world, n_total):
    import torch.multiprocessing as mp
    from numericalnim_amd import distributed as nd
    O = oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, dim, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if dim == 1:
        ref = O.solve_ode_batch(O.RHS_NEG_Y, [], nd.c2_y0_numpy(0, n_total), n_total, 0, [0.0, 1.0], O.new_options(dt=2.0 ** -6), "rk4")["y"][-1, 0]
    else:
        base = nd.c2_y0_numpy(0, n_total)
        ref = O.solve_ode_batch(O.RHS_LORENZ, [10.0, 28.0, 8.0 / 3.0], np.stack([base, np.ones_like(base), np.ones_like(base)]), n_total, 3,
                                [0.0, 0.25], O.new_options(), "dopri54")["y"][-1]
    assert got.shape == ref.shape
    assert np.array_equal(got, ref)


def test_shard_ranges_partition():
    from numericalnim_amd import distributed as nd
    for n in (0, 1, 7, 8, 1000, 10_000_001):
        for w in (1, 2, 3, 8):
            rs = [nd.shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in rs) - min(h - l for l, h in rs) <= 1
