"""Multi-GPU plumbing: the batch of independent IVPs shards as contiguous index ranges, one rank per GPU,
no exchange during integration; one all-gather (RCCL over xGMI when the backend is "nccl") reassembles
the final-state tensor.  Nothing couples trajectories in the reference (one solveODE call per IVP,
ode.nim:589), so this is the only collective on the path."""
import numpy as np


def shard_range(n_total, rank, world):
    """Contiguous index range [lo, hi) of rank `rank`: lo = n_total*rank//world."""
    return n_total * rank // world, n_total * (rank + 1) // world


def c2_y0_numpy(lo, hi):
    """BASELINE.json config C2/C5 initial states: y0[i] = 1 + (i mod 2^20) * 2^-20 for global index i."""
    i = np.arange(lo, hi, dtype=np.int64)
    return 1.0 + (i % (1 << 20)).astype(np.float64) * 2.0 ** -20


def c2_y0_torch(lo, hi, device):
    import torch
    i = torch.arange(lo, hi, dtype=torch.int64, device=device)
    return 1.0 + (i % (1 << 20)).to(torch.float64) * 2.0 ** -20


def all_gather_states(local, group=None):
    """All-gather equal-sized per-rank final states (last axis = IVP index) into the global tensor."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if local.dim() == 1:
        out = torch.empty(world * local.shape[0], dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    # [..., n_local] -> gather along a new leading axis, then move it next to the IVP axis
    parts = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(parts, local.contiguous(), group=group)
    return torch.cat(list(parts.unbind(0)), dim=-1)
