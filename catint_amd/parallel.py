"""Batch sharding over the GPUs of one node and the single exchange of the path: the gather of the
polarization observables.  Operating points are independent (the reference treats them as an MPI task
farm, catint/calculator.py:209-212; gather of the per-rank result dicts catint/catint_io.py:167-178), so
each rank owns one contiguous block of lanes and nothing is communicated during the solve."""
import numpy as np


def shard_bounds(B, world, rank):
    """Contiguous block [lo, hi) of lanes for `rank`; the first B % world ranks get one extra lane."""
    base, rem = divmod(int(B), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_observables(local, B, dist=None, device=None):
    """all_gather of the per-lane observable table [B_local, n_obs] -> [B, n_obs] in lane order on every
    rank.  `dist` is torch.distributed (backend nccl = RCCL over xGMI on the GPU box, gloo in CPU tests)."""
    local = np.ascontiguousarray(local, dtype=np.float64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    nmax = max(shard_bounds(B, world, r)[1] - shard_bounds(B, world, r)[0] for r in range(world))
    buf = torch.zeros((nmax, local.shape[1]), dtype=torch.float64, device=device)
    buf[:local.shape[0]] = torch.from_numpy(local).to(buf.device)
    out = torch.empty((world * nmax, local.shape[1]), dtype=torch.float64, device=buf.device)
    dist.all_gather_into_tensor(out, buf)     # concatenated form: accepted by both RCCL and gloo
    out = out.cpu().numpy().reshape(world, nmax, local.shape[1])
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(B, world, r)
        parts.append(out[r, :hi - lo])
    return np.concatenate(parts, axis=0)
