"""Batch sharding over the GPUs of one node and the single exchange of the path: the gather of the
polarization observables.  Operating points are independent (the reference treats them as an MPI task
farm, catint/calculator.py:209-212; gather of the per-rank result dicts catint/catint_io.py:167-178), so
each rank owns one contiguous block of lanes and nothing is communicated during the solve."""
import os
import socket
import subprocess
import sys

import numpy as np


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank, world, port, base=None):
    """Environment of one rank of a single-node job (what `torch.distributed.run` would set): the reference's task farm assigns
    descriptor points by `itask % mpi_size != mpi_rank` (catint/calculator.py:209-212); here a rank is a process that owns one GPU."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(port))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL between processes needs it on this driver
    return env


def spawn_ranks(nranks, argv, extra_env=None, stdout=None, stderr=None, timeout=None):
    """Start `nranks` child processes running `argv` (one per GPU, rendezvous on 127.0.0.1) and wait for them; returns the
    largest exit code.  Must be called BEFORE the calling process touches the GPU (no HIP call, no torch.cuda.is_available()):
    children of a GPU-initialised process are off limits on the GPU pool, and the parent has nothing to compute anyway.  Rank r's
    stdout/stderr go to stdout[r] / stderr[r] when lists of files are given, else they are inherited (rank 0 prints the result)."""
    port = free_port()
    procs = []
    for r in range(int(nranks)):
        env = rank_env(r, nranks, port)
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen(list(argv), env=env, stdout=None if stdout is None else stdout[r],
                                      stderr=None if stderr is None else stderr[r]))
    import time
    rc, t0 = 0, time.time()
    try:
        while any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs):      # a failed rank must not leave the others waiting in a collective
                break
            if timeout is not None and time.time() - t0 > timeout:
                rc = 1
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
                rc = max(rc, 1)
            else:
                rc = max(rc, abs(p.returncode))
    return rc


def shard_bounds(B, world, rank):
    """Contiguous block [lo, hi) of lanes for `rank`; the first B % world ranks get one extra lane."""
    base, rem = divmod(int(B), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_observables(local, B, dist=None, device=None):
    """all_gather of the per-lane observable table [B_local, n_obs] -> [B, n_obs] in lane order on every
    rank.  `dist` is torch.distributed (backend nccl = RCCL over xGMI on the GPU box, gloo in CPU tests)."""
    local = np.ascontiguousarray(local, dtype=np.float64)
    if dist is None or not dist.is_initialized():
        return local
    # (a world of ONE rank still runs the collective: that is how a one-GPU box exercises the RCCL leg -- tests/test_gpu_parallel.py)
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


def aligned_start(dist, device=None, margin=0.002):
    """Start a timed region on every rank at the SAME instant instead of at barrier exit: after the barrier the ranks agree on a
    deadline (all-reduce MAX of `now + margin`; the ranks of one node read one CLOCK_MONOTONIC) and spin until it has passed.
    Barrier-exit skew (tens of microseconds between ranks) would otherwise count against a region of a hundred microseconds as lost
    scaling that is not there.  Returns (t_start of this rank, deadline); without a process group: (now, now)."""
    import time
    if dist is None or not dist.is_initialized():
        t = time.perf_counter()
        return t, t
    import torch
    d = torch.tensor([time.perf_counter() + margin], dtype=torch.float64, device=device)
    dist.all_reduce(d, op=dist.ReduceOp.MAX)
    deadline = float(d.cpu()[0])
    while time.perf_counter() < deadline:
        pass
    return time.perf_counter(), deadline


def gather_numbers(values, dist, device=None):
    """[world][len(values)] table of every rank's numbers (all_gather); without a process group: one row."""
    import torch
    mine = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if dist is None or not dist.is_initialized():
        return mine.cpu().numpy()[None, :]
    every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(every, mine)
    return np.stack([e.cpu().numpy() for e in every])
