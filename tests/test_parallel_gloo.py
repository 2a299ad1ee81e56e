"""N>1 path on CPU: two gloo ranks shard a batch and gather the observable table (no GPU needed)."""
import os
import socket
import subprocess
import sys

import numpy as np

from catint_amd.parallel import shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, %r)
from catint_amd.parallel import shard_bounds, gather_observables
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
B, nobs = int(sys.argv[1]), 5
lo, hi = shard_bounds(B, world, rank)
lanes = np.arange(lo, hi)
local = np.stack([np.sin(lanes * (j + 1.0)) + j for j in range(nobs)], axis=1)   # stands in for the per-lane solve
full = gather_observables(local, B, dist)
ref = np.stack([np.sin(np.arange(B) * (j + 1.0)) + j for j in range(nobs)], axis=1)
assert full.shape == (B, nobs), full.shape
assert np.array_equal(full, ref)
dist.barrier()
dist.destroy_process_group()
print('rank', rank, 'ok', lo, hi)
''' % ROOT


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(B, world=2):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, '-c', WORKER, str(B)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    return [o for o, _ in outs]


def test_shard_bounds_cover_batch():
    for B in (1, 2, 7, 1024, 1025):
        for world in (1, 2, 3, 8):
            blocks = [shard_bounds(B, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == B
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gather_even_and_ragged():
    outs = run_world(64)
    assert all('ok' in o for o in outs)
    outs = run_world(37)     # ragged: 19 + 18 lanes
    assert all('ok' in o for o in outs)
