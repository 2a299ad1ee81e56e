"""GPU: the N > 1 path with REAL solves -- two ranks (one process each, as bench.py --gpus N launches them) share the one GPU of the
test box, each advances its contiguous block of lanes with the HIP path and the surface observables are gathered with
torch.distributed (gloo, and RCCL = backend 'nccl' where two ranks may share a device); every rank must hold the table a single
rank computes for the whole batch.  (tests/test_parallel_gloo.py covers the gather alone on the CPU.)"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
import torch
import torch.distributed as dist
from catint_amd.synthetic import make_batch
from catint_amd.host import solver_from_problem
from catint_amd.parallel import shard_bounds, gather_observables
backend = sys.argv[2]
torch.cuda.set_device(0)
if backend == 'nccl':
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
else:
    dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
B, N, nx, steps = int(sys.argv[1]), 3, 130, 12
p, c0, pb, vz, fl = make_batch(B, N, nx, seed=5, phi_max=0.02, dt_factor=1e-4)
lo, hi = shard_bounds(B, world, rank)
with solver_from_problem(p, 'Crank-Nicolson', batch_capacity=hi - lo) as s:
    s.set_batch(c0[lo:hi], pb[lo:hi], vz[lo:hi], fl[lo:hi])
    s.step(steps)
    cs, vs, es = s.get_surface()
    assert (s.get_status() == 0).all()
local = np.concatenate([cs, vs[:, None], es[:, None]], axis=1)
dev = torch.device('cuda', 0) if backend == 'nccl' else torch.device('cpu')
full = gather_observables(local, B, dist, device=dev)
with solver_from_problem(p, 'Crank-Nicolson', batch_capacity=B) as s:       # the whole batch on one handle
    s.set_batch(c0, pb, vz, fl)
    s.step(steps)
    cs, vs, es = s.get_surface()
ref = np.concatenate([cs, vs[:, None], es[:, None]], axis=1)
assert full.shape == (B, N + 2) and np.array_equal(full, ref), np.abs(full - ref).max()
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


def test_one_rank_rccl_gathers_on_a_device_tensor():
    """The RCCL leg itself on the one GPU of the test box: backend 'nccl' with a world of ONE rank, started as a fresh process --
    init_process_group with a device id, gather_observables -> all_gather_into_tensor on a DEVICE fp64 tensor (the helper runs the
    collective for a single rank too), barrier, destroy_process_group.  What bench.py --gpus N does on every rank
    (reference: the MPI task farm catint/calculator.py:209-212 and its gather catint/catint_io.py:167-178)."""
    procs, outs = run_world(48, 'nccl', world=1)
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0 and 'ok' in o, e[-2000:]


def test_bench_under_rccl_with_one_rank_walks_every_collective_of_the_multi_gpu_line():
    """bench.py with backend 'nccl' and a process group of ONE rank (CATINT_FORCE_DIST): barrier, the all-reduce of the aligned start,
    the all-gathers of the timed regions and of the observables, the per-rank configs[3] / configs[4] share records -- every collective
    of `bench.py --gpus 8`, on DEVICE tensors through RCCL, on the one GPU of the test box."""
    import json
    env = dict(os.environ, CATINT_FORCE_DIST='1', CATINT_BENCH_SHARE_DIV='16', RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(free_port()), HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('CATINT_DIST_BACKEND', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '4', '--warmup', '1', '--no-pmc', '--no-cpu-baseline',
                          '--shares-only'], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads(out.stdout)
    assert d['n_gpus'] == 1 and d['gather']['backend'] == 'nccl' and d['start_skew_us'] == 0.0
    for key in ('configs3_share', 'configs4_share'):
        for leg in ('compat_per_step', 'newton'):
            assert 'error' not in d[key][leg] and d[key][leg]['lanes_ok'] == d[key][leg]['lanes_total']


def test_bench_two_ranks_print_one_json_line_with_the_8_gpu_shares():
    """`bench.py --gpus 2` as its own launcher (two gloo ranks on the one GPU; RCCL refuses two ranks on a device): stdout is exactly
    one JSON line, it carries n_gpus = 2, the start skew of the aligned timed region and one GPU's share of configs[3] / configs[4]
    per rank (here on 1/16 of the batch)."""
    import json
    env = dict(os.environ, CATINT_DIST_BACKEND='gloo', CATINT_BENCH_SHARE_DIV='16', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1', '--no-pmc',
                          '--no-cpu-baseline'], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads(out.stdout)                       # the WHOLE of stdout is one JSON document
    assert d['n_gpus'] == 2 and len(d['per_rank_timesteps_per_s']) == 2 and d['start_skew_us'] >= 0.0
    for key in ('configs3_share', 'configs4_share'):
        for leg in ('compat_per_step', 'newton'):
            r = d[key][leg]
            assert 'error' not in r, r
            assert r['lanes_ok'] == r['lanes_total'] and len(r['per_rank_timesteps_per_s']) == 2 and r['value'] > 0


def run_world(B, backend, world=2):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, '-c', WORKER, str(B), backend], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    return procs, outs


@pytest.mark.parametrize('B', [48, 37])
def test_two_ranks_solve_their_shards_and_gather_gloo(B):
    procs, outs = run_world(B, 'gloo')
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0 and 'ok' in o, e[-2000:]


def test_two_ranks_solve_their_shards_and_gather_rccl():
    procs, outs = run_world(48, 'nccl')
    if any(p.returncode != 0 for p in procs) and any('Duplicate GPU' in e or 'invalid usage' in e or 'unhandled' in e for _, e in outs):
        pytest.skip('this RCCL build refuses two ranks on one device')
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0 and 'ok' in o, e[-2000:]
