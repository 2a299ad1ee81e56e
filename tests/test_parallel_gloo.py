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


ALIGN_WORKER = r'''
import os, sys, time
import torch.distributed as dist
sys.path.insert(0, %r)
from catint_amd.parallel import aligned_start, gather_numbers
dist.init_process_group('gloo')
rank = dist.get_rank()
time.sleep(0.05 * rank)                    # the ranks arrive at different times ...
dist.barrier()
t0, deadline = aligned_start(dist)         # ... and leave together
assert t0 >= deadline
tab = gather_numbers([t0, rank], dist)
assert tab.shape == (2, 2) and list(tab[:, 1]) == [0.0, 1.0]
assert tab[:, 0].max() - tab[:, 0].min() < 0.05, tab
dist.barrier()
dist.destroy_process_group()
print('rank', rank, 'ok')
''' % ROOT


def test_aligned_start_of_a_timed_region():
    """bench.py's timed regions start on a deadline all ranks agree on (same node, same monotonic clock), not on barrier exit."""
    port = free_port()
    procs = [subprocess.Popen([sys.executable, '-c', ALIGN_WORKER], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                                       MASTER_PORT=str(port))) for r in range(2)]
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0 and 'ok' in o, e[-2000:]


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


def test_spawn_ranks_sets_the_rendezvous_environment(tmp_path):
    """The launcher behind `python bench.py --gpus N`: N processes, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, output per rank."""
    from catint_amd.parallel import spawn_ranks
    code = ("import os; print(os.environ['RANK'], os.environ['LOCAL_RANK'], os.environ['WORLD_SIZE'], os.environ['MASTER_ADDR'], "
            "os.environ['MASTER_PORT'], os.environ['HSA_ENABLE_IPC_MODE_LEGACY'], os.environ.get('EXTRA'))")
    files = [open(tmp_path / ('r%d.txt' % r), 'w') for r in range(3)]
    rc = spawn_ranks(3, [sys.executable, '-c', code], extra_env={'EXTRA': 'x'}, stdout=files, timeout=60)
    for f in files:
        f.close()
    assert rc == 0
    rows = [open(tmp_path / ('r%d.txt' % r)).read().split() for r in range(3)]
    assert [row[0] for row in rows] == ['0', '1', '2'] and [row[1] for row in rows] == ['0', '1', '2']
    assert all(row[2] == '3' and row[3] == '127.0.0.1' and row[6] == 'x' for row in rows)
    assert len({row[4] for row in rows}) == 1 and int(rows[0][4]) > 0          # one port for all ranks
    # a failing rank is reported
    assert spawn_ranks(2, [sys.executable, '-c', "import os, sys; sys.exit(3 if os.environ['RANK'] == '1' else 0)"], timeout=60) == 3


def test_bench_gpus_flag_reaches_the_launcher(monkeypatch):
    """bench.py --gpus N with no RANK in the environment must hand over to spawn_ranks with its own command line, before torch / HIP."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_spawn(n, argv, **kw):
        seen['n'], seen['argv'] = n, argv
        return 0

    import catint_amd.parallel as par
    monkeypatch.setattr(par, 'spawn_ranks', fake_spawn)
    monkeypatch.delenv('RANK', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '7', '--warmup', '2'])
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert seen['n'] == 4 and seen['argv'][1].endswith('bench.py') and seen['argv'][2:] == ['--gpus', '4', '--steps', '7', '--warmup', '2']
