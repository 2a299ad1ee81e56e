#!/usr/bin/env python3
"""One GPU's share of configs[3] (32768 x 6 x 1024, 2.1 GB), one launch per step: rate over time from a cold start."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from catint_amd.synthetic import make_batch
from catint_amd.host import solver_from_problem
B, N, nx = 32768, 6, 1024
p2, c2, pb2, vz2, fl2 = make_batch(B, N, nx, seed=55, phi_max=0.025, dt_factor=1e-5)
s = solver_from_problem(p2, 'Crank-Nicolson', batch_capacity=B); s.set_batch(c2, pb2, vz2, fl2)
alg = 16.0 * (N + 1) * nx * B


def series(tag, n=12, k=8, spl=1):
    out = []
    for _ in range(n):
        s.timer_start(); s.step(k, spl); ms = s.timer_stop()
        out.append(alg * k / (ms * 1e-3) / 8e12)
    print('%-44s %s' % (tag, ' '.join('%.3f' % v for v in out)), flush=True)


series('cold start, 8 per-step launches each')
series('continuing')
series('continuing')
time.sleep(0.5)
series('after 0.5 s idle')
series('fused 32 steps per launch', n=8, k=32, spl=32)
print('lanes ok', int((s.get_status() == 0).sum()))
s.close()
