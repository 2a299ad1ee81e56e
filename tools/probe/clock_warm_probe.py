#!/usr/bin/env python3
"""Per-step launches at B = 8192 (168 MB of state) right after 0.2 s of load from another handle: transient or sustained?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from catint_amd.synthetic import make_batch
from catint_amd.host import solver_from_problem
p1, c1, pb1, vz1, fl1 = make_batch(1024, 3, 512, seed=1000, phi_max=0.025, dt_factor=1e-5)
w = solver_from_problem(p1, 'Crank-Nicolson', batch_capacity=1024); w.set_batch(c1, pb1, vz1, fl1)
p2, c2, pb2, vz2, fl2 = make_batch(8192, 3, 512, seed=77, phi_max=0.025, dt_factor=1e-5)
s = solver_from_problem(p2, 'Crank-Nicolson', batch_capacity=8192); s.set_batch(c2, pb2, vz2, fl2)
alg = 16.0 * 4 * 512 * 8192


def series(tag, n=12, k=20):
    out = []
    for _ in range(n):
        s.timer_start(); s.step(k, 1); ms = s.timer_stop()
        out.append(alg * k / (ms * 1e-3) / 8e12)
    print('%-40s %s' % (tag, ' '.join('%.3f' % v for v in out)), flush=True)


s.step(64, 1); s.synchronize()
series('after 64 warm-up launches')
time.sleep(0.5)
series('after 0.5 s idle')
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.2:
    for _ in range(4):
        w.step(256, 256)
    w.synchronize()
series('after 0.2 s of fused load (other handle)')
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    s.step(50, 1); s.synchronize()
series('after 0.5 s of its own per-step launches')
s.close(); w.close()
