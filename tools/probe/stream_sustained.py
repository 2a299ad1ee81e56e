#!/usr/bin/env python3
"""Sustained rate (fraction of 8 TB/s, algorithmic bytes) of the streaming regime: one GPU's share of configs[3] (2.1 GB) and
N = 3, nx = 512, B = 32768 (671 MB), one launch per step and fused, after the clocks are up.  CATINT_PNP_LIB selects the library."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from catint_amd.synthetic import make_batch
from catint_amd.host import solver_from_problem
import numpy as np
print('lib', os.environ.get('CATINT_PNP_LIB', 'default'))
for (B, N, nx, kf) in [(32768, 6, 1024, 32), (32768, 3, 512, 64)]:
    p, c, pb, vz, fl = make_batch(B, N, nx, seed=55, phi_max=0.025, dt_factor=1e-5)
    with solver_from_problem(p, 'Crank-Nicolson', batch_capacity=B) as s:
        s.set_batch(c, pb, vz, fl)
        alg = 16.0 * (N + 1) * nx * B
        for _ in range(8):
            s.step(8, 1)
        s.synchronize()
        out = []
        for _ in range(8):
            s.timer_start(); s.step(8, 1); ms = s.timer_stop()
            out.append(alg * 8 / (ms * 1e-3) / 8e12)
        fo = []
        for _ in range(4):
            s.timer_start(); s.step(kf, kf); ms = s.timer_stop()
            fo.append(alg * kf / (ms * 1e-3) / 8e12)
        ok = int((s.get_status() == 0).sum())
        print('B=%d N=%d nx=%d  per step: median %.4f (%s)  fused %d: median %.4f  lanes ok %d' % (
            B, N, nx, np.median(out), ' '.join('%.3f' % v for v in out), kf, np.median(fo), ok), flush=True)
