#!/usr/bin/env python3
"""pnp_set_batch with and without its two trailing trivial dispatches (CATINT_PNP_NO_POST_UPLOAD_DISPATCH=1), shapes that fill the chip
in one round and shapes that do not; 64-step launches timed one by one right after the upload."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from catint_amd.synthetic import make_batch
from catint_amd.host import solver_from_problem
print('CATINT_PNP_NO_POST_UPLOAD_DISPATCH =', os.environ.get('CATINT_PNP_NO_POST_UPLOAD_DISPATCH'))
for (B, N, nx) in [(1024, 3, 512), (960, 3, 512), (8192, 3, 512), (1024, 6, 512), (4096, 6, 1024)]:
    prob, c0, pb, vz, fl = make_batch(B, N, nx, seed=1000, phi_max=0.025, dt_factor=1e-5)
    with solver_from_problem(prob, 'Crank-Nicolson', batch_capacity=B) as s:
        s.set_batch(c0, pb, vz, fl)
        for rep in range(2):
            for _ in range(10):
                s.step(256, 256)
            s.synchronize()
            s.set_batch(c0, pb, vz, fl)
            out = []
            for i in range(5):
                s.timer_start(); s.step(64, 64); out.append(s.timer_stop() * 1e3 / 64)
            print('B=%5d N=%d nx=%4d after upload: %s us/step' % (B, N, nx, ' '.join('%.2f' % v for v in out)), flush=True)
