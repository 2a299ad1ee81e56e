#!/usr/bin/env python3
"""Why are the first two fused launches after pnp_set_batch 1.6x slower for the 3-waves-per-lane variant?  Separates
'state dependent' (restart from c0) from 'idle dependent' (host sleep) from 'process start'."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from catint_amd import _capi                      # noqa: E402
from catint_amd.synthetic import make_batch       # noqa: E402

B = 1024
prob, c0, pb, vz, fl = make_batch(B, 3, 512, seed=0, dt_factor=float(os.environ.get('DT_FACTOR', '1e-5')))
s = _capi.PnpSolver(3, 512, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Crank-Nicolson', batch_capacity=B)
s.set_batch(c0, pb, vz, fl)


def burst(tag, n=6, spl=256):
    out = []
    for _ in range(n):
        s.timer_start()
        s.step(spl, spl)
        out.append(s.timer_stop() / spl * 1e3)
    print('%-44s %s' % (tag, ' '.join('%.2f' % x for x in out)), flush=True)


burst('process start')
for _ in range(200):
    s.step(256, 256)
s.synchronize()
burst('after 200 more launches')
time.sleep(0.05)
burst('after 50 ms of idle')
c_now = s.get_state(potential=False)
c_now = c_now[0] if isinstance(c_now, tuple) else c_now
burst('after a read-back (sync + D2H copy)')
s.set_batch(c_now.reshape(B, -1), pb, vz, fl)
burst('after set_batch(current state)')
s.set_batch(c0, pb, vz, fl)
burst('after set_batch(initial state c0)')
burst('... continued')
s.set_batch(c0, pb, vz, fl)
burst('after set_batch(c0), 64-step launches', n=12, spl=64)
s.set_batch(c0, pb, vz, fl)
burst('after set_batch(c0), 8-step launches', n=12, spl=8)
burst('... then 256-step launches')
s.set_batch(c0, pb, vz, fl)
burst('after set_batch(c0), 1024-step launches', n=4, spl=1024)
s.synchronize()
s.get_status()
burst('after get_status (sync + 4 KB D2H)')
s.synchronize()
burst('after synchronize only')
rng = np.random.default_rng(1)
s.set_batch(c0 * (1.0 + 1e-3 * rng.standard_normal(c0.shape)), pb, vz, fl)
burst('after set_batch(c0 with 1e-3 noise)', n=8)
s.set_batch(c0, pb, vz, fl)
burst('after set_batch(c0) again', n=8)
pb2 = pb.copy(); pb2[:, 0] = 0.0
s.set_batch(c0, pb2, np.zeros(B), fl)
burst('after set_batch(c0), wall potential 0', n=8)
