#!/usr/bin/env python3
"""Follow-up of slow_start_probe.py: which host-side event puts the next fused launches of the 3-waves-per-lane kernel into
the slow mode (8-9 us/step instead of 5.6)?"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from catint_amd import _capi                      # noqa: E402
from catint_amd.synthetic import make_batch       # noqa: E402

B = 1024
prob, c0, pb, vz, fl = make_batch(B, 3, 512, seed=0, dt_factor=1e-5)
s = _capi.PnpSolver(3, 512, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Crank-Nicolson', batch_capacity=B)
s.set_batch(c0, pb, vz, fl)


def burst(tag, n=5, spl=256):
    out = []
    for _ in range(n):
        s.timer_start()
        s.step(spl, spl)
        out.append(s.timer_stop() / spl * 1e3)
    print('%-50s %s' % (tag, ' '.join('%.2f' % x for x in out)), flush=True)


for _ in range(100):
    s.step(256, 256)
s.synchronize()
burst('steady state')
for rep in range(3):
    s.get_surface()
    burst('after get_surface (small kernel + D2H)')
for rep in range(3):
    s.set_flux(fl)
    burst('after set_flux (small H2D)')
for rep in range(3):
    s.set_pb(pb, vz)
    burst('after set_pb (small H2D)')
for rep in range(3):
    c_now = s.get_state(potential=False)
    c_now = c_now[0] if isinstance(c_now, tuple) else c_now
    burst('after get_state (12 MB D2H)')
for rep in range(3):
    s.set_batch(c_now.reshape(B, -1), pb, vz, fl)
    burst('after set_batch(current state)')
for rep in range(3):
    s.set_batch(c0, pb, vz, fl)
    burst('after set_batch(c0)')
for rep in range(3):
    s.set_batch(c0, pb, vz, fl)
    s.step(2, 1)
    burst('after set_batch(c0) + 2 one-step launches')
print('--- cures')
for rep in range(6):
    s.set_batch(c0, pb, vz, fl)
    burst('after set_batch(c0)')
for rep in range(6):
    s.set_batch(c0, pb, vz, fl)
    s.get_surface(); s.get_surface()
    burst('after set_batch(c0) + 2 get_surface')
for rep in range(6):
    s.set_batch(c0, pb, vz, fl)
    time.sleep(0.02)
    burst('after set_batch(c0) + 20 ms sleep')
for rep in range(6):
    s.set_batch(c0, pb, vz, fl)
    s.step(1, 1)
    burst('after set_batch(c0) + 1 one-step launch')
for rep in range(6):
    s.set_batch(c0, pb, vz, fl)
    s.step(8, 8)
    burst('after set_batch(c0) + 1 eight-step launch')
