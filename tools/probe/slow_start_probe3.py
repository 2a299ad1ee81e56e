#!/usr/bin/env python3
"""Which part of an upload puts the next launches into the slow mode?  mol_rhs() uploads 12.6 MB into a buffer the step
kernel never touches; set_flux/set_pb are small uploads; set_batch rewrites the state buffer itself."""
import os
import sys

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
for rep in range(5):
    s.mol_rhs(c0)
    burst('after mol_rhs (12.6 MB H2D elsewhere + kernels + D2H)')
for rep in range(5):
    s.set_batch(c0, pb, vz, fl)
    burst('after set_batch')
