#!/usr/bin/env python3
"""Health of the headline trajectory over many timesteps: lanes with a sticky status, min concentration, finite check."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from catint_amd import _capi                      # noqa: E402
from catint_amd.synthetic import make_batch       # noqa: E402

B = 1024
dtf = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-5
prob, c0, pb, vz, fl = make_batch(B, 3, 512, seed=0, dt_factor=dtf)
s = _capi.PnpSolver(3, 512, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Crank-Nicolson', batch_capacity=B)
s.set_batch(c0, pb, vz, fl)
done = 0
for n in [256, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536]:
    s.step(n, 256)
    done += n
    st = s.get_status()
    c = s.get_state(potential=False)
    c = c[0] if isinstance(c, tuple) else c
    worst = np.unravel_index(np.argmin(c), c.shape)
    print('steps %6d: status counts %s, finite %s, min c %.3e at lane %d species %d point %d (phiM %.3f V), bulk min %.3e'
          % (done, np.bincount(st, minlength=4).tolist(), bool(np.isfinite(c).all()), c.min(), worst[0], worst[1], worst[2], pb[worst[0], 0],
             c0.min()), flush=True)
