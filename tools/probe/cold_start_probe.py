#!/usr/bin/env python3
"""Per-launch time of the fused headline launch (B=1024, N=3, nx=512, 256 steps) over the first seconds of a process:
does a fresh box / fresh process run slower for a while (clock ramp) or is a slow run slow throughout (placement)?"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from catint_amd import _capi                      # noqa: E402
from catint_amd.synthetic import make_batch       # noqa: E402

if os.environ.get('PROBE_TORCH'):
    import torch
    torch.cuda.init()
    _keep = torch.zeros(int(os.environ.get('PROBE_TORCH_MB', '64')) << 18, device='cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
prob, c0, pb, vz, fl = make_batch(B, 3, 512, seed=0, dt_factor=1e-5)
s = _capi.PnpSolver(3, 512, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Crank-Nicolson', batch_capacity=B)
s.set_batch(c0, pb, vz, fl)
t0 = time.time()
out = []
for i in range(n):
    s.timer_start()
    s.step(256, 256)
    ms = s.timer_stop()
    out.append((time.time() - t0, ms))
ms = np.array([m for _, m in out])
print('launches %d, us/step: first %.2f, min %.2f, median %.2f, max %.2f' % (n, ms[0] / 256 * 1e3, ms.min() / 256 * 1e3,
                                                                             np.median(ms) / 256 * 1e3, ms.max() / 256 * 1e3))
for i in list(range(0, 10)) + list(range(10, n, max(1, n // 25))):
    print('t=%6.3f s  %.2f us/step' % (out[i][0], out[i][1] / 256 * 1e3))
