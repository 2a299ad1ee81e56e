#!/usr/bin/env python3
"""Diagnosis (library built with -DPNP_SWEEP_STAMPS): where a Newton iteration of the sweep kernel spends its cycles."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
os.environ['CATINT_NEWTON_KERNEL'] = 'sweep'
from catint_amd import _capi                      # noqa: E402
from catint_amd.synthetic import make_batch       # noqa: E402

N, nx, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
prob, c0, pb, vz, fl = make_batch(B, N, nx, seed=0, phi_max=0.2, dt_factor=0.1)
s = _capi.PnpSolver(N, nx, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Newton', batch_capacity=B)
radii = [4.1e-10, 3.6e-10, 3.3e-10, 3e-10, 3e-10, 3e-10, 4.5e-10, 3.5e-10][:N]
s.set_newton(wall_bc='stern', stern_capacitance=0.2, tol=1e-8, mpb_radius=radii)
s.set_batch(c0, np.nan_to_num(pb), vz, fl)
s.step(2)
code = s.newton_iterations().astype(np.int64)
cyc = s.get_status().astype(np.int64)
fwd = (code % 1000) / 1000.0
bwd = (code // 1000) / 1000.0
print('N=%d nx=%d B=%d: cycles per row and iteration (median) %d; forward %.1f %%, backward %.1f %%, update %.1f %%'
      % (N, nx, B, np.median(cyc), 100 * np.median(fwd), 100 * np.median(bwd), 100 * (1 - np.median(fwd) - np.median(bwd))))
