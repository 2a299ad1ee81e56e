#!/usr/bin/env python3
"""512-register builds of the row-per-thread Newton kernel (tools/probe/libtrap_*.so): is the solve bitwise
reproducible run to run, and does it agree with the oracle?  One process per library (CATINT_PNP_LIB)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
os.environ['CATINT_NEWTON_KERNEL'] = 'generic'
from catint_amd import _capi
import tests.test_gpu_newton as T

def solve(N, nx, B, seed):
    D, q, cb, dx, phiM = T.make_lanes(N, nx, B, seed)
    c0 = np.repeat(cb[:, :, None], nx, axis=2)
    pb = np.zeros((B, 4)); pb[:, 0] = phiM
    with _capi.PnpSolver(N, nx, dx, 1.0, T.BETA, T.EPS, D, q, method='Newton', batch_capacity=B) as s:
        s.set_newton()
        s.set_batch(c0, pb, np.zeros(B), np.zeros((B, N)))
        st = s.solve_stationary()
        c, phi, _, _ = s.get_state()
        return c, phi, s.newton_iterations(), st

print('library', os.environ.get('CATINT_PNP_LIB', 'default'))
for (N, nx, B) in [(5, 70, 5), (6, 96, 5), (6, 64, 64), (5, 200, 300), (6, 300, 300), (3, 128, 5)]:
    r = [solve(N, nx, B, N * 1000 + nx) for _ in range(4)]
    same = all(np.array_equal(r[0][0], x[0]) and np.array_equal(r[0][1], x[1]) for x in r[1:])
    dmax = max(np.abs(r[0][0] - x[0]).max() / np.abs(r[0][0]).max() for x in r[1:])
    fin = all(np.isfinite(x[0]).all() for x in r)
    print('N=%d nx=%d B=%d reproducible=%s maxdiff=%.2e finite=%s its=%s status=%s' % (N, nx, B, same, dmax, fin, r[0][2][:5], r[0][3][:5]), flush=True)
