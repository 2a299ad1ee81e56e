"""dev probe: is the physical-mode solve bitwise reproducible run to run?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from catint_amd import _capi
import tests.test_gpu_newton as T

def solve(N, nx, B, seed):
    D, q, cb, dx, phiM = T.make_lanes(N, nx, B, seed)
    c0 = np.repeat(cb[:, :, None], nx, axis=2)
    pb = np.zeros((B, 4)); pb[:, 0] = phiM
    with _capi.PnpSolver(N, nx, dx, 1.0, T.BETA, T.EPS, D, q, method='Newton', batch_capacity=B) as s:
        s.set_newton()
        s.set_batch(c0, pb, np.zeros(B), np.zeros((B, N)))
        s.solve_stationary()
        c, phi, _, _ = s.get_state()
        return c, phi, s.newton_iterations()

for kern in ('', 'generic'):
    for ex in ('', 'global'):
        os.environ['CATINT_NEWTON_KERNEL'] = kern
        os.environ['CATINT_NEWTON_EXCHANGE'] = ex
        for (N, nx) in [(3, 128), (3, 512), (6, 96), (5, 70), (7, 50), (2, 64)]:
            r = [solve(N, nx, 5, N * 1000 + nx) for _ in range(4)]
            same = all(np.array_equal(r[0][0], x[0]) and np.array_equal(r[0][1], x[1]) for x in r[1:])
            dmax = max(np.abs(r[0][0] - x[0]).max() / np.abs(r[0][0]).max() for x in r[1:])
            print('kernel=%-8s exchange=%-7s N=%d nx=%d reproducible=%s maxdiff=%.2e its=%s' % (kern or 'auto', ex or 'auto', N, nx, same, dmax, r[0][2]))
