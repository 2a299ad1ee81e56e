"""dev probe: MODE 2 (reactions) kernel with a zero-rate table must equal the MODE 1 kernel"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from catint_amd import _capi
import tests.test_gpu_newton as T

def solve(N, nx, rx, mpb, kern):
    os.environ['CATINT_NEWTON_KERNEL'] = kern
    D, q, cb, dx, phiM = T.make_lanes(N, nx, 3, 5)
    c0 = np.repeat(cb[:, :, None], nx, axis=2)
    pb = np.zeros((3, 4)); pb[:, 0] = phiM
    with _capi.PnpSolver(N, nx, dx, 1.0, T.BETA, T.EPS, D, q, method='Newton', batch_capacity=3) as s:
        s.set_newton(mpb_radius=([4.1e-10] + [0.0] * (N - 1)) if mpb else None)
        if rx:
            s.set_reactions([([0, 1], [0], 0.0, 0.0)])
        s.set_batch(c0, pb, np.zeros(3), np.zeros((3, N)))
        st = s.solve_stationary()
        c, phi, _, _ = s.get_state()
        return c, phi, s.newton_iterations()

NS = [int(x) for x in os.environ.get("PROBE_N", "2 3 4 5 6 7 8").split()]
for kern in ('generic', ''):
    for N in NS:
        for nx in (40, 200):
            for mpb in (False, True):
                a = solve(N, nx, False, mpb, kern); b = solve(N, nx, True, mpb, kern)
                d = np.abs(a[0] - b[0]).max() / np.abs(a[0]).max()
                print('kernel=%-8s N=%d nx=%3d mpb=%d  rel diff %.2e its %s %s %s' % (kern or 'auto', N, nx, mpb, d, a[2], b[2], 'BAD' if d > 1e-9 else ''))
