"""Debug: state after k Newton iterations, lane kernel against the default kernels (first deviating row / variable)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from catint_amd import _capi
from tests.test_gpu_newton import BETA, EPS, make_lanes

N, nx, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kw = {}
if len(sys.argv) > 4 and sys.argv[4] == 'mpb':
    kw = {'stern_capacitance': 0.25, 'wall_bc': 'stern', 'mpb_radius': [3.5e-10] * N}
D, q, cb, dx, phiM = make_lanes(N, nx, B, N * 31 + nx)
c0 = np.repeat(cb[:, :, None], nx, axis=2)
pb = np.zeros((B, 4)); pb[:, 0] = phiM
for maxit in (1, 2, 3):
    out = {}
    for kern in ('lane', 'team' if N >= 2 else 'generic'):
        os.environ['CATINT_NEWTON_KERNEL'] = kern
        with _capi.PnpSolver(N, nx, dx, 1.0, BETA, EPS, D, q, method='Newton', batch_capacity=B) as s:
            s.set_newton(maxit=maxit, **kw)
            s.set_batch(c0, pb, np.zeros(B), np.zeros((B, N)))
            s.solve_stationary()
            c, phi, _, _ = s.get_state()
            out[kern] = (c.copy(), phi.copy())
    (c1, p1), (c2, p2) = out.values()
    dc = np.abs(c1 - c2) / np.abs(c2).max()
    dp = np.abs(p1 - p2)
    print('maxit', maxit, 'max dc', dc.max(), 'at', np.unravel_index(dc.argmax(), dc.shape), 'max dphi', dp.max(), 'at', np.unravel_index(dp.argmax(), dp.shape))
    if maxit == 1:
        b = 0
        np.set_printoptions(linewidth=200, precision=4)
        print('phi lane', p1[b]); print('phi ref ', p2[b])
        print('c0 lane', c1[b, 0]); print('c0 ref ', c2[b, 0])
