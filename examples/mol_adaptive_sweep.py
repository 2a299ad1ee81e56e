#!/usr/bin/env python3
"""Method-of-lines polarization sweep with the adaptive integrator on the device.

The reference runs its `calc='dopri5'` / `'dop853'` path one operating point at a time, scipy calling the Python right-hand side
(catint/calculator_old.py:821-973).  Here every descriptor point is a GPU lane with its own adaptive Runge-Kutta integration
(Hairer's DOPRI5 / DOP853 as scipy wraps them, pnp_integrate_dopri5 / pnp_integrate_dop853): own step size, own accept / reject
history, no host round trip per right-hand-side evaluation.

    python examples/mol_adaptive_sweep.py --lanes 64 --calc dop853
    python examples/mol_adaptive_sweep.py --lanes 4096 --calc odeint --tmax 1e-6      # stiff: steps far beyond dx^2 / (2 D)
"""
import argparse
import collections
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from catint_amd.calculator import Calculator      # noqa: E402
from catint_amd.transport import Transport        # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--lanes', type=int, default=32)
    ap.add_argument('--nx', type=int, default=96)
    ap.add_argument('--calc', default='dopri5', choices=['dopri5', 'dop853', 'odeint', 'lsoda'],
                    help="'odeint' / 'lsoda': the stiff integrator on the device (Runge-Kutta-Chebyshev, pnp_integrate_rkc)")
    ap.add_argument('--dt', type=float, default=5e-10)
    ap.add_argument('--tmax', type=float, default=5e-9)
    a = ap.parse_args(argv)
    phis = list(np.linspace(-0.05, 0.05, a.lanes))
    species = collections.OrderedDict([('K+', {'bulk_concentration': 30.0}), ('Cl-', {'bulk_concentration': 10.0}),
                                       ('HCO3-', {'bulk_concentration': 20.0})])
    tp = Transport(species=species, system={'phiM': 0.0, 'boundary thickness': 2e-8}, nx=a.nx,
                   pb_bound={'potential': {'wall': 'phiM', 'bulk': 0.0}}, descriptors={'phiM': phis})
    calc = Calculator(transport=tp, calc=a.calc, dt=a.dt, tmax=a.tmax, ntout=2)
    t0 = time.time()
    calc.run()
    dt = time.time() - t0
    st = calc.ode_stats
    print('%d lanes x %d species x %d points, %s, %d intervals of %.1e s: %.3f s; attempted steps per lane %d..%d (rejected %d..%d), '
          'right-hand-side evaluations %d in total, %d lanes ok'
          % (a.lanes, tp.nspecies, tp.nx, a.calc, tp.nt, tp.dt, dt, st[:, 0].min(), st[:, 0].max(), st[:, 2].min(), st[:, 2].max(),
             st[:, 3].sum(), int((calc.ode_idid == 1).sum())))
    print('phiM [V]   c_K+(0)    c_Cl-(0)   phi(1) [V]')
    for i in np.linspace(0, a.lanes - 1, min(a.lanes, 7)).astype(int):
        d = tp.alldata[i]
        print('%7.3f   %8.4f   %8.4f   %9.5f' % (phis[i], d['species']['K+']['surface_concentration'], d['species']['Cl-']['surface_concentration'],
                                               d['system']['potential'][1]))
    return {'seconds': dt, 'stats': st, 'idid': calc.ode_idid, 'surface_K': np.array([tp.alldata[i]['species']['K+']['surface_concentration'] for i in range(a.lanes)])}


if __name__ == '__main__':
    main()
