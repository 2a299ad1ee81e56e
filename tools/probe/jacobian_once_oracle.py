#!/usr/bin/env python3
"""CPU only: how many iterations per timestep does the chord iteration need (Jacobian of the first iterate kept for the whole step --
what the reference asks COMSOL for, comsol_model.py:526,530 jtech "once") on bench.py's large-batch workload, against the Newton
iteration the library runs?  Oracle arithmetic (oracle/pnp_physical.py), a sample of the lanes, the bench's tolerance 1e-8.
The cost model of DESIGN.md section 7b turns the counts into a verdict.
usage: python tools/probe/jacobian_once_oracle.py [N nx lanes steps]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
from catint_amd.synthetic import make_batch      # noqa: E402
from oracle import pnp_physical as PH            # noqa: E402
import bench                                      # noqa: E402


def main():
    N, nx, lanes, steps = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (8, 512, 12, 6)
    prob, c0, pb, vz, fl = make_batch(4096, N, nx, seed=4446, phi_max=0.2, dt_factor=0.1)      # bench.newton_solver's inputs
    pick = np.linspace(0, 4095, lanes).astype(int)
    rows = []
    for b in pick:
        cb = c0[b].reshape(N, nx)[:, -1]
        p = PH.PhysicalProblem(D=prob.D, charges=prob.charges, beta=prob.beta, eps=prob.eps, dx=prob.dx, nx=nx, c_bulk=cb, phiM=pb[b, 0],
                               flux=fl[b], stern_capacitance=0.2, mpb_radius=bench.RADII8[:N])
        row = {'lane': int(b), 'phiM': float(pb[b, 0])}
        ref = None
        for name, kw in (('newton', {}), ('chord', dict(jacobian_once=True)), ('newton_error_estimate', dict(estimate=True))):
            c, phi, its = PH.integrate(p, c0[b].reshape(N, nx).copy(), np.zeros(nx), prob.dt, steps, tol=1e-8, maxit=50, **kw)
            row[name] = [int(i) for i in its]
            if ref is None:
                ref = c
            else:
                row[name + '_state_vs_newton'] = float(np.abs(c - ref).max() / np.abs(ref).max())
        rows.append(row)
        print(json.dumps(row), flush=True)
    mean = {k: float(np.mean([np.mean(r[k]) for r in rows])) for k in ('newton', 'chord', 'newton_error_estimate')}
    # a factorising iteration costs 1 (the chord's first one 1.2: it also stores D'^-1 and the behind block), a reuse
    # iteration 0.5 in instructions -- and 1.2 in BYTES (reads D'^-1, the behind block and T: ~260 doubles per row against 215)
    k = mean['chord']
    print(json.dumps({'mean_iterations_per_step': mean, 'cost_model': {
        'newton': mean['newton'], 'chord_where_instructions_bind': 1.2 + 0.5 * (k - 1.0), 'chord_where_hbm_binds': 1.2 + 1.2 * (k - 1.0)}}))


if __name__ == '__main__':
    main()
