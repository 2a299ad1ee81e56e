#!/usr/bin/env python3
"""CPU only (oracle arithmetic): would block-Thomas records kept in SINGLE precision for the back-substitution (the elimination itself in
double) change the Newton iteration?  The lane kernels stream 90 of their 215 doubles per grid row as the record [T | t]; T in fp32
would take 19 % off the bytes of an HBM-bound launch.  Here: the bench's transient workload, Newton with the exact block solve against
Newton whose back-substitution uses float32(T) (t stays double): iteration counts per step and the difference of the states."""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
from catint_amd.synthetic import make_batch      # noqa: E402
from oracle import pnp_physical as PH            # noqa: E402
import bench                                      # noqa: E402


def thomas(L, M, U, rhs, t32):
    """block Thomas from the wall to the bulk; rhs, result [(N+1), nx]; t32: the back-substitution sees float32(T)"""
    nx, nb, _ = M.shape
    T = np.zeros((nx, nb, nb)); t = np.zeros((nx, nb))
    for i in range(nx):
        D = M[i] - (L[i] @ T[i - 1] if i > 0 else 0.0)
        r = rhs[:, i] - (L[i] @ t[i - 1] if i > 0 else 0.0)
        T[i] = np.linalg.solve(D, U[i]) if i < nx - 1 else 0.0
        t[i] = np.linalg.solve(D, r)
    Tb = T.astype(np.float32).astype(np.float64) if t32 else T
    x = np.zeros((nx, nb))
    x[nx - 1] = t[nx - 1]
    for i in range(nx - 2, -1, -1):
        x[i] = t[i] - Tb[i] @ x[i + 1]
    return x.T


def main():
    N, nx, lanes, steps = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (8, 512, 6, 8)
    prob, c0, pb, vz, fl = make_batch(4096, N, nx, seed=4446, phi_max=0.2, dt_factor=0.1)
    tot = {'exact': 0, 'f32_T': 0}
    worst = 0.0
    for b in np.linspace(0, 4095, lanes).astype(int):
        cb = c0[b].reshape(N, nx)[:, -1]
        p = PH.PhysicalProblem(D=prob.D, charges=prob.charges, beta=prob.beta, eps=prob.eps, dx=prob.dx, nx=nx, c_bulk=cb, phiM=pb[b, 0],
                               flux=fl[b], stern_capacitance=0.2, mpb_radius=bench.RADII8[:N])
        out = {}
        for name, t32 in (('exact', False), ('f32_T', True)):
            c, phi, its = PH.integrate(p, c0[b].reshape(N, nx).copy(), np.zeros(nx), prob.dt, steps, tol=1e-8, maxit=50,
                                       solver=lambda L, M, U, r, t32=t32: thomas(L, M, U, r, t32))
            out[name] = (c, its)
            tot[name] += sum(its)
        d = float(np.abs(out['f32_T'][0] - out['exact'][0]).max() / np.abs(out['exact'][0]).max())
        worst = max(worst, d)
        print(json.dumps({'lane': int(b), 'phiM': float(pb[b, 0]), 'exact': [int(i) for i in out['exact'][1]], 'f32_T': [int(i) for i in out['f32_T'][1]],
                          'state_rel_diff': d}), flush=True)
    print(json.dumps({'iterations': tot, 'worst_state_rel_diff': worst}))


if __name__ == '__main__':
    main()
