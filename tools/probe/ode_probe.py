"""Throughput of the device DOPRI5 (pnp_integrate_dopri5) against the round-1 arrangement (scipy drives pnp_mol_rhs from the host)."""
import json
import sys
import time

import numpy as np
import scipy.integrate as si

from catint_amd.host import solver_from_problem
from catint_amd.synthetic import make_batch


def main():
    out = []
    for (N, nx, B) in [(2, 256, 1), (2, 256, 4096), (3, 512, 8192), (6, 1024, 4096)]:
        prob, c0, pb, vz, flux = make_batch(B, N, nx, seed=1, dt_factor=2e-3)
        nt = 4
        with solver_from_problem(prob, 'FTCS', batch_capacity=B) as s:
            s.set_batch(c0, pb, vz, flux)
            s.integrate_dopri5(1, [0], nsteps=10000)          # warm-up (code objects, buffers)
            s.set_batch(c0, pb, vz, flux)
            t0 = time.perf_counter()
            cout, idid, stats, _ = s.integrate_dopri5(nt, [nt - 1], nsteps=10000)
            t1 = time.perf_counter() - t0
            rec = {'N': N, 'nx': nx, 'B': B, 'intervals': nt, 'seconds': t1, 'ok_lanes': int((idid == 1).sum()),
                   'attempted_steps_max': int(stats[:, 0].max()), 'attempted_steps_sum': int(stats[:, 0].sum()),
                   'rhs_evaluations_sum': int(stats[:, 3].sum()),
                   'lane_steps_per_s': float(stats[:, 0].sum() / t1), 'rhs_point_updates_per_s': float(stats[:, 3].sum() * N * nx / t1)}
            if B <= 4096 and N == 2:
                # round-1 arrangement on ONE lane of the same batch: scipy dopri5, every RHS a host round trip of the whole batch state
                state = c0.copy()
                calls = [0]

                def f(t, y):
                    calls[0] += 1
                    state[0] = y
                    return s.mol_rhs(state)[0]
                r = si.ode(f).set_integrator('dopri5', nsteps=10000)
                r.set_initial_value(c0[0].copy())
                t0 = time.perf_counter()
                for _ in range(nt):
                    r.integrate(r.t + prob.dt)
                t2 = time.perf_counter() - t0
                rec['scipy_driven_seconds_one_lane'] = t2
                rec['scipy_driven_rhs_calls'] = calls[0]
                rec['scipy_match'] = float(np.abs(r.y - cout[0, 0]).max() / np.abs(r.y).max())
            out.append(rec)
            print(json.dumps(rec), flush=True)
    return out


if __name__ == '__main__':
    main()
