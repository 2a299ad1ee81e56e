#!/usr/bin/env python3
"""The stiff method-of-lines integrator on the device (pnp_integrate_rkc) over a batch, against scipy's odeint driving the device
right-hand side one operating point at a time (what calc='odeint' does for a single point, as the reference): operating-point
intervals per second, right-hand sides per second, stage counts.

    python tools/probe/rkc_probe.py [out.jsonl]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))


def main():
    import scipy.integrate as si
    from catint_amd.synthetic import make_batch
    from catint_amd.host import solver_from_problem
    out = open(sys.argv[1], 'w') if len(sys.argv) > 1 else None
    for B, N, nx, factor, nt in ((1024, 3, 512, 200.0, 4), (4096, 3, 512, 200.0, 4), (4096, 3, 512, 2000.0, 2), (8192, 6, 1024, 200.0, 2)):
        prob, c0, pb, vz, fl = make_batch(B, N, nx, seed=3, phi_max=0.025, dt_factor=1.0)
        h_expl = prob.dx ** 2 / (2.0 * max(prob.D))          # explicit stability limit of the diffusion part
        prob.dt = factor * h_expl
        with solver_from_problem(prob, 'FTCS', batch_capacity=B) as s:
            s.set_batch(c0, pb, vz, fl)
            s.integrate_rkc(1, [0])                         # warm-up (allocations)
            s.set_batch(c0, pb, vz, fl)
            t0 = time.perf_counter()
            cout, idid, stats, t_end = s.integrate_rkc(nt, [nt - 1])
            wall = time.perf_counter() - t0
            rec = {'B': B, 'N': N, 'nx': nx, 'interval_over_explicit_limit': factor, 'intervals': nt, 'wall_s': wall,
                   'lanes_ok': int((idid == 1).sum()), 'lane_intervals_per_s': B * nt / wall,
                   'rhs_per_lane': float((stats[:, 3] + stats[:, 5]).mean()), 'rhs_lane_evaluations_per_s': float((stats[:, 3] + stats[:, 5]).sum()) / wall,
                   'steps_per_lane': float(stats[:, 0].mean()), 'rejected_per_lane': float(stats[:, 2].mean()), 'max_stages': int(stats[:, 6].max()),
                   'ticks_upper_bound': int((stats[:, 3] + stats[:, 5]).max())}
            # scipy's odeint on ONE operating point, device right-hand side (a host <-> device round trip per evaluation)
            nref = 2
            s1 = solver_from_problem(prob, 'FTCS', batch_capacity=1)
            t0 = time.perf_counter()
            errs = []
            for b in range(nref):
                s1.set_batch(c0[b:b + 1], pb[b:b + 1], vz[b:b + 1], fl[b:b + 1])
                ref = si.odeint(lambda y, t: s1.mol_rhs(y[None, :])[0], c0[b], np.arange(nt + 1) * prob.dt, rtol=1e-8, atol=1e-12,
                                ml=N, mu=N, mxstep=100000)
                errs.append(float(np.abs(cout[0, b] - ref[-1]).max() / np.abs(ref[-1]).max()))
            t_ode = (time.perf_counter() - t0) / nref
            s1.close()
            rec.update({'odeint_s_per_lane_device_rhs': t_ode, 'speedup_vs_odeint_per_lane': t_ode * B / wall, 'relerr_vs_odeint': max(errs)})
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + '\n')
            out.flush()


if __name__ == '__main__':
    main()
