#!/usr/bin/env python3
"""Stationary solves (one Newton solve per operating point from the bulk state, iteration counts spread widely) with the lane kernels
against the lane teams: the lane kernels finish with their slowest point (makespan = max iterations x pace of a wave), the
workgroup-per-point kernels with the sum of the iterations.

    python tools/probe/stationary_choice.py [out.jsonl]
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))


def run(N, nx, B, kernel, phi_max):
    from catint_amd import _capi
    from catint_amd.synthetic import make_batch
    os.environ['CATINT_NEWTON_KERNEL'] = kernel
    prob, c0, pb, vz, fl = make_batch(B, N, nx, seed=1, phi_max=phi_max, dt_factor=0.1)
    radii = [4.1e-10, 3.6e-10, 3.3e-10, 3e-10, 3e-10, 3e-10, 4.5e-10, 3.5e-10][:N]
    with _capi.PnpSolver(prob.N, prob.nx, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Newton', batch_capacity=B) as s:
        s.set_newton(wall_bc='stern', stern_capacitance=0.2, tol=1e-8, maxit=80, mpb_radius=radii)
        s.set_batch(c0, np.nan_to_num(pb), vz, fl)
        s.synchronize()
        s.timer_start()
        st = s.solve_stationary()
        ms = s.timer_stop()
        it = s.newton_iterations()
        # a second solve from a perturbed wall potential (what a continuation stage looks like)
        pb2 = np.nan_to_num(pb).copy()
        pb2[:, 0] *= 1.15
        s.set_pb(pb2, vz)
        s.timer_start()
        st2 = s.solve_stationary()
        ms2 = s.timer_stop()
        it2 = s.newton_iterations()
    return ms, it, int((st == 0).sum()), ms2, it2, int((st2 == 0).sum())


def main():
    out = open(sys.argv[1], 'w') if len(sys.argv) > 1 else None
    for N, nx in ((8, 512), (6, 384)):
        for B in (1536, 2048, 4096, 8192, 16384):
            for phi_max in (0.2, 0.6):
                row = {'N': N, 'nx': nx, 'B': B, 'phi_max': phi_max}
                for kernel in ('lane2', 'lane', 'team', 'sweep'):
                    ms, it, ok, ms2, it2, ok2 = run(N, nx, B, kernel, phi_max)
                    row[kernel] = {'ms_first': ms, 'ms_stage': ms2, 'ok': ok, 'ok_stage': ok2}
                row.update({'its_first_mean': float(it.mean()), 'its_first_max': int(it.max()), 'its_stage_mean': float(it2.mean()), 'its_stage_max': int(it2.max())})
                print(json.dumps(row), flush=True)
                if out:
                    out.write(json.dumps(row) + '\n')
                    out.flush()


if __name__ == '__main__':
    main()
