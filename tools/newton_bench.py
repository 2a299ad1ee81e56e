#!/usr/bin/env python3
"""Timing of the physical mode (fully implicit coupled Newton, pnp_newton.hip) on a synthetic batch
(phiM ~ U(-0.2, 0.2) V, c_bulk log-uniform, L = 40 Debye lengths, dt = 0.1 lambda_D L / D_max).

    python tools/newton_bench.py [--batch 1024 --nspecies 3 --nx 512 --steps 20 --stern]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=1024)
    ap.add_argument('--nspecies', type=int, default=3)
    ap.add_argument('--nx', type=int, default=512)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--phi-max', type=float, default=0.2)
    ap.add_argument('--dt-factor', type=float, default=0.1)
    ap.add_argument('--stern', action='store_true')
    ap.add_argument('--mpb', action='store_true', help='steric ions (MPB radii 3-4.5 Angstrom)')
    ap.add_argument('--reactions', action='store_true', help='one buffer-like homogeneous reaction')
    ap.add_argument('--tol', type=float, default=1e-8)
    a = ap.parse_args()
    from catint_amd import _capi
    from catint_amd.synthetic import make_batch
    prob, c0, pb, vz, fl = make_batch(a.batch, a.nspecies, a.nx, seed=0, phi_max=a.phi_max, dt_factor=a.dt_factor)
    pb = np.nan_to_num(pb)
    s = _capi.PnpSolver(prob.N, prob.nx, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Newton',
                        batch_capacity=a.batch)
    radii = [4.1e-10, 3.6e-10, 3.3e-10, 3e-10, 3e-10, 3e-10, 4.5e-10, 3.5e-10][:a.nspecies] if a.mpb else None
    if a.stern:
        s.set_newton(wall_bc='stern', stern_capacitance=0.2, tol=a.tol, mpb_radius=radii)
    else:
        s.set_newton(tol=a.tol, mpb_radius=radii)
    if a.reactions and a.nspecies >= 3:
        s.set_reactions([([1], [2], 5.0e3, 5.0e3)])
    s.set_batch(c0, pb, vz, fl)
    # one launch per warm-up step: the first two or three launches after an upload run slower from start to end
    w_it = 0
    for _ in range(a.warmup):
        s.step(1)
        w_it += s.newton_iterations().sum()
    s.synchronize()
    s.timer_start()
    s.step(a.steps)
    ms = s.timer_stop()
    it = s.newton_iterations()
    st = s.get_status()
    out = {
        'workload': 'physical mode, batch=%d, %d species, %d points, %s wall' % (a.batch, a.nspecies, a.nx, 'Stern' if a.stern else 'Dirichlet'),
        'timesteps_per_s': a.batch * a.steps / (ms * 1e-3),
        'newton_iterations_per_s': float(it.sum()) / (ms * 1e-3),
        'mean_newton_iterations_per_step': float(it.sum()) / (a.batch * a.steps),
        'warmup_mean_iterations_per_step': float(w_it) / (a.batch * a.warmup),
        'ms_per_step': ms / a.steps, 'lanes_ok': int((st == 0).sum()), 'tol': a.tol,
    }
    print(json.dumps(out))


if __name__ == '__main__':
    main()
