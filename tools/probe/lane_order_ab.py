#!/usr/bin/env python3
"""Lane kernels with and without the ordering of the operating points by expected Newton iterations (CATINT_LANE_ORDER), one process,
alternating, several repetitions per setting: timesteps/s, and the share of a wave's lane-iterations that are spent waiting for the
wave's slowest point (from the per-point iteration counts, in slot order of the identity layout).

    python tools/probe/lane_order_ab.py [out.jsonl]
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))


def run(N, nx, B, steps, order, kernel):
    from catint_amd import _capi
    from catint_amd.synthetic import make_batch
    os.environ['CATINT_LANE_ORDER'] = str(order)
    os.environ['CATINT_NEWTON_KERNEL'] = kernel
    prob, c0, pb, vz, fl = make_batch(B, N, nx, seed=0, phi_max=0.2, dt_factor=0.1)
    radii = [4.1e-10, 3.6e-10, 3.3e-10, 3e-10, 3e-10, 3e-10, 4.5e-10, 3.5e-10][:N]
    with _capi.PnpSolver(prob.N, prob.nx, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Newton',
                         batch_capacity=B) as s:
        s.set_newton(wall_bc='stern', stern_capacitance=0.2, tol=1e-8, mpb_radius=radii)
        s.set_batch(c0, np.nan_to_num(pb), vz, fl)
        s.step(2)
        s.synchronize()
        s.timer_start()
        s.step(steps)
        ms = s.timer_stop()
        it = s.newton_iterations()
    lanes = 32 if kernel == 'lane' else 16
    pad = (-B) % lanes
    g = np.concatenate([it, np.zeros(pad, it.dtype)]).reshape(-1, lanes)
    waste_identity = float(g.max(axis=1).sum() * lanes / max(it.sum(), 1))
    gs = np.sort(np.concatenate([it, np.zeros(pad, it.dtype)]))[::-1].reshape(-1, lanes)
    waste_sorted = float(gs.max(axis=1).sum() * lanes / max(it.sum(), 1))
    return B * steps / (ms * 1e-3), float(it.mean()) / steps, waste_identity, waste_sorted


def main():
    out = open(sys.argv[1], 'w') if len(sys.argv) > 1 else None
    for N, nx, B, steps, kernel in ((8, 512, 32768, 10, 'lane'), (6, 1024, 32768, 6, 'lane'), (8, 512, 8192, 10, 'lane2'), (8, 512, 65536, 6, 'lane')):
        res = {0: [], 1: []}
        for rep in range(4):
            for order in (0, 1):
                r, its, w_id, w_sorted = run(N, nx, B, steps, order, kernel)
                res[order].append(r)
        rec = {'N': N, 'nx': nx, 'B': B, 'steps': steps, 'kernel': kernel, 'identity': res[0], 'ordered': res[1],
               'median_identity': float(np.median(res[0])), 'median_ordered': float(np.median(res[1])),
               'iterations_per_step': its, 'wave_iterations_over_point_iterations_identity': w_id,
               'wave_iterations_over_point_iterations_if_sorted_by_this_launch': w_sorted}
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + '\n')
            out.flush()


if __name__ == '__main__':
    main()
