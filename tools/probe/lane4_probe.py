#!/usr/bin/env python3
"""Lane-quad kernel (pnp_lane4.hip) against the lane-pair, lane and lane-team kernels: one process, one device, kernels alternated
(NEWTON_KERNEL is a per-handle option: the environment is read at handle creation).  One JSON line per shape.

    python tools/probe/lane4_probe.py [--out gpurun_out/lane4_probe.jsonl] ["N NX B" ...]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
RADII8 = [4.1e-10, 3.6e-10, 3.3e-10, 3e-10, 3e-10, 3e-10, 4.5e-10, 3.5e-10]


def run(N, nx, B, kern, steps, reps=2):
    from catint_amd import _capi
    from catint_amd.synthetic import make_batch
    os.environ['CATINT_NEWTON_KERNEL'] = kern
    prob, c0, pb, vz, fl = make_batch(B, N, nx, seed=4444, phi_max=0.2, dt_factor=0.1)
    best = 0.0
    with _capi.PnpSolver(prob.N, prob.nx, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Newton',
                         batch_capacity=B) as s:
        s.set_newton(wall_bc='stern', stern_capacitance=0.2, tol=1e-8, mpb_radius=RADII8[:N])
        for _ in range(reps):
            s.set_batch(c0, np.nan_to_num(pb), vz, fl)
            s.step(1)
            s.synchronize()
            s.timer_start()
            s.step(steps)
            ms = s.timer_stop()
            it = s.newton_iterations()
            ok = int((s.get_status() == 0).sum())
            best = max(best, B * steps / (ms * 1e-3))
    return best, float(it.sum()) / (B * steps), ok


def main():
    args = sys.argv[1:]
    out = 'gpurun_out/lane4_probe.jsonl'
    if args and args[0] == '--out':
        out, args = args[1], args[2:]
    shapes = [tuple(int(v) for v in a.split()) for a in args] or [(8, 512, 1024), (8, 512, 2048), (8, 512, 4096), (8, 512, 8192),
                                                                   (8, 512, 16384), (6, 1024, 4096), (6, 1024, 8192), (8, 4096, 2048),
                                                                   (8, 4096, 8192), (7, 384, 4096), (5, 512, 8192)]
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, 'a') as f:
        for N, nx, B in shapes:
            steps = 4 if B * nx >= 8e6 else 8
            row = {'N': N, 'nx': nx, 'B': B, 'steps': steps}
            for kern in ('lane4', 'lane2', 'lane', 'team'):
                if kern == 'team' and B * nx > 4e6:
                    continue
                try:
                    t0 = time.time()
                    r, its, ok = run(N, nx, B, kern, steps)
                    row[kern] = r
                    row['its_' + kern] = its
                    row['ok_' + kern] = ok
                except Exception as e:      # noqa: BLE001
                    row['error_' + kern] = str(e)[:200]
            print(json.dumps(row), flush=True)
            f.write(json.dumps(row) + '\n')
            f.flush()


if __name__ == '__main__':
    main()
