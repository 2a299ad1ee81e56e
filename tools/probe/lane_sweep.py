#!/usr/bin/env python3
"""Lane kernel against the default kernel choice over species count x grid x batch (physical mode, transient steps, steric ions +
Stern wall as in bench.py's physical_mode): one JSON line per shape -> profiles/r03_lane_sweep.jsonl feeds newton_lane_preferred.

    python tools/probe/lane_sweep.py [--out gpurun_out/lane_sweep.jsonl] [--point-ions]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))


def run(N, nx, B, kern, steps, mpb):
    from catint_amd import _capi
    from catint_amd.synthetic import make_batch
    os.environ['CATINT_NEWTON_KERNEL'] = kern
    prob, c0, pb, vz, fl = make_batch(B, N, nx, seed=0, phi_max=0.2, dt_factor=0.1)
    radii = [4.1e-10, 3.6e-10, 3.3e-10, 3e-10, 3e-10, 3e-10, 4.5e-10, 3.5e-10][:N] if mpb else None
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
        ok = int((s.get_status() == 0).sum())
    return B * steps / (ms * 1e-3), float(it.sum()) / (B * steps), ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='gpurun_out/lane_sweep.jsonl')
    ap.add_argument('--point-ions', action='store_true')
    ap.add_argument('--budget-gb', type=float, default=40.0)
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    t0 = time.time()
    with open(a.out, 'a') as f:
        for N in (2, 3, 4, 5, 6, 7, 8):
            for nx in (128, 512, 1024, 4096):
                for B in (512, 2048, 8192, 32768, 131072):
                    rec_gb = B * nx * ((N + 1) * (N + 2) + 3 * (N + 2)) * 8 / 1e9
                    if rec_gb > a.budget_gb or B * nx * (N + 1) > 4.5e9:
                        continue
                    steps = 4 if B * nx >= 8e6 else 8
                    row = {'N': N, 'nx': nx, 'B': B, 'mpb': not a.point_ions}
                    for kern in ('lane', ''):
                        try:
                            r, its, ok = run(N, nx, B, kern, steps, not a.point_ions)
                        except Exception as e:      # noqa: BLE001
                            row['error_' + (kern or 'default')] = str(e)[:200]
                            continue
                        row[(kern or 'default')] = r
                        row['its'] = its
                        row['ok_' + (kern or 'default')] = ok
                    f.write(json.dumps(row) + '\n')
                    f.flush()
                    print(json.dumps(row), 'elapsed %.0f s' % (time.time() - t0), flush=True)


if __name__ == '__main__':
    main()
