#!/usr/bin/env python3
"""Median time per fused timestep of every instantiated (waves per lane W, species per wave G) variant of step_kernel on one
shape; the variant is forced through CATINT_PNP_WAVES_PER_GRID / CATINT_PNP_SPECIES_PER_WAVE in child processes.

    python tools/probe/step_config_probe.py N nx B [N nx B ...]
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)


def child(N, nx, B, spl):
    from catint_amd import _capi
    from catint_amd.synthetic import make_batch
    prob, c0, pb, vz, fl = make_batch(B, N, nx, seed=0, dt_factor=1e-5)
    s = _capi.PnpSolver(N, nx, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Crank-Nicolson', batch_capacity=B)
    s.set_batch(c0, pb, vz, fl)
    reps = 1 if spl >= 8 else 64        # one-step launches are timed in groups
    for _ in range(6):
        s.step(spl * reps, spl)
    ms = []
    for _ in range(30):
        s.timer_start()
        s.step(spl * reps, spl)
        ms.append(s.timer_stop() / reps)
    ok = int((s.get_status() == 0).sum())
    us = float(np.median(ms)) / spl * 1e3
    print(json.dumps({'us_per_step': us, 'frac': 16.0 * (N + 1) * nx * B / (us * 1e-6) / 8e12, 'ok': ok}))


if __name__ == '__main__':
    if sys.argv[1] == '--child':
        child(*[int(x) for x in sys.argv[2:6]])
        sys.exit(0)
    shapes = [int(x) for x in sys.argv[1:]]
    for i in range(0, len(shapes), 3):
        N, nx, B = shapes[i:i + 3]
        row = []
        for W, G in [(0, 0), (1, 1), (1, 2), (1, 3), (2, 1), (2, 2), (3, 1), (4, 1), (-1, 1), (-3, 1)]:
            env = dict(os.environ)
            if W > 0:
                env.update(CATINT_PNP_KERNEL='2', CATINT_PNP_WAVES_PER_GRID=str(W), CATINT_PNP_SPECIES_PER_WAVE=str(G))
            elif W < 0:      # register-resident kernel with -W waves per lane
                env.update(CATINT_PNP_KERNEL='4', CATINT_PNP_WAVES_PER_GRID=str(-W))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', str(N), str(nx), str(B), os.environ.get('PROBE_SPL', '256')], env=env,
                               capture_output=True, text=True, timeout=300)
            try:
                d = json.loads(r.stdout.strip().splitlines()[-1])
                row.append('%s %.3f' % ('default' if not W else ('rrW%d' % -W if W < 0 else 'W%dG%d' % (W, G)), d['frac']))
            except Exception:
                row.append('W%dG%d failed' % (W, G))
        print('N=%d nx=%d B=%d: %s' % (N, nx, B, ' | '.join(row)), flush=True)
