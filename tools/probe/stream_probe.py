#!/usr/bin/env python3
"""Streaming regime of the compat timestep: one launch per timestep (state read from and written to HBM by every launch) and
fused launches, on working sets beyond the 256 MiB Infinity Cache, for every kernel variant that can run the shape.

    python tools/probe/stream_probe.py N nx B [N nx B ...]          (PROBE_VARIANTS=default,W1G1,... to restrict)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)

VARIANTS = {
    'default': {},
    'W1G1': dict(CATINT_PNP_KERNEL='2', CATINT_PNP_WAVES_PER_GRID='1', CATINT_PNP_SPECIES_PER_WAVE='1'),
    'W1G2': dict(CATINT_PNP_KERNEL='2', CATINT_PNP_WAVES_PER_GRID='1', CATINT_PNP_SPECIES_PER_WAVE='2'),
    'W1G3': dict(CATINT_PNP_KERNEL='2', CATINT_PNP_WAVES_PER_GRID='1', CATINT_PNP_SPECIES_PER_WAVE='3'),
    'W2G1': dict(CATINT_PNP_KERNEL='2', CATINT_PNP_WAVES_PER_GRID='2', CATINT_PNP_SPECIES_PER_WAVE='1'),
    'W3G1': dict(CATINT_PNP_KERNEL='2', CATINT_PNP_WAVES_PER_GRID='3', CATINT_PNP_SPECIES_PER_WAVE='1'),
    'rrW1': dict(CATINT_PNP_KERNEL='4', CATINT_PNP_WAVES_PER_GRID='1'),
    'rrW2': dict(CATINT_PNP_KERNEL='4', CATINT_PNP_WAVES_PER_GRID='2'),
    'rrW3': dict(CATINT_PNP_KERNEL='4', CATINT_PNP_WAVES_PER_GRID='3'),
    'st': dict(CATINT_PNP_KERNEL='5'),
    'stg': dict(CATINT_PNP_KERNEL='6'),
    'stg2': dict(CATINT_PNP_KERNEL='7'),
}
KEYS = ['CATINT_PNP_KERNEL', 'CATINT_PNP_WAVES_PER_GRID', 'CATINT_PNP_SPECIES_PER_WAVE']      # (CATINT_PNP_ST_WAVES_PER_CU passes through)


def measure(prob, c0, pb, vz, fl, B, N, nx, spl, nsteps, reps):
    from catint_amd import _capi
    s = _capi.PnpSolver(N, nx, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Crank-Nicolson', batch_capacity=B)
    s.set_batch(c0, pb, vz, fl)
    s.step(nsteps, spl)
    s.synchronize()
    ms = []
    for _ in range(reps):
        s.timer_start()
        s.step(nsteps, spl)
        ms.append(s.timer_stop() / nsteps)
    ok = int((s.get_status() == 0).sum())
    s.close()
    us = float(np.median(ms)) * 1e3
    return us, 16.0 * (N + 1) * nx * B / (us * 1e-6) / 8e12, ok


def main():
    from catint_amd.synthetic import make_batch
    shapes = [int(x) for x in sys.argv[1:]]
    names = os.environ.get('PROBE_VARIANTS', 'default,W1G1,W1G2,W1G3,W3G1,rrW1').split(',')
    spls = [int(x) for x in os.environ.get('PROBE_SPL', '1,32').split(',')]
    for i in range(0, len(shapes), 3):
        N, nx, B = shapes[i:i + 3]
        prob, c0, pb, vz, fl = make_batch(B, N, nx, seed=0, phi_max=0.025, dt_factor=1e-5)
        state_mb = 8.0 * (N + 2) * ((nx + 15) // 16 * 16) * B / 1e6
        for spl in spls:
            row = []
            for name in names:
                for k in KEYS:
                    os.environ.pop(k, None)
                os.environ.update(VARIANTS[name])
                try:
                    nsteps = spl if spl > 1 else 8
                    us, frac, ok = measure(prob, c0, pb, vz, fl, B, N, nx, spl, nsteps, 5)
                    row.append('%s %.3f%s' % (name, frac, '' if ok == B else ' (ok %d)' % ok))
                except Exception as e:   # variant not instantiated for this shape
                    row.append('%s n/a' % name)
            print('N=%d nx=%d B=%d state=%.0fMB spl=%d: %s' % (N, nx, B, state_mb, spl, ' | '.join(row)), flush=True)


if __name__ == '__main__':
    main()
