#!/usr/bin/env python3
"""Measured kernel-selection table for the compat timestep: for every shape (species N, points per lane P via nx, batch bucket,
one launch per step or fused launches) time every kernel variant the library can run and write the results as JSON lines
(gpurun_out/config_sweep.jsonl).  tools/make_step_table.py turns the winners into catint_amd/csrc/pnp_step_table.h, which
choose_step_config reads (a measured table instead of a hand-tuned threshold tree).

    python tools/probe/config_sweep.py [out.jsonl]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
from tools.probe.stream_probe import VARIANTS, KEYS      # noqa: E402

VARIANTS = dict(VARIANTS)
VARIANTS.update({
    'W2G2': dict(CATINT_PNP_KERNEL='2', CATINT_PNP_WAVES_PER_GRID='2', CATINT_PNP_SPECIES_PER_WAVE='2'),
    'W4G1': dict(CATINT_PNP_KERNEL='2', CATINT_PNP_WAVES_PER_GRID='4', CATINT_PNP_SPECIES_PER_WAVE='1'),
})
NAMES = ['W1G1', 'W1G2', 'W1G3', 'W2G1', 'W2G2', 'W3G1', 'W4G1', 'rrW1', 'rrW2', 'rrW3', 'st', 'stg2']
SPECIES = [2, 3, 4, 6, 8]
GRIDS = [int(x) for x in os.environ.get('SWEEP_GRIDS', '66,130,258,512,1024').split(',')]            # P = 1, 2, 4, 8, 16
BATCHES = [1024, 4096, 16384]


def measure(prob, c0, pb, vz, fl, B, N, nx, spl):
    from catint_amd import _capi
    s = _capi.PnpSolver(N, nx, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Crank-Nicolson', batch_capacity=B)
    s.set_batch(c0, pb, vz, fl)
    nsteps = spl if spl > 1 else 8
    s.step(nsteps, spl)
    s.step(nsteps, spl)
    s.synchronize()
    ms = []
    for _ in range(5):
        s.timer_start()
        s.step(nsteps, spl)
        ms.append(s.timer_stop() / nsteps)
    ok = int((s.get_status() == 0).sum())
    s.close()
    us = float(np.median(ms)) * 1e3
    return us, 16.0 * (N + 1) * nx * B / (us * 1e-6) / 8e12, ok


def main():
    from catint_amd.synthetic import make_batch
    out = open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'config_sweep.jsonl'), 'w')
    t0 = time.time()
    for N in SPECIES:
        for nx in GRIDS:
            for B in BATCHES:
                prob, c0, pb, vz, fl = make_batch(B, N, nx, seed=0, phi_max=0.025, dt_factor=1e-5)
                for spl in (1, 64):
                    for name in ['default'] + NAMES:
                        if name.startswith(('rr', 'st')) and nx <= 66:
                            continue
                        if name == 'stg2' and nx != 1024:        # the gradient row in LDS pays at 16 points per lane only
                            continue
                        for k in KEYS:
                            os.environ.pop(k, None)
                        os.environ.update(VARIANTS[name])
                        try:
                            us, frac, ok = measure(prob, c0, pb, vz, fl, B, N, nx, spl)
                        except Exception as e:
                            continue
                        out.write(json.dumps({'N': N, 'nx': nx, 'B': B, 'fused': spl > 1, 'variant': name, 'us_per_step': us, 'frac': frac,
                                              'ok': ok == B}) + '\n')
                        out.flush()
                print('N=%d nx=%d B=%d done (%.0f s)' % (N, nx, B, time.time() - t0), flush=True)


if __name__ == '__main__':
    main()
