#!/usr/bin/env python3
"""Do the kernel families walk the same bits?  One launch per step (table choice), fused launches (table choice), and every forced
family on a few shapes: max relative difference against the register-resident kernel."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from catint_amd.synthetic import make_batch
from catint_amd.host import solver_from_problem

def run(p, method, c0, pb, vz, fl, nsteps, spl):
    with solver_from_problem(p, method, batch_capacity=c0.shape[0]) as s:
        s.set_batch(c0, pb, vz, fl)
        s.step(nsteps, spl)
        return s.get_state()

KEYS = ['CATINT_PNP_KERNEL', 'CATINT_PNP_WAVES_PER_GRID', 'CATINT_PNP_SPECIES_PER_WAVE']
for (N, nx, B) in [(3, 512, 1024), (2, 130, 300), (6, 1024, 64), (4, 258, 500), (3, 512, 4096)]:
    for method in ('Crank-Nicolson', 'FTCS'):
        p, c0, pb, vz, fl = make_batch(B, N, nx, seed=nx, phi_max=0.02, dt_factor=1e-4 if method == 'Crank-Nicolson' else 2e-5)
        for k in KEYS: os.environ.pop(k, None)
        os.environ.update(CATINT_PNP_KERNEL='4', CATINT_PNP_WAVES_PER_GRID='1')
        ref = run(p, method, c0, pb, vz, fl, 16, 1)
        out = []
        for name, env, spl in [('table/step', {}, 1), ('table/fused', {}, 16), ('W1G1', dict(CATINT_PNP_KERNEL='2', CATINT_PNP_WAVES_PER_GRID='1', CATINT_PNP_SPECIES_PER_WAVE='1'), 1),
                               ('W3G1', dict(CATINT_PNP_KERNEL='2', CATINT_PNP_WAVES_PER_GRID='3', CATINT_PNP_SPECIES_PER_WAVE='1'), 16),
                               ('W1G2', dict(CATINT_PNP_KERNEL='2', CATINT_PNP_WAVES_PER_GRID='1', CATINT_PNP_SPECIES_PER_WAVE='2'), 16),
                               ('rrW3', dict(CATINT_PNP_KERNEL='4', CATINT_PNP_WAVES_PER_GRID='3'), 16), ('st', dict(CATINT_PNP_KERNEL='5'), 16)]:
            for k in KEYS: os.environ.pop(k, None)
            os.environ.update(env)
            got = run(p, method, c0, pb, vz, fl, 16, spl)
            d = max(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300) for a, b in zip(got, ref))
            out.append('%s %.1e' % (name, d))
        print('N=%d nx=%d B=%d %s: %s' % (N, nx, B, method, ' | '.join(out)), flush=True)
