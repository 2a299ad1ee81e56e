#!/usr/bin/env python3
"""Slow launches after pnp_set_batch: do trivial dispatches on the handle's own queue use them up?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
os.environ['CATINT_PNP_NO_POST_UPLOAD_DISPATCH'] = '1'      # show the effect pnp_set_batch's Poisson dispatch removes
from catint_amd.synthetic import make_batch
from catint_amd.host import solver_from_problem
prob, c0, pb, vz, fl = make_batch(1024, 3, 512, seed=1000, phi_max=0.025, dt_factor=1e-5)
s = solver_from_problem(prob, 'Crank-Nicolson', batch_capacity=1024)
s.set_batch(c0, pb, vz, fl)


def run(tag, cure):
    for _ in range(40):
        s.step(256, 256)
    s.synchronize()
    s.set_batch(c0, pb, vz, fl)
    cure()
    out = []
    for i in range(6):
        s.timer_start(); s.step(64, 64); out.append(s.timer_stop() * 1e3 / 64)
    print('%-44s %s' % (tag, ' '.join('%.2f' % v for v in out)), flush=True)


for rep in range(2):
    run('nothing', lambda: None)
    run('extra synchronize', lambda: s.synchronize())
    run('1 x get_surface', lambda: s.get_surface())
    run('2 x get_surface', lambda: (s.get_surface(), s.get_surface()))
    run('3 x get_surface', lambda: (s.get_surface(), s.get_surface(), s.get_surface()))
    run('1-step launch', lambda: s.step(1, 1))
    run('2 x 1-step launch + sync', lambda: (s.step(1, 1), s.step(1, 1), s.synchronize()))
    run('4 x 1-step launch', lambda: (s.step(1, 1), s.step(1, 1), s.step(1, 1), s.step(1, 1)))
s.close()
