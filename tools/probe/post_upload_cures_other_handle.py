#!/usr/bin/env python3
"""After s.set_batch: does a medium kernel on ANOTHER handle/queue cure s's slow launches (device-wide state) or not (queue/buffer state)?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from catint_amd.synthetic import make_batch
from catint_amd.host import solver_from_problem
os.environ['CATINT_PNP_NO_POST_UPLOAD_DISPATCH'] = '1'
prob, c0, pb, vz, fl = make_batch(1024, 3, 512, seed=1000, phi_max=0.025, dt_factor=1e-5)
s = solver_from_problem(prob, 'Crank-Nicolson', batch_capacity=1024)
s2 = solver_from_problem(prob, 'Crank-Nicolson', batch_capacity=1024)
s.set_batch(c0, pb, vz, fl); s2.set_batch(c0, pb, vz, fl)


def run(tag, cure):
    for _ in range(10):
        s.step(256, 256)
    s.synchronize()
    s.set_batch(c0, pb, vz, fl)
    cure()
    out = []
    for i in range(5):
        s.timer_start(); s.step(64, 64); out.append(s.timer_stop() * 1e3 / 64)
    print('%-50s %s' % (tag, ' '.join('%.2f' % v for v in out)), flush=True)


for rep in range(2):
    run('nothing', lambda: None)
    run('other handle: 1-step launch + sync', lambda: (s2.step(1, 1), s2.synchronize()))
    run('other handle: 64-step launch + sync', lambda: (s2.step(64, 64), s2.synchronize()))
    run('other handle: get_surface', lambda: s2.get_surface())
    run('this handle: 1-step launch + sync', lambda: (s.step(1, 1), s.synchronize()))
    run('this handle: get_state (poisson kernel + download)', lambda: s.get_state())
s.close(); s2.close()
