#!/usr/bin/env python3
"""Slow launches after pnp_set_batch: device state or the handle's data?  (a) upload into ANOTHER handle, time this one;
(b) a batch that does not fill the chip exactly; (c) download the state, upload the SAME (evolved) state again."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
os.environ['CATINT_PNP_NO_POST_UPLOAD_DISPATCH'] = '1'      # show the effect pnp_set_batch's Poisson dispatch removes
from catint_amd.synthetic import make_batch
from catint_amd.host import solver_from_problem


def series(s, tag, k=64, n=5):
    out = []
    for i in range(n):
        s.timer_start(); s.step(k, k); ms = s.timer_stop()
        out.append(ms * 1e3 / k)
    print('%-58s %s' % (tag, ' '.join('%.2f' % v for v in out)), flush=True)


def settle(s):
    for _ in range(40):
        s.step(256, 256)
    s.synchronize()


prob, c0, pb, vz, fl = make_batch(1024, 3, 512, seed=1000, phi_max=0.025, dt_factor=1e-5)
s = solver_from_problem(prob, 'Crank-Nicolson', batch_capacity=1024)
s2 = solver_from_problem(prob, 'Crank-Nicolson', batch_capacity=1024)
s.set_batch(c0, pb, vz, fl); s2.set_batch(c0, pb, vz, fl)
for rep in range(2):
    settle(s); series(s, 'control')
    settle(s); s2.set_batch(c0, pb, vz, fl); series(s, 'upload into ANOTHER handle, time this one')
    settle(s); s.set_batch(c0, pb, vz, fl); series(s, 'upload initial state into this handle')
    settle(s); s.set_batch(c0, pb, vz, fl); s.step(64, 64); s.step(64, 64); s.step(64, 64); s.synchronize()
    cur = s.get_state()[0].reshape(1024, -1)
    settle(s); s.set_batch(cur, pb, vz, fl); series(s, 'upload a state 192 steps old into this handle')
    settle(s); s.set_batch(cur * (1 + 1e-9), pb, vz, fl); series(s, 'same, perturbed by 1e-9')
for B in (960, 512, 2048):
    p2, c2, pb2, vz2, fl2 = make_batch(B, 3, 512, seed=1000, phi_max=0.025, dt_factor=1e-5)
    with solver_from_problem(p2, 'Crank-Nicolson', batch_capacity=B) as t:
        t.set_batch(c2, pb2, vz2, fl2)
        settle(t); series(t, 'B = %d control' % B)
        settle(t); t.set_batch(c2, pb2, vz2, fl2); series(t, 'B = %d after upload' % B)
s.close(); s2.close()
