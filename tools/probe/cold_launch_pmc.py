#!/usr/bin/env python3
"""What is different about the first launches after pnp_set_batch?  Run under rocprofv3 --pmc: upload, then five 64-step launches,
twice; the per-dispatch counters of launch 1 vs launch 5 tell HBM traffic / L2 behaviour / clocks apart."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from catint_amd.synthetic import make_batch
from catint_amd.host import solver_from_problem
prob, c0, pb, vz, fl = make_batch(1024, 3, 512, seed=1000, phi_max=0.025, dt_factor=1e-5)
s = solver_from_problem(prob, 'Crank-Nicolson', batch_capacity=1024)
s.set_batch(c0, pb, vz, fl)
for _ in range(40):
    s.step(256, 256)
s.synchronize()
for rep in range(2):
    s.set_batch(c0, pb, vz, fl)
    for i in range(5):
        s.timer_start(); s.step(64, 64); ms = s.timer_stop()
        print('rep %d launch %d: %.1f us/step' % (rep, i, ms * 1e3 / 64))
s.close()
