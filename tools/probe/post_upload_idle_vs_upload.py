#!/usr/bin/env python3
"""Is the slow start after pnp_set_batch caused by the upload or by the idle time it implies?  Settle, then (a) sleep for a while
with NO device work and time five 64-step launches one by one, (b) upload and do the same, (c) upload, then keep the device busy
with throw-away launches shorter than the slow phase, then time."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
os.environ['CATINT_PNP_NO_POST_UPLOAD_DISPATCH'] = '1'      # show the effect pnp_set_batch's Poisson dispatch removes
from catint_amd.synthetic import make_batch
from catint_amd.host import solver_from_problem
prob, c0, pb, vz, fl = make_batch(1024, 3, 512, seed=1000, phi_max=0.025, dt_factor=1e-5)
s = solver_from_problem(prob, 'Crank-Nicolson', batch_capacity=1024)
s.set_batch(c0, pb, vz, fl)


def settle():
    for _ in range(40):
        s.step(256, 256)
    s.synchronize()


def five(tag):
    out = []
    for i in range(5):
        s.timer_start(); s.step(64, 64); ms = s.timer_stop()
        out.append(ms * 1e3 / 64)
    print('%-34s %s us/step' % (tag, ' '.join('%.2f' % v for v in out)), flush=True)


for rep in range(2):
    for gap in (0.0, 0.0005, 0.002, 0.01, 0.05, 0.3):
        settle()
        time.sleep(gap)
        five('idle %.4f s, no upload' % gap)
    settle()
    t0 = time.perf_counter(); s.set_batch(c0, pb, vz, fl); up = time.perf_counter() - t0
    five('upload (%.1f ms)' % (up * 1e3))
    settle()
    s.set_batch(c0, pb, vz, fl)
    time.sleep(0.01)
    five('upload + 10 ms idle')
s.close()
