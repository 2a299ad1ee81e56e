#!/usr/bin/env python3
"""One configuration of the streaming probe in one process (for rocprofv3):  stream_run.py N nx B variant spl nlaunch"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tools.probe.stream_probe import VARIANTS, KEYS, measure      # noqa: E402
from catint_amd.synthetic import make_batch                      # noqa: E402

N, nx, B = [int(x) for x in sys.argv[1:4]]
variant, spl, nl = sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
for k in KEYS:
    os.environ.pop(k, None)
os.environ.update(VARIANTS[variant])
prob, c0, pb, vz, fl = make_batch(B, N, nx, seed=0, phi_max=0.025, dt_factor=1e-5)
us, frac, ok = measure(prob, c0, pb, vz, fl, B, N, nx, spl, spl * nl, 3)
print('N=%d nx=%d B=%d %s spl=%d: %.1f us/step frac %.3f ok %d' % (N, nx, B, variant, spl, us, frac, ok))
