#!/usr/bin/env python3
"""Compat integrator rates of ONE library build (CATINT_PNP_LIB) on the bench shapes: headline (1024 x 3 x 512, 20 and 256 steps per launch),
8192 x 3 x 512 per step, one GPU's share of configs[3] per step and fused.  tools/probe/compat_ab.sh alternates builds in one call."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench


def rate(B, N, nx, nsteps, spl, reps=9, warm=6):
    s, inp = bench.compat_solver(B, N, nx, 'Crank-Nicolson', 5)
    s.set_batch(*inp[1:])
    for _ in range(warm):
        s.step(nsteps, spl)
    s.synchronize()
    ms = bench.timed_steps(s, nsteps, spl, reps=reps)
    ok = int((s.get_status() == 0).sum())
    c = s.get_state(potential=False)
    s.close()
    return B * nsteps / (ms * 1e-3), ok, float(np.abs(c).sum())


def main():
    out = {}
    for name, args in (('headline_20', (1024, 3, 512, 20, 20)), ('headline_256', (1024, 3, 512, 256, 256)), ('n4_1024_fused', (1024, 4, 512, 64, 64)),
                       ('per_step_8192', (8192, 3, 512, 20, 1)), ('configs3_per_step', (32768, 6, 1024, 8, 1)), ('configs3_fused', (32768, 6, 1024, 32, 32)),
                       ('nx128_fused', (4096, 3, 130, 64, 64))):
        r, ok, chk = rate(*args)
        out[name] = {'timesteps_per_s': r, 'ok': ok, 'checksum': chk}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
