#!/usr/bin/env python3
"""Timesteps/s of one Newton kernel family on bench.py's large-batch shapes (N = 8 steric, Stern wall, 20 steps in one launch), several
repetitions in one process.  usage: python tools/probe/lane_rate.py KERNEL "N NX B STEPS" ..."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench


def main():
    kern = sys.argv[1]
    os.environ['CATINT_NEWTON_KERNEL'] = kern
    for spec in sys.argv[2:]:
        N, nx, B, steps = (int(v) for v in spec.split())
        s, inp = bench.newton_solver(B, N, nx, 4446, 0, steric=True)
        rates = []
        if os.environ.get('LANE_RATE_TUNE'):          # (the workspace placed first: pnp_tune_placement)
            s.set_batch(*inp[1:])
            s.step(1)
            s.tune_placement(2, 4)
        for _ in range(3):
            s.set_batch(*inp[1:])
            s.step(1)
            s.synchronize()
            ms = bench.timed_steps(s, steps, 0)
            it = s.newton_iterations()
            ok = int((s.get_status() == 0).sum())
            rates.append(B * steps / (ms * 1e-3))
        s.close()
        print(json.dumps({'kernel': kern, 'N': N, 'nx': nx, 'B': B, 'steps': steps, 'timesteps_per_s': rates, 'iterations_per_step': float(it.sum()) / (B * steps),
                          'lanes_ok': ok}), flush=True)


if __name__ == '__main__':
    main()
