#!/usr/bin/env python3
"""Lane-quad kernel: start stagger of the four wave groups (option LANE_STAGGER, units of ~3.4 us), alternated on ONE handle (same
workspace, same process): timesteps/s of 20-step launches from the same state.
usage: python tools/probe/lane4_stagger_same_handle.py "N NX B" [stagger ...]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench


def main():
    N, nx, B = (int(v) for v in sys.argv[1].split()) if len(sys.argv) > 1 else (8, 512, 8192)
    vals = [int(v) for v in sys.argv[2:]] or [0, 40, 80, 160, 320, 640]
    steps = 20 if nx <= 512 else 4
    s, inp = bench.newton_solver(B, N, nx, 4444, 0, steric=True)
    res = {v: [] for v in vals}
    for rep in range(5):
        for v in vals:
            s.set_option('LANE_STAGGER', str(v))
            s.set_batch(*inp[1:])
            s.step(1)
            s.synchronize()
            ms = bench.timed_steps(s, steps, 0)
            res[v].append(B * steps / (ms * 1e-3))
    s.close()
    print(json.dumps({'shape': [N, nx, B], 'median_timesteps_per_s': {str(v): round(float(np.median(r))) for v, r in res.items()},
                      'min_max': {str(v): [round(min(r)), round(max(r))] for v, r in res.items()}}))


if __name__ == '__main__':
    main()
