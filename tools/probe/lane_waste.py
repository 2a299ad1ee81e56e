#!/usr/bin/env python3
"""How many record bytes do finished operating points stream while they wait for the slowest point of their wave?  From the iteration
counts of a launch and the slot order it ran in: waste = sum over waves (points per wave x max - sum) / sum.
usage: python tools/probe/lane_waste.py ["N NX B steps" ...]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench


def waste(its, order, per_wave):
    v = its[order] if len(order) == len(its) else its
    pad = (-len(v)) % per_wave
    w = np.concatenate([v, np.zeros(pad, v.dtype)]).reshape(-1, per_wave)
    return float((w.max(axis=1) * per_wave).sum() - v.sum()) / float(v.sum()), int(w.max()), float(v.mean())


def main():
    shapes = [tuple(int(v) for v in a.split()) for a in sys.argv[1:]] or [(8, 512, 32768, 20), (8, 512, 32768, 6), (8, 512, 8192, 20), (6, 1024, 32768, 10)]
    for N, nx, B, steps in shapes:
        per_wave = 8 if (896 <= B < 10240) else (16 if B < 14336 else 32)
        s, inp = bench.newton_solver(B, N, nx, 4446, 0, steric=True)
        s.set_batch(*inp[1:])
        s.step(1)
        s.synchronize()
        row = {'N': N, 'nx': nx, 'B': B, 'steps': steps, 'points_per_wave': per_wave}
        for tag in ('ordered_by_a_1_step_call', 'ordered_by_the_previous_%d_step_call' % steps):
            ms = bench.timed_steps(s, steps, 0)
            its = s.newton_iterations().astype(np.int64)
            order = s.lane_order().astype(np.int64)
            w, mx, mean = waste(its, order, per_wave)
            w0, _, _ = waste(its, np.arange(B), per_wave)
            row[tag] = {'timesteps_per_s': B * steps / (ms * 1e-3), 'waste': w, 'waste_unordered': w0, 'slowest_point': mx, 'mean_iterations': mean,
                        'launch_bound_by_slowest_point_over_mean': mx / mean}
        s.close()
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
