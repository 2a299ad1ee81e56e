#!/usr/bin/env python3
"""Lane kernel with the record columns in single precision (option LANE_RECORDS = f32) against double, workspace placed in both:
timesteps/s, iterations per step, lanes converged, difference of the final states.  usage: python tools/probe/f32_records_probe.py ["N NX B STEPS" ...]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench


def main():
    shapes = [tuple(int(v) for v in a.split()) for a in sys.argv[1:]] or [(8, 512, 32768, 20), (6, 1024, 32768, 10), (8, 512, 24576, 20)]
    for N, nx, B, steps in shapes:
        row = {'N': N, 'nx': nx, 'B': B, 'steps': steps}
        states = {}
        for fmt in ('f64', 'f32'):
            for ee in (False, True):
                s, inp = bench.newton_solver(B, N, nx, 4446, 0, steric=True, error_estimate=ee)
                s.set_option('NEWTON_KERNEL', 'lane')
                s.set_option('LANE_RECORDS', fmt)
                s.set_batch(*inp[1:])
                s.step(1)
                s.tune_placement(2, 6)
                s.synchronize()
                ms = bench.timed_steps(s, steps, 0)
                it = s.newton_iterations()
                key = fmt + ('+error_estimate' if ee else '')
                row[key] = {'timesteps_per_s': round(B * steps / (ms * 1e-3)), 'iterations_per_step': round(float(it.sum()) / (B * steps), 4),
                            'ok': int((s.get_status() == 0).sum())}
                if not ee:
                    states[fmt] = (s.get_state()[0][:2048].copy(), it.copy())
                s.close()
                del inp
        a, b = states['f64'], states['f32']
        row['state_rel_diff_f32_vs_f64'] = float(np.abs(a[0] - b[0]).max() / np.abs(a[0]).max())
        row['lanes_with_other_iteration_count'] = int((a[1] != b[1]).sum())
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
