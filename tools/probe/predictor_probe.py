#!/usr/bin/env python3
"""Timesteps/s and Newton iterations per step with and without the predictor / BDF2 on bench.py's large-batch shapes (20 steps).
usage: python tools/probe/predictor_probe.py ["N NX B" ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench


def main():
    shapes = [tuple(int(v) for v in a.split()) for a in sys.argv[1:]] or [(8, 512, 8192), (8, 512, 32768), (6, 1024, 32768), (8, 4096, 8192), (3, 512, 1024)]
    for N, nx, B in shapes:
        steps = 8 if nx >= 4096 else 20
        row = {'N': N, 'nx': nx, 'B': B, 'steps': steps}
        for name, kw in (('plain', {}), ('predictor', dict(predictor=True)), ('bdf2', dict(time_order=2)), ('predictor_bdf2', dict(predictor=True, time_order=2)),
                         ('error_estimate', dict(error_estimate=True)), ('predictor_error_estimate', dict(predictor=True, error_estimate=True))):
            s, inp = bench.newton_solver(B, N, nx, 4446, 0, steric=N >= 5, **kw)
            s.set_batch(*inp[1:])
            s.step(2)
            s.synchronize()
            ms = bench.timed_steps(s, steps, 0)
            it = s.newton_iterations()
            row[name] = {'timesteps_per_s': B * steps / (ms * 1e-3), 'iterations_per_step': float(it.sum()) / (B * steps), 'ok': int((s.get_status() == 0).sum())}
            s.close()
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
