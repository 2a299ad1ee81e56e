#!/usr/bin/env python3
"""Stationary solves from the bulk state (damped first iterations): lane kernel with the update fused into the back-substitution or not.
usage: python tools/probe/stationary_fused_ab.py "N NX B" ..."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench


def main():
    for spec in sys.argv[1:]:
        N, nx, B = (int(v) for v in spec.split())
        row = {'N': N, 'nx': nx, 'B': B}
        for fused in ('0', '1', '0', '1'):
            s, inp = bench.newton_solver(B, N, nx, 4446, 0, steric=N >= 5)
            s.set_option('NEWTON_KERNEL', 'lane')
            s.set_option('LANE_FUSED', fused)
            s.set_batch(*inp[1:])
            s.solve_stationary()                 # warm-up (workspace, lane order by iteration counts)
            s.set_batch(*inp[1:])
            s.synchronize()
            t0 = time.perf_counter()
            st = s.solve_stationary()
            dt = time.perf_counter() - t0
            it = s.newton_iterations()
            row.setdefault('fused' if fused == '1' else 'separate', []).append(round(dt * 1e3, 2))
            row['iterations_mean_max'] = [float(it.mean()), int(it.max())]
            row['ok'] = int((st == 0).sum())
            s.close()
            del inp
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
