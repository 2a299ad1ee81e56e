#!/usr/bin/env python3
"""Timesteps/s of the lane-kernel families on the bench's workload (one warm-up step, then STEPS steps in one call, best of 2), one device,
one process -- the measurement behind newton_lane*_preferred and the lane kernel's fused flag.
usage: python tools/probe/family_rates.py [--families lane+fused,workgroup] "N NX B STEPS" ..."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench

FAMILIES = (('lane4', {'NEWTON_KERNEL': 'lane4'}), ('lane2', {'NEWTON_KERNEL': 'lane2'}), ('lane', {'NEWTON_KERNEL': 'lane', 'LANE_FUSED': '0'}),
            ('lane+fused', {'NEWTON_KERNEL': 'lane', 'LANE_FUSED': '1'}), ('workgroup', {'NEWTON_KERNEL': 'workgroup'}))


def main():
    argv = sys.argv[1:]
    only = None
    if argv and argv[0] == '--families':
        only = set(argv[1].split(','))
        argv = argv[2:]
    for spec in argv:
        N, nx, B, steps = (int(v) for v in spec.split())
        row = {'N': N, 'nx': nx, 'B': B, 'steps': steps}
        for name, opts in FAMILIES:
            if only is not None and name not in only:
                continue
            if name == 'lane4' and N < 5:
                continue
            if only is None and name == 'workgroup' and B * nx * N > 8192 * 512 * 8:
                continue
            s, inp = bench.newton_solver(B, N, nx, 4446, 0, steric=N >= 5)
            for k, v in opts.items():
                s.set_option(k, v)
            best = 0.0
            for _ in range(2):
                s.set_batch(*inp[1:])
                s.step(1)
                s.synchronize()
                ms = bench.timed_steps(s, steps, 0)
                best = max(best, B * steps / (ms * 1e-3))
            ok = int((s.get_status() == 0).sum())
            s.close()
            del inp
            row[name] = round(best)
            if ok != B:
                row[name + '_ok'] = ok
        fam = {k: v for k, v in row.items() if k in dict(FAMILIES)}
        row['fastest'] = max(fam, key=fam.get)
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
