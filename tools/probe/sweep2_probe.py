#!/usr/bin/env python3
"""Physical mode, large blocks: lane-team kernel vs one-sided sweep vs two-sided sweep over the batch size
(synthetic inputs of bench.py: phiM ~ U(+-0.2 V), dt = 0.1 lambda_D L / D_max, steric ions, Stern wall, tol 1e-8)."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import bench

shapes = [(8, 512, B) for B in (2048, 4096, 8192, 16384, 32768)] + [(6, 1024, B) for B in (4096, 8192, 16384, 32768)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for (N, nx, B) in shapes:
    row = {'N': N, 'nx': nx, 'B': B}
    ref = None
    for kern in ('team', 'sweep', 'both'):
        os.environ['CATINT_NEWTON_KERNEL'] = kern
        s, inp = bench.newton_solver(B, N, nx, 4444, 0, steric=True)
        s.set_batch(*inp[1:])
        s.step(1)
        s.synchronize()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.15:      # clocks
            s.step(1)
            s.synchronize()
        s.set_batch(*inp[1:])
        s.step(1)
        s.timer_start(); s.step(3); ms = s.timer_stop()
        it = s.newton_iterations()
        ok = int((s.get_status() == 0).sum())
        c = s.get_state()[0]
        s.close()
        row[kern] = {'timesteps_per_s': B * 3 / (ms * 1e-3), 'iterations_per_s': float(it.sum()) / (ms * 1e-3), 'ok': ok}
        if ref is None:
            ref = (c, it)
        else:
            row[kern]['max_rel_diff_vs_team'] = float(np.abs(c - ref[0]).max() / np.abs(ref[0]).max())
            row[kern]['same_iterations'] = bool(np.array_equal(it, ref[1]))
    print(json.dumps(row), flush=True)
