"""Row chunks on separate HIP streams (CATINT_PNP_STEP_STREAMS) and alternating row order (CATINT_PNP_ALTERNATE_ROWS) for pnp_step calls of one launch per timestep: timesteps/s and fraction
of the HBM roofline at 1..4 chunks, several shapes, all in ONE process on one device (devices of the pool differ by ~10 %).
Usage: python tools/probe/step_streams.py [out.jsonl]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench

SHAPES = [(32768, 6, 1024, 8), (8192, 4, 512, 20), (16384, 4, 512, 20), (4096, 6, 1024, 8), (16384, 8, 256, 10), (2048, 4, 512, 20), (1024, 4, 512, 20)]


def main():
    out = open(sys.argv[1], 'w') if len(sys.argv) > 1 else None
    for B, N, nx, steps in SHAPES:
        for S, alt in ((1, 0), (1, 1), (2, 0), (2, 1), (3, 0), (4, 0), (1, 0), (1, 1), (2, 0), (2, 1)):
            os.environ['CATINT_PNP_STEP_STREAMS'] = str(S)
            os.environ['CATINT_PNP_ALTERNATE_ROWS'] = str(alt)
            s, inp = bench.compat_solver(B, N, nx, 'Crank-Nicolson', 5)
            s.set_batch(*inp[1:])
            for _ in range(6):
                s.step(steps, 1)
            s.synchronize()
            ms = bench.timed_steps(s, steps, 1, reps=7)
            ok = int((s.get_status() == 0).sum())
            s.close()
            rate = B * steps / (ms * 1e-3)
            frac = 16.0 * (N + 1) * nx * rate / 1e9 / bench.HBM_PEAK_GBS
            rec = {'B': B, 'N': N, 'nx': nx, 'steps': steps, 'chunks': S, 'alternate': alt, 'us_per_step': ms / steps * 1e3, 'timesteps_per_s': rate,
                   'frac': frac, 'lanes_ok': ok}
            print(json.dumps(rec), flush=True)
            if out:
                out.write(json.dumps(rec) + '\n')
                out.flush()


if __name__ == '__main__':
    main()
