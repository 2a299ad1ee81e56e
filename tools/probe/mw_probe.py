"""step_kernel_mw (grids beyond 1026 points: several waves per tridiagonal system), one launch per timestep and fused, at one GPU's
share of BASELINE configs[4] and at a 2050-point grid: timesteps/s and fraction of the HBM roofline.
Usage: python tools/probe/mw_probe.py [out.jsonl]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench

SHAPES = [(8192, 8, 4096, 4), (2048, 8, 4096, 8), (8192, 6, 2050, 6), (512, 4, 4098, 16)]


def main():
    out = open(sys.argv[1], 'w') if len(sys.argv) > 1 else None
    for B, N, nx, steps in SHAPES:
        s, inp = bench.compat_solver(B, N, nx, 'Crank-Nicolson', 5)
        s.set_batch(*inp[1:])
        del inp
        for spl in (1, steps):
            for _ in range(3):
                s.step(steps, spl)
            s.synchronize()
            ms = bench.timed_steps(s, steps, spl, reps=5)
            rate = B * steps / (ms * 1e-3)
            rec = {'B': B, 'N': N, 'nx': nx, 'steps': steps, 'steps_per_launch': spl, 'row_chunks': s.step_row_chunks(steps // spl),
                   'us_per_step': ms / steps * 1e3, 'timesteps_per_s': rate, 'frac': 16.0 * (N + 1) * nx * rate / 1e9 / bench.HBM_PEAK_GBS,
                   'lanes_ok': int((s.get_status() == 0).sum())}
            print(json.dumps(rec), flush=True)
            if out:
                out.write(json.dumps(rec) + '\n')
                out.flush()
        s.close()


if __name__ == '__main__':
    main()
