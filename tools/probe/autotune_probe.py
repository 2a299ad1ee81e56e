#!/usr/bin/env python3
"""pnp_autotune over batch sizes around the kernel-family crossovers: time per timestep of every family on this device, the fastest,
and what the library's thresholds choose.  usage: python tools/probe/autotune_probe.py ["N NX B" ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench


def main():
    shapes = [tuple(int(v) for v in a.split()) for a in sys.argv[1:]] or \
        [(8, 512, b) for b in (512, 1024, 2048, 4096, 8192, 10240, 12288, 14336, 16384, 24576, 32768)] + \
        [(6, 1024, b) for b in (2048, 8192, 16384, 32768)] + [(3, 512, b) for b in (1024, 8192, 32768)] + [(8, 4096, 8192)]
    for N, nx, B in shapes:
        s, inp = bench.newton_solver(B, N, nx, 4446, 0, steric=N >= 5)
        s.set_batch(*inp[1:])
        del inp
        s.step(1)
        s.synchronize()
        lib = s.default_family()
        fastest, ms = s.autotune(6)
        s.close()
        print(json.dumps({'N': N, 'nx': nx, 'B': B, 'ms_per_timestep': {k: round(v, 4) for k, v in ms.items()}, 'fastest': fastest,
                          'library_choice': lib, 'library_choice_over_fastest': round(ms[lib] / ms[fastest], 4)}), flush=True)


if __name__ == '__main__':
    main()
