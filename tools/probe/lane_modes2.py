#!/usr/bin/env python3
"""Does the rate mode of the lane kernel at 32 768 points (1.52 / 1.66 / 1.82e6 timesteps/s on some boxes, per process) follow the PLACEMENT
of its 16 GB workspace?  One process: before each solver instance a dummy device allocation of a different size is made (and kept), so
the workspace lands elsewhere; two timings per instance.
usage: python tools/probe/lane_modes2.py [instances] [nx]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench


def main():
    n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    nx = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    rows = []
    keep = []
    for k in range(n_inst):
        pad = int(k * 1.37e9) + k * 4096 * 17
        if pad:
            keep.append(torch.empty(pad, dtype=torch.uint8, device='cuda'))
        s, inp = bench.newton_solver(32768, 8, nx, 4446, 0, steric=True)
        rates = []
        for _ in range(2):
            s.set_batch(*inp[1:])
            s.step(1)
            s.synchronize()
            ms = bench.timed_steps(s, 20, 0)
            rates.append(round(32768 * 20 / (ms * 1e-3) / 1e6 * nx / 512.0, 3))      # (per 512 rows)
        s.close()
        del inp
        rows.append({'pad_GB': round(pad / 1e9, 2), 'rates': rates})
    print(json.dumps({'pid': os.getpid(), 'nx': nx, 'rates_per_512_rows': [r['rates'][0] for r in rows]}), flush=True)


if __name__ == '__main__':
    main()
