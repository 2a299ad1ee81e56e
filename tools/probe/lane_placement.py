#!/usr/bin/env python3
"""Does the lane kernel's rate depend on WHERE its workspace lies?  The rate at 8 x 512 x 32768 differs between processes on one device
(1.25-1.36e6 or 1.45-1.66e6 timesteps/s, constant inside a process).  Inside ONE process: device allocations of several
sizes are made (and kept) before the solver is created, which moves its buffers; the same launch is timed for every placement.
usage: python tools/probe/lane_placement.py [B]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench


def main():
    import torch
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    os.environ['CATINT_NEWTON_KERNEL'] = 'lane'
    pads = [0, 1 << 20, 3 << 20, (17 << 20) + 4096, 64 << 20, (1 << 30) + (2 << 20), 0, (5 << 30)]
    keep = []
    for pad in pads:
        if pad:
            keep.append(torch.empty(pad, dtype=torch.uint8, device='cuda'))
        s, inp = bench.newton_solver(B, 8, 512, 4446, 0, steric=True)
        rates = []
        for _ in range(2):
            s.set_batch(*inp[1:])
            s.step(1)
            s.synchronize()
            ms = bench.timed_steps(s, 20, 0)
            rates.append(B * 20 / (ms * 1e-3))
        s.close()
        print(json.dumps({'pad_bytes_added_before': pad, 'held_allocations': len(keep), 'timesteps_per_s': rates}), flush=True)


if __name__ == '__main__':
    main()
