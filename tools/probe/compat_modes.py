#!/usr/bin/env python3
"""Does the compat step's rate on a beyond-cache state (configs[3] share: 32768 x 6 x 1024, 2.1 GB, one launch per timestep) depend on
where the handle's buffers lie?  One process, instances created after dummy allocations of different sizes."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench


def main():
    keep, out = [], []
    for k in range(6):
        pad = int(k * 0.77e9)
        if pad:
            keep.append(torch.empty(pad, dtype=torch.uint8, device='cuda'))
        s, inp = bench.compat_solver(32768, 6, 1024, 'Crank-Nicolson', 55, 0)
        s.set_batch(*inp[1:])
        del inp
        s.step(4, 1)
        s.synchronize()
        r = []
        for _ in range(2):
            s.timer_start()
            s.step(8, 1)
            ms = s.timer_stop()
            r.append(round(32768 * 8 / (ms * 1e-3) / 1e7, 3))
        s.close()
        out.append(r)
    print(json.dumps({'pid': os.getpid(), 'rates_1e7_timesteps_per_s': out}), flush=True)


if __name__ == '__main__':
    main()
