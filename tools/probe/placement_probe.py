#!/usr/bin/env python3
"""pnp_tune_placement on the 32 768-point record: rate before, the trials, rate after; several solver instances (placements) per process."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench


def main():
    shape = tuple(int(v) for v in sys.argv[1].split()) if len(sys.argv) > 1 else (8, 512, 32768)
    N, nx, B = shape
    keep = []
    for k in range(4):
        pad = int(k * 1.37e9)
        if pad:
            keep.append(torch.empty(pad, dtype=torch.uint8, device='cuda'))
        s, inp = bench.newton_solver(B, N, nx, 4446, 0, steric=True)

        def rate():
            s.set_batch(*inp[1:])
            s.step(1)
            s.synchronize()
            ms = bench.timed_steps(s, 20 if nx <= 512 else 10, 0)
            return round(B * (20 if nx <= 512 else 10) / (ms * 1e-3) / 1e6, 3)
        before = rate()
        s.set_batch(*inp[1:])
        s.step(1)
        trials = s.tune_placement(2, 4)
        after = [rate(), rate()]
        s.close()
        del inp
        print(json.dumps({'instance': k, 'before': before, 'trials_ms_per_step': [round(t, 2) for t in trials], 'after': after}), flush=True)


if __name__ == '__main__':
    main()
