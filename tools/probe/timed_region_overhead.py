#!/usr/bin/env python3
"""Where the host-side microseconds of bench.py's timed region go on the headline shape (one launch of 20 fused timesteps, ~106 us of
kernel): median wall time of each call of the region over 2000 repetitions."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    s, inp = bench.compat_solver(1024, 3, 512, 'Crank-Nicolson', 1000, 0)
    s.set_batch(*inp[1:])
    for _ in range(200):
        s.step(steps, steps)
    s.synchronize()
    seg = {k: [] for k in ('timer_start', 'step', 'timer_stop', 'synchronize', 'torch_sync', 'total', 'event_us')}
    pc = time.perf_counter
    for _ in range(2000):
        s.synchronize()
        torch.cuda.synchronize()
        t0 = pc()
        s.timer_start()
        t1 = pc()
        s.step(steps, steps)
        t2 = pc()
        ev = s.timer_stop()
        t3 = pc()
        s.synchronize()
        t4 = pc()
        torch.cuda.synchronize()
        t5 = pc()
        for k, v in zip(('timer_start', 'step', 'timer_stop', 'synchronize', 'torch_sync', 'total'), (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0)):
            seg[k].append(v * 1e6)
        seg['event_us'].append(ev * 1e3)
    print({k: round(float(np.median(v)), 2) for k, v in seg.items()})
    # the same launch with nothing but a launch and one stream synchronisation around it
    only = []
    for _ in range(2000):
        s.synchronize()
        t0 = pc()
        s.step(steps, steps)
        s.synchronize()
        only.append((pc() - t0) * 1e6)
    print({'launch_plus_synchronize_us': round(float(np.median(only)), 2)})
    s.close()


if __name__ == '__main__':
    main()
