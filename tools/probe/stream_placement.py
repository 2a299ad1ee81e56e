#!/usr/bin/env python3
"""Raw streaming over 16 GB windows of one 44 GB allocation at different offsets (torch copy_ / fill_): is the achievable HBM bandwidth of
this device a function of WHERE the window lies?  (The lane kernel's rate at 32 768 points is: tools/probe/lane_modes2.py.)"""
import json
import sys

import torch


def main():
    gb = 1 << 30
    total, win = 44 * gb, 16 * gb
    x = torch.empty(total // 8, dtype=torch.float64, device='cuda')
    x.zero_()
    out = []
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    offs = [float(v) for v in sys.argv[1:]] or [0, 1.37, 2.74, 4.11, 5.48, 6.85, 8.22, 9.59, 12.0, 16.0, 20.0, 24.0, 28.0]
    for off_gb in offs:
        a = int(off_gb * gb) // 8
        n = win // 8
        w = x[a:a + n]
        half = n // 2
        res = {}
        for name, fn in (('fill', lambda: w.fill_(1.0)), ('copy', lambda: w[:half].copy_(w[half:2 * half]))):
            fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(4):
                ev0.record()
                fn()
                ev1.record()
                torch.cuda.synchronize()
                ts.append(ev0.elapsed_time(ev1))
            ms = sorted(ts)[1]
            res[name] = round(win / (ms * 1e-3) / 1e12, 3)          # TB/s moved (fill: written; copy: read + written)
        out.append((off_gb, res))
    print(json.dumps({'TB_per_s_by_offset_GB': out}))


if __name__ == '__main__':
    main()
