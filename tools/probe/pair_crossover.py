#!/usr/bin/env python3
"""N = 4 (5 x 5 blocks): the pair kernel (512-register build, round 3) against the lane kernel and the sweep kernel over grid x batch, one
device, one call -> the N = 4 thresholds of newton_lane_preferred / newton_sweep_preferred.

    python tools/probe/pair_crossover.py [out.jsonl]
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lane_sweep import run


def main():
    out = open(sys.argv[1], 'w') if len(sys.argv) > 1 else None
    for mpb in (True, False):
        for nx in (128, 512, 1024):
            for B in (2048, 4096, 8192, 16384, 32768, 65536):
                if B * nx > 2.4e7:
                    continue
                row = {'N': 4, 'nx': nx, 'B': B, 'mpb': mpb}
                for kern in ('pair', 'lane', 'sweep'):
                    r, its, ok = run(4, nx, B, kern, 6, mpb)
                    row[kern] = r
                    row['its_' + kern] = its
                    row['ok_' + kern] = ok
                print(json.dumps(row), flush=True)
                if out:
                    out.write(json.dumps(row) + '\n')
                    out.flush()


if __name__ == '__main__':
    main()
