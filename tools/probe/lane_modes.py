#!/usr/bin/env python3
"""The lane kernel's rate at 32 768 points varies between PROCESSES on some boxes (1.69e6 / 1.83e6 with the same library).  Does it vary
between solver INSTANCES of one process (fresh workspace allocation each time), and between repetitions on one instance?
usage: python tools/probe/lane_modes.py [instances] [reps]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench


def main():
    n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = []
    for k in range(n_inst):
        s, inp = bench.newton_solver(32768, 8, 512, 4446, 0, steric=True)
        rates = []
        for _ in range(reps):
            s.set_batch(*inp[1:])
            s.step(1)
            s.synchronize()
            ms = bench.timed_steps(s, 20, 0)
            rates.append(round(32768 * 20 / (ms * 1e-3) / 1e6, 3))
        s.close()
        del inp
        rows.append(rates)
    print(json.dumps({'pid': os.getpid(), 'rates_per_instance_Mtimesteps_per_s': rows}), flush=True)


if __name__ == '__main__':
    main()
