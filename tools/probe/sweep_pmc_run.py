#!/usr/bin/env python3
"""One short run of the physical mode on a large-block shape for rocprofv3 --pmc (see sweep_pmc.sh): N, nx, B, kernel from argv."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
N, nx, B, kern = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
os.environ['CATINT_NEWTON_KERNEL'] = kern
import bench
s, inp = bench.newton_solver(B, N, nx, 4444, 0, steric=True)
s.set_batch(*inp[1:])
s.step(1)
s.step(2)
s.synchronize()
print('iterations', int(s.newton_iterations().sum()), 'ok', int((s.get_status() == 0).sum()))
s.close()
