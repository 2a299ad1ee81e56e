"""dev probe: are the compat kernels bitwise reproducible run to run (incl. the instances that use accumulator registers)?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from catint_amd.synthetic import make_batch
from catint_amd.host import solver_from_problem

def run(B, N, nx, method, nsteps, spl):
    p, c0, pb, vz, fl = make_batch(B, N, nx, seed=9, phi_max=0.02, dt_factor=1e-5 if method == 'Crank-Nicolson' else 2e-6)
    with solver_from_problem(p, method, batch_capacity=B) as s:
        s.set_batch(c0, pb, vz, fl)
        s.step(nsteps, spl)
        return s.get_state()

bad = 0
for (B, N, nx) in [(1024, 3, 512), (256, 6, 1024), (64, 8, 4096), (4096, 3, 512), (512, 2, 200), (128, 3, 2050)]:
    for method in ('Crank-Nicolson', 'FTCS'):
        for spl in (0, 1):
            r = [run(B, N, nx, method, 40, spl) for _ in range(3)]
            same = all(all(np.array_equal(a, b) for a, b in zip(r[0], x)) for x in r[1:])
            bad += 0 if same else 1
            print('B=%d N=%d nx=%d %s spl=%d reproducible=%s' % (B, N, nx, method, spl, same), flush=True)
print('not reproducible:', bad)
