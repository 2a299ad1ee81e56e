import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from catint_amd.synthetic import make_batch
from catint_amd.host import solver_from_problem

def run(p, method, c0, pb, vz, fl, nsteps, spl):
    with solver_from_problem(p, method, batch_capacity=c0.shape[0]) as s:
        s.set_batch(c0, pb, vz, fl)
        s.step(nsteps, spl)
        return s.get_state() + (s.get_status(),)

N, nx, B = [int(x) for x in sys.argv[1:4]]
method = sys.argv[4] if len(sys.argv) > 4 else 'Crank-Nicolson'
p, c0, pb, vz, fl = make_batch(B, N, nx, seed=nx + N, phi_max=0.02, dt_factor=1e-4)
rng = np.random.default_rng(nx)
c0 = c0 * (1 + 0.05 * rng.uniform(-1, 1, c0.shape))
fl = rng.uniform(-1e-4, 1e-4, fl.shape)
for k in ('CATINT_PNP_KERNEL', 'CATINT_PNP_ST_WAVES_PER_CU'):
    os.environ.pop(k, None)
for nsteps in (1, 4):
    os.environ.pop('CATINT_PNP_KERNEL', None)
    ref = run(p, method, c0, pb, vz, fl, nsteps, 1)
    os.environ['CATINT_PNP_KERNEL'] = '4'
    rr = run(p, method, c0, pb, vz, fl, nsteps, 1)
    print('nsteps', nsteps, 'rr vs default equal:', [bool(np.array_equal(a, b)) for a, b in zip(rr[:4], ref[:4])])
    for mode in ('5', '6'):
        os.environ['CATINT_PNP_KERNEL'] = mode
        for wcu in ('1', None):
            if wcu: os.environ['CATINT_PNP_ST_WAVES_PER_CU'] = wcu
            else: os.environ.pop('CATINT_PNP_ST_WAVES_PER_CU', None)
            for spl in (1, nsteps):
                got = run(p, method, c0, pb, vz, fl, nsteps, spl)
                out = []
                for name, a, r in zip('cvgl', got[:4], rr[:4]):
                    d = np.abs(a - r)
                    bad = np.argwhere(d > 0)
                    out.append('%s: max %.2e nbad %d first %s' % (name, d.max() / np.abs(r).max(), len(bad), bad[:3].tolist()))
                print('mode', mode, 'wcu', wcu, 'spl', spl, 'status', int((got[4] != 0).sum()), ' | '.join(out))
