import os, sys, json
import numpy as np
sys.path.insert(0, '/root/repo')
from catint_amd import _capi
from catint_amd.synthetic import make_batch
def run(B, est, steps=10, kernel='lane'):
    os.environ['CATINT_NEWTON_KERNEL'] = kernel
    N, nx = 8, 512
    prob, c0, pb, vz, fl = make_batch(B, N, nx, seed=0, phi_max=0.2, dt_factor=0.1)
    radii = [4.1e-10, 3.6e-10, 3.3e-10, 3e-10, 3e-10, 3e-10, 4.5e-10, 3.5e-10][:N]
    with _capi.PnpSolver(prob.N, prob.nx, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Newton', batch_capacity=B) as s:
        s.set_newton(wall_bc='stern', stern_capacitance=0.2, tol=1e-8, mpb_radius=radii, error_estimate=est)
        s.set_batch(c0, np.nan_to_num(pb), vz, fl)
        s.step(2)
        s.synchronize()
        s.timer_start()
        s.step(steps)
        ms = s.timer_stop()
        it = s.newton_iterations()
    lanes = 32
    g = it[:B // lanes * lanes].reshape(-1, lanes)
    gs = np.sort(it)[::-1][:B // lanes * lanes].reshape(-1, lanes)
    return {'B': B, 'est': est, 'steps_per_s': B * steps / (ms * 1e-3), 'its_per_step': float(it.mean()) / steps, 'it_hist': np.bincount(it).tolist()[-8:], 'min': int(it.min()), 'max': int(it.max()),
            'wave_waste_sorted': float(gs.max(axis=1).sum() * lanes / it.sum()), 'lane_it_per_s': float(it.sum()) / (ms * 1e-3)}
for B in (32768, 8192):
    for est in (False, True, False, True):
        print(json.dumps(run(B, est, kernel='lane' if B > 10000 else 'lane2')), flush=True)
