import os, sys, json
import numpy as np
sys.path.insert(0, '/root/repo')
os.environ['CATINT_NEWTON_KERNEL'] = 'lane2'
from catint_amd import _capi
from catint_amd.synthetic import make_batch
N, nx, B, steps = 8, 512, 8192, 10
prob, c0, pb, vz, fl = make_batch(B, N, nx, seed=0, phi_max=0.2, dt_factor=0.1)
radii = [4.1e-10, 3.6e-10, 3.3e-10, 3e-10, 3e-10, 3e-10, 4.5e-10, 3.5e-10][:N]
with _capi.PnpSolver(prob.N, prob.nx, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Newton', batch_capacity=B) as s:
    s.set_newton(wall_bc='stern', stern_capacitance=0.2, tol=1e-8, mpb_radius=radii)
    s.set_batch(c0, np.nan_to_num(pb), vz, fl)
    per = []
    for k in range(steps):
        s.step(1)
        per.append(s.newton_iterations().copy())
per = np.array(per)      # [steps][B]
tot = per.sum(axis=0)
print('per-step mean', per.mean(axis=1).round(2), 'max', per.max(axis=1))
print('total mean %.2f max %d p50 %d p90 %d p99 %d' % (tot.mean(), tot.max(), *np.percentile(tot, [50, 90, 99])))
print('sum over steps of per-step max', per.max(axis=1).sum(), ' vs mean total', tot.mean())
for frac in (0.01, 0.02, 0.05, 0.1, 0.2):
    k = int(B * (1 - frac))
    idx = np.argsort(tot)[:k]
    print('drop top %.0f%%: max total of the rest %d (ratio to mean %.3f)' % (frac * 100, tot[idx].max(), tot[idx].max() / tot.mean()))
print('corr phiM vs tot', np.corrcoef(np.abs(pb[:, 0]), tot)[0, 1])
