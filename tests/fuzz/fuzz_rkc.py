"""dev probe: the stiff integrator on the device (pnp_integrate_rkc) against the explicit one (pnp_integrate_dopri5) on random
method-of-lines problems -- species 1 ... 6, grids 20 ... 1500 (incl. two waves per system), batches 1 ... 40, intervals of 0.5 ... 50 times the
explicit stability limit: both at rtol 1e-8, states must agree to 1e-5 (two different formulas, global error ~ 10-50 x rtol each).  Where they
do not, scipy's odeint on the same right-hand side decides ('ok^': the explicit integrator is the one that is off -- on lanes whose
dielectric relaxation makes the problem stiff it runs at its stability limit, 1e-3 from the solution while reporting success)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from catint_amd.synthetic import make_batch
from catint_amd.host import solver_from_problem

rng = np.random.default_rng(int(os.environ.get('FUZZ_SEED', '1')))
ncase = int(os.environ.get('FUZZ_CASES', '30'))
bad = 0
t0 = time.time()
for case in range(ncase):
    N = int(rng.choice([2, 3, 4, 6]))
    nx = int(rng.choice([20, 33, 64, 65, 130, 257, 512, 700, 1100, 1500]))
    B = int(rng.integers(1, 41))
    factor = float(10 ** rng.uniform(-0.3, 1.7))
    nt = int(rng.integers(1, 4))
    prob, c0, pb, vz, fl = make_batch(B, N, nx, seed=7000 + case, phi_max=float(rng.uniform(0.005, 0.05)), dt_factor=1.0)
    prob.dt = factor * prob.dx ** 2 / (2.0 * max(prob.D))
    c0 = c0 * (1 + 0.1 * rng.uniform(-1, 1, c0.shape))
    fl = rng.uniform(-1e-5, 1e-5, fl.shape) * (rng.random() < 0.5)
    try:
        with solver_from_problem(prob, 'FTCS', batch_capacity=B) as s:
            s.set_batch(c0, pb, vz, fl)
            a, ia, sa, _ = s.integrate_rkc(nt, [nt - 1], rtol=1e-8, atol=1e-14)
            s.set_batch(c0, pb, vz, fl)
            b, ib, sb, _ = s.integrate_dopri5(nt, [nt - 1], rtol=1e-8, atol=1e-14, nsteps=200000)
        good = (ia == 1) & (ib == 1)            # (DOPRI5 leaves stiff lanes with IDID -4: nothing to compare there)
        errs = np.array([np.abs(a[0, l] - b[0, l]).max() / np.abs(b[0, l]).max() if good[l] else 0.0 for l in range(B)])
        err = float(errs.max())
        ok = (ia == 1).all() and err < 1e-5
        if not ok and (ia == 1).all():           # who is off?  the worst lane against scipy's odeint (LSODA) on the device right-hand side
            import scipy.integrate as si
            l = int(errs.argmax())
            with solver_from_problem(prob, 'FTCS', batch_capacity=1) as s1:
                s1.set_batch(c0[l:l + 1], pb[l:l + 1], vz[l:l + 1], fl[l:l + 1])
                ref = si.odeint(lambda y, t: s1.mol_rhs(y[None, :])[0], c0[l], np.arange(nt + 1) * prob.dt, rtol=1e-11, atol=1e-15, mxstep=500000)
            sc = np.abs(ref[-1]).max()
            e_rkc, e_dop = np.abs(a[0, l] - ref[-1]).max() / sc, np.abs(b[0, l] - ref[-1]).max() / sc
            print('   lane %d: rkc vs odeint %.1e, dopri5 vs odeint %.1e; rkc stats %s dopri5 stats %s' % (l, e_rkc, e_dop, sa[l], sb[l]))
            if e_rkc < 1e-5 and e_dop > 10 * e_rkc:       # DOPRI5 at its stability limit (scipy's own behaviour: the device code is pinned to it)
                ok, tag = True, 'ok^'
        tag = 'ok ' if ok else 'BAD'
    except Exception as e:
        ok, tag, err, sa, sb, ia, ib = False, 'EXC', -1.0, str(e)[:80], '', '', ''
    bad += 0 if ok else 1
    if not ok or tag != 'ok ' or case % 10 == 0:
        print('%s case %2d N=%d nx=%d B=%d interval=%.2f x explicit limit, nt=%d: relerr %.1e; rkc steps %s stages <= %s, dopri5 steps %s; idid %s %s' % (
            tag, case, N, nx, B, factor, nt, err, sa[:, 0].max() if ok or tag == 'BAD' else sa, sa[:, 6].max() if ok or tag == 'BAD' else '',
            sb[:, 0].max() if ok or tag == 'BAD' else '', set(np.asarray(ia).tolist()) if tag != 'EXC' else '', set(np.asarray(ib).tolist()) if tag != 'EXC' else ''), flush=True)
print('%d cases, %d bad, %.1f s' % (ncase, bad, time.time() - t0))
