"""dev probe: randomised configurations of the legacy (compat) integrators, GPU vs C oracle."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from oracle import c_oracle as CO
from oracle import pnp_ref as R
from catint_amd.host import solver_from_problem

CO.load()
rng = np.random.default_rng(int(os.environ.get('FUZZ_SEED', '1')))
ncase = int(os.environ.get('FUZZ_CASES', '100'))
F, BETA, EPS = 96485.33289, 1.0 / (8.3144598 * 298.14), 78.36 * 8.854187817e-12
DS = [1.957e-9, 1.185e-9, 2.032e-9, 1.334e-9, 0.923e-9, 5.273e-9, 2.06e-9, 1.792e-9, 9.311e-9, 1.91e-9, 2.23e-9, 1.0e-9, 1.5e-9, 0.8e-9, 2.5e-9, 3e-9]
ZS = [1, -1, -1, 1, -2, -1, 1, -1, 1, 0, 0, 2, -1, 1, 0, -1]
bad = 0
t0 = time.time()
for case in range(ncase):
    N = int(rng.choice([1, 2, 3, 3, 4, 5, 6, 7, 8, 11, 16]))
    nx = int(rng.choice([5, 6, 9, 33, 64, 66, 67, 130, 200, 258, 514, 515, 1026, 1027, 1500, 2050, 2051, 3000, 4098]))
    B = int(rng.integers(1, 5))
    method = str(rng.choice(['Crank-Nicolson', 'FTCS']))
    mode = int(rng.integers(0, 5))
    lf = bool(rng.random() < 0.3)
    mig = True if method == 'Crank-Nicolson' else bool(rng.random() < 0.8)
    D = np.array(DS[:N]); q = np.array(ZS[:N], float) * F
    cb = np.exp(rng.uniform(np.log(1.0), np.log(30.0), (B, N)))
    lam = np.sqrt(EPS / BETA / max((q ** 2 * 10.0).sum(), 1.0))
    dx = lam / rng.uniform(3, 12)
    dt = rng.uniform(0.02, 0.2) * dx * dx / D.max() * (0.2 if method == 'FTCS' else 1.0)
    pbv = np.full((B, 4), np.nan)
    vw, vb = rng.uniform(-0.02, 0.02, B), rng.uniform(-0.002, 0.002, B)
    gw, gb = rng.uniform(-1e5, 1e5, B), rng.uniform(-1e4, 1e4, B)
    if mode == 0: pbv[:, 0], pbv[:, 1] = vw, vb
    if mode == 1: pbv[:, 0], pbv[:, 3] = vw, gb
    if mode == 2: pbv[:, 2], pbv[:, 1] = gw, vb
    if mode == 3: pbv[:, 0], pbv[:, 2] = vw, gw
    if mode == 4: pbv[:, 1], pbv[:, 3] = vb, gb
    vz = rng.uniform(-0.02, 0.02, B)
    fl = rng.uniform(-1e-5, 1e-5, (B, N)) * (rng.random() < 0.6)
    nsteps = int(rng.integers(1, 12))
    spl = int(rng.choice([0, 1, 3]))
    c0 = np.repeat(cb[:, :, None], nx, axis=2) * rng.uniform(0.9, 1.1, (B, N, nx))
    p = R.Problem(D=D, charges=q, beta=BETA, eps=EPS, dx=dx, nx=nx, dt=dt, pb=pbv[0], vzeta=float(vz[0]), flux_bound=fl[0],
                  lax_friedrich=lf, use_migration=mig)
    if os.environ.get('FUZZ_ONLY') and case != int(os.environ['FUZZ_ONLY']):
        continue
    if os.environ.get('FUZZ_DUMP'):          # the case as data (to freeze an outlier into a regression test)
        np.savez_compressed(os.environ['FUZZ_DUMP'], N=N, nx=nx, B=B, method=method, lf=lf, mig=mig, D=D, q=q, dx=dx, dt=dt, pb=pbv, vz=vz,
                            fl=fl, nsteps=nsteps, spl=spl, c0=c0, beta=BETA, eps=EPS)
    if os.environ.get('FUZZ_GROWTH'):       # one case, error against the oracle step by step: rounding amplification or a discrepancy?
        for ns in (1, 2, 3, 5, 8, nsteps):
            with solver_from_problem(p, method, batch_capacity=B) as s:
                s.set_batch(c0.reshape(B, N * nx), pbv, vz, fl)
                s.step(ns, spl)
                c = s.get_state()[0]
            ref = np.ascontiguousarray(c0.copy())
            CO.steps(p, method, ref, pbv, vz, fl, ns)
            print('  steps %2d: max rel diff %.2e   (max |c| %.3e, dt %.2e, dx %.2e)' % (ns, np.abs(c - ref).max() / np.abs(ref).max(), np.abs(ref).max(), dt, dx))
    try:
        with solver_from_problem(p, method, batch_capacity=B) as s:
            s.set_batch(c0.reshape(B, N * nx), pbv, vz, fl)
            s.step(nsteps, spl)
            c, v, g, l = s.get_state()
        ref = np.ascontiguousarray(c0.copy())
        pot = CO.steps(p, method, ref, pbv, vz, fl, nsteps)
        scale = np.abs(ref).max()
        dc = np.abs(c - ref).max() / scale if np.isfinite(scale) else 0.0
        dv = np.abs(v - pot[0]).max() / max(np.abs(pot[0]).max(), 1e-30) if mig else 0.0
        ok = (dc < 1e-9 and dv < 1e-9) or not np.isfinite(ref).all()
        tag = 'ok ' if ok else 'BAD'
        if not ok:
            # the reference scheme is unstable on some random states: the solution grows by orders of magnitude within a few steps and
            # amplifies the one-step rounding difference.  That class ('ok^'): first step within 1e-10 AND the solution grew > 10 x.
            with solver_from_problem(p, method, batch_capacity=B) as s1:
                s1.set_batch(c0.reshape(B, N * nx), pbv, vz, fl)
                s1.step(1, spl)
                c1 = s1.get_state()[0]
            r1 = np.ascontiguousarray(c0.copy())
            CO.steps(p, method, r1, pbv, vz, fl, 1)
            d1 = np.abs(c1 - r1).max() / np.abs(r1).max()
            growth = scale / np.abs(c0).max()
            if d1 < 1e-10 and growth > 10.0 and dc < 1e-4:
                ok, tag = True, 'ok^'
                globals()['amplified'] = globals().get('amplified', 0) + 1
    except Exception as e:
        ok, tag, dc, dv = False, 'EXC', -1.0, -1.0
        err = str(e)[:100]
    bad += 0 if ok else 1
    if not ok or tag == 'ok^' or case % 20 == 0:
        print('%s case %3d %s N=%d nx=%d B=%d pb_mode=%d LF=%d mig=%d steps=%d spl=%d  dc=%.1e dv=%.1e %s' % (
            tag, case, method, N, nx, B, mode, lf, mig, nsteps, spl, dc, dv, err if tag == 'EXC' else ''), flush=True)
print('%d cases, %d bad, %d amplified by an unstable trajectory of the reference scheme, %.1f s' % (ncase, bad, globals().get('amplified', 0), time.time() - t0))
