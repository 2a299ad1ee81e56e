"""dev probe: randomised BATCHES (64 ... 400 operating points) through the lane / lane-pair kernel against the lane-team kernel on the
same inputs (GPU against GPU: the oracle takes ~0.1 s per operating point): states, Newton iteration counts, status flags.  Exercises
what the small-batch fuzz (tests/fuzz/fuzz_newton.py, B <= 3) cannot: the ordering of the operating points by expected iterations,
several groups per launch, workspace chunks, lane masks, padding lanes of the last group."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from catint_amd import _capi
from catint_amd.host import graded_mesh
import tests.test_gpu_newton as T

rng = np.random.default_rng(int(os.environ.get('FUZZ_SEED', '1')))
ncase = int(os.environ.get('FUZZ_CASES', '40'))
bad = 0
t0 = time.time()


def solve(kernel, env, N, nx, B, D, q, cb, dx, phiM, kw, x, flux, stationary, dt, nsteps, mask, rx=None, velocity=0.0):
    for k in ('CATINT_NEWTON_KERNEL', 'CATINT_NEWTON_LANE_GROUPS', 'CATINT_LANE_ORDER'):
        os.environ.pop(k, None)
    os.environ['CATINT_NEWTON_KERNEL'] = kernel
    os.environ.update(env)
    c0 = np.repeat(cb[:, :, None], nx, axis=2)
    pb = np.zeros((B, 4))
    pb[:, 0] = phiM
    with _capi.PnpSolver(N, nx, dx, dt, T.BETA, T.EPS, D, q, method='Newton', batch_capacity=B) as s:
        s.set_newton(**kw)
        if x is not None:
            s.set_grid(x * dx)
        if rx:
            s.set_reactions(rx)
        if velocity:
            s.set_convection(velocity)
        s.set_batch(c0, pb, np.zeros(B), flux)
        if stationary:
            s.solve_stationary()
            it1 = s.newton_iterations()
            pb[:, 0] *= 1.1
            s.set_pb(pb, np.zeros(B))
            if mask is not None:
                s.set_lane_mask(mask)
            s.solve_stationary()
        else:
            s.step(nsteps)
            it1 = s.newton_iterations()
            if mask is not None:
                s.set_lane_mask(mask)
            s.step(2)
        c, phi, _, _ = s.get_state()
        return c, phi, it1, s.newton_iterations(), s.get_status()


for case in range(ncase):
    kernel = str(rng.choice(['lane4', 'lane2', 'lane'], p=[0.4, 0.25, 0.35]))
    N = int(rng.integers(5, 9)) if kernel != 'lane' else int(rng.integers(2, 9))
    rx, velocity = None, 0.0            # (round 4: homogeneous reactions and the convection term run on the lane kernels)
    if rng.random() < 0.35:
        rx = []
        for _ in range(int(rng.integers(1, 4))):
            nl, nr = int(rng.integers(0, 3)), int(rng.integers(1, 3))
            rx.append(([int(k) for k in rng.integers(0, N, nl)], [int(k) for k in rng.integers(0, N, nr)], float(10 ** rng.uniform(0, 4)),
                       float(10 ** rng.uniform(0, 4))))
    nx = int(rng.choice([9, 17, 33, 64, 65, 100, 129, 200, 257]))
    B = int(rng.integers(64, 401))
    kw = {'tol': 1e-10, 'maxit': 60}
    if rng.random() < 0.5:
        kw.update(wall_bc='stern', stern_capacitance=float(rng.uniform(0.05, 0.4)), phi_pzc=float(rng.uniform(-0.05, 0.05)))
    if rng.random() < 0.5:
        kw['mpb_radius'] = [float(a) for a in rng.uniform(0, 4.2e-10, N) * (rng.random(N) < 0.7)]
    D, q, cb, dx, phiM = T.make_lanes(N, nx, B, 5000 + case, phi_lo=-0.25, phi_hi=0.25)
    x = None
    if rng.random() < 0.4:
        x = np.cumsum(np.concatenate([[0.0], np.geomspace(0.4, 2.5, nx - 1)]))
    flux = np.zeros((B, N))
    if rng.random() < 0.4:
        flux[:, int(rng.integers(0, N))] = rng.uniform(-2e-5, 2e-5, B)
    stationary = rng.random() < 0.5
    dt = float(0.2 * (6 * dx) * (nx * dx) / D.max() * 10 ** rng.uniform(-1, 1))
    nsteps = int(rng.integers(1, 5))
    mask = (rng.random(B) < 0.7).astype(np.int32) if rng.random() < 0.4 else None
    env = {}
    if rng.random() < 0.4:
        env['CATINT_NEWTON_LANE_GROUPS'] = str(int(rng.integers(1, 4)))
    if rng.random() < 0.2:
        env['CATINT_LANE_ORDER'] = '0'
    try:
        if rng.random() < 0.2:
            velocity = float(rng.choice([-2.0, 3.0]) * D.max() / ((nx - 1) * dx))
        got = solve(kernel, env, N, nx, B, D, q, cb, dx, phiM, kw, x, flux, stationary, dt, nsteps, mask, rx, velocity)
        ref = solve('team', {}, N, nx, B, D, q, cb, dx, phiM, kw, x, flux, stationary, dt, nsteps, mask, rx, velocity)
        good = (ref[4] == 0) & (got[4] == 0)
        sc = np.abs(ref[0]).max(axis=2, keepdims=True)
        dc = float((np.abs(got[0] - ref[0]) / (sc + 1e-300))[good].max()) if good.any() else 0.0
        dp = float(np.abs(got[1] - ref[1])[good].max()) if good.any() else 0.0
        same_it = np.array_equal(got[2], ref[2]) and np.array_equal(got[3][good], ref[3][good])
        near = (np.abs(got[2].astype(int) - ref[2]).max() <= 1) and (np.abs(got[3].astype(int) - ref[3])[good].max() <= 1 if good.any() else True)
        ok = np.array_equal(got[4], ref[4]) and dc < 1e-8 and dp < 1e-9 and (same_it or near)
        tag = 'ok ' if ok and same_it else ('ok~' if ok else 'BAD')
        if not ok and os.environ.get('FUZZ_VERBOSE'):
            d1 = np.flatnonzero(got[2] != ref[2]); d2 = np.flatnonzero(got[3] != ref[3]); d3 = np.flatnonzero(got[4] != ref[4])
            print('   first solve: lanes', d1[:8], 'got', got[2][d1[:8]], 'ref', ref[2][d1[:8]], '| second:', d2[:8], got[3][d2[:8]], ref[3][d2[:8]],
                  '| status:', d3[:8], got[4][d3[:8]], ref[4][d3[:8]], '| mask', None if mask is None else mask[d1[:8]])
    except Exception as e:
        ok, tag, dc, dp, same_it = False, 'EXC', -1, -1, str(e)[:100]
    bad += 0 if ok else 1
    if not ok or tag != 'ok ' or case % 10 == 0:
        print('%s case %2d %s rx=%d v=%d N=%d nx=%d B=%d stern=%d mpb=%d grid=%d flux=%d stat=%d mask=%d env=%s  dc=%.1e dphi=%.1e same_iterations=%s not_converged=%d' % (
            tag, case, kernel, len(rx) if rx else 0, velocity != 0.0, N, nx, B, 'wall_bc' in kw, 'mpb_radius' in kw, x is not None, bool(flux.any()), stationary, mask is not None, env, dc, dp, same_it,
            int((ref[4] != 0).sum()) if tag != 'EXC' else -1), flush=True)
print('%d cases, %d bad, %.1f s' % (ncase, bad, time.time() - t0))
