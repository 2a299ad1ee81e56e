"""GPU: every BASELINE.json configuration at ITS batch (one GPU's share where the config spans 8 GPUs), through the kernel the
library selects at that batch -- VERDICT r01 "configs_untested".

  configs[2]  4096-lane CO2R voltage sweep (examples/co2r_physical_sweep.py, the reference's run.py system)
  configs[3]  32768 lanes x 6 species x 1024 points: compat mode (C oracle subsample) and >= 100 implicit timesteps
  configs[4]  8192 lanes x 8 species (size-modified, Stern wall) x 4096 points: physical mode through the sweep kernel
              (sampled lanes vs oracle/pnp_physical.py incl. Newton iteration counts, lane-permutation property) and compat mode

Sampled lanes are compared with the oracles (rtol 1e-9 compat, 2e-9 physical, as everywhere); whole-batch properties
(status, finiteness, lane independence) cover the rest."""
import os

import numpy as np
import pytest

from catint_amd import _capi
from catint_amd.synthetic import make_batch
from catint_amd.host import solver_from_problem
from oracle import c_oracle as CO
from oracle import pnp_physical as PH

pytestmark = pytest.mark.gpu
RTOL = 1e-9
RADII8 = [4.1e-10, 3.6e-10, 3.3e-10, 3e-10, 3e-10, 3e-10, 4.5e-10, 3.5e-10]


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


# ---- configs[4]: 65536 lanes over 8 GPUs -> 8192 lanes per GPU, 8 species, 4096 points ----------------------------------------
def test_config5_share_physical_mode_sweep_kernel_8192_lanes():
    B, N, nx = 8192, 8, 4096
    prob, c0, pb, vz, fl = make_batch(B, N, nx, seed=8, phi_max=0.2, dt_factor=0.1)
    pb = np.nan_to_num(pb)
    kw = dict(wall_bc='stern', stern_capacitance=0.2, tol=1e-9, mpb_radius=RADII8)

    def solve(idx, steps):
        with _capi.PnpSolver(N, nx, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Newton',
                             batch_capacity=len(idx)) as s:
            s.set_newton(**kw)
            s.set_batch(c0[idx], pb[idx], vz[idx], fl[idx])
            s.step(steps)
            cs, vs, es = s.get_surface()
            full = s.get_state() if len(idx) <= 64 else None
            return cs, vs, s.newton_iterations(), s.get_status(), full

    cs, vs, its, st, _ = solve(np.arange(B), 2)                     # the library's choice at this batch: the lane kernel (pnp_lane.hip)
    assert np.all(st == 0) and np.all(np.isfinite(cs)) and cs.min() > 0 and its.min() >= 4 and its.max() <= 2 * 50
    # lane permutation: teams pick the lanes up in another order, results must follow the lanes
    perm = np.random.default_rng(3).permutation(B)
    cs2, vs2, its2, st2, _ = solve(perm, 2)
    assert np.array_equal(its2, its[perm]) and np.array_equal(cs2, cs[perm]) and np.array_equal(vs2, vs[perm])
    # sampled lanes against the oracle: states, potentials and Newton iteration counts
    sub = np.array([0, 1234, 5000, 8191])
    for b in sub:
        p = PH.PhysicalProblem(D=prob.D, charges=prob.charges, beta=prob.beta, eps=prob.eps, dx=prob.dx, nx=nx,
                               c_bulk=c0[b].reshape(N, nx)[:, -1], phiM=pb[b, 0], stern_capacitance=0.2, mpb_radius=RADII8)
        rc, rphi, rit = PH.integrate(p, c0[b].reshape(N, nx), np.zeros(nx), prob.dt, 2, tol=1e-9)
        assert sum(rit) == its[b], (b, rit, its[b])
        assert np.abs(cs[b] - rc[:, 0]).max() <= 2e-9 * np.abs(rc).max()
        assert abs(vs[b] - rphi[0]) <= 2e-9 * 0.2
    # the same lanes solved as a small batch (lane-team kernel): whole profiles against the oracle, surface values against the sweep
    cs3, vs3, its3, st3, full = solve(sub, 2)
    assert np.array_equal(its3, its[sub]) and np.abs(cs3 - cs[sub]).max() <= 1e-9 * np.abs(cs).max()
    b = sub[1]
    p = PH.PhysicalProblem(D=prob.D, charges=prob.charges, beta=prob.beta, eps=prob.eps, dx=prob.dx, nx=nx,
                           c_bulk=c0[b].reshape(N, nx)[:, -1], phiM=pb[b, 0], stern_capacitance=0.2, mpb_radius=RADII8)
    rc, rphi, rit = PH.integrate(p, c0[b].reshape(N, nx), np.zeros(nx), prob.dt, 2, tol=1e-9)
    assert np.abs(full[0][1] - rc).max() <= 2e-9 * np.abs(rc).max() and np.abs(full[1][1] - rphi).max() <= 2e-9 * 0.2


def test_half_size_batch_of_large_blocks_lane_kernel_and_both_sweeps():
    """4096 lanes x 8 size-modified species x 512 points: the library's choice is the lane kernel (one operating point per lane) --
    lane permutation property, sampled lanes against the oracle incl. iteration counts, and the same lanes through the one-sided and
    the two-sided sweep kernels (lane teams; 585 / 1366 waves)."""
    B, N, nx = 4096, 8, 512
    prob, c0, pb, vz, fl = make_batch(B, N, nx, seed=18, phi_max=0.2, dt_factor=0.1)
    pb = np.nan_to_num(pb)
    kw = dict(wall_bc='stern', stern_capacitance=0.2, tol=1e-9, mpb_radius=RADII8)

    def solve(idx, steps):
        with _capi.PnpSolver(N, nx, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Newton',
                             batch_capacity=len(idx)) as s:
            s.set_newton(**kw)
            s.set_batch(c0[idx], pb[idx], vz[idx], fl[idx])
            s.step(steps)
            cs, vs, es = s.get_surface()
            return cs, vs, s.newton_iterations(), s.get_status()

    cs, vs, its, st = solve(np.arange(B), 3)
    assert np.all(st == 0) and cs.min() > 0 and its.min() >= 6
    perm = np.random.default_rng(4).permutation(B)
    cs2, vs2, its2, st2 = solve(perm, 3)
    assert np.array_equal(its2, its[perm]) and np.array_equal(cs2, cs[perm]) and np.array_equal(vs2, vs[perm])
    for b in (0, 777, 4095):
        p = PH.PhysicalProblem(D=prob.D, charges=prob.charges, beta=prob.beta, eps=prob.eps, dx=prob.dx, nx=nx,
                               c_bulk=c0[b].reshape(N, nx)[:, -1], phiM=pb[b, 0], stern_capacitance=0.2, mpb_radius=RADII8)
        rc, rphi, rit = PH.integrate(p, c0[b].reshape(N, nx), np.zeros(nx), prob.dt, 3, tol=1e-9)
        assert sum(rit) == its[b], (b, rit, its[b])
        assert np.abs(cs[b] - rc[:, 0]).max() <= 2e-9 * np.abs(rc).max() and abs(vs[b] - rphi[0]) <= 2e-9 * 0.2
    for kern in ('sweep', 'both'):
        os.environ['CATINT_NEWTON_KERNEL'] = kern
        try:
            cs1, vs1, its1, st1 = solve(np.arange(B), 3)
        finally:
            del os.environ['CATINT_NEWTON_KERNEL']
        assert np.array_equal(its1, its) and np.abs(cs1 - cs).max() <= 1e-10 * np.abs(cs).max() and not np.array_equal(cs1, cs)


def test_config5_share_compat_mode_8192_lanes():
    B, N, nx = 8192, 8, 4096
    p, c0, pb, vz, fl = make_batch(B, N, nx, seed=9, phi_max=0.02, dt_factor=1e-4)
    nsteps = 3
    with solver_from_problem(p, 'Crank-Nicolson', batch_capacity=B) as s:
        s.set_batch(c0, pb, vz, fl)
        s.step(nsteps)
        st = s.get_status()
        c, v, g, l = s.get_state()
    assert (st == 0).all() and np.isfinite(c).all()
    assert np.array_equal(c[:, :, -1], c0.reshape(B, N, nx)[:, :, -1])          # Dirichlet bulk value held (calculator_old.py:540)
    sub = np.array([0, 999, 4096, 8191])
    ref = np.ascontiguousarray(c0[sub].reshape(len(sub), N, nx).copy())
    rv, rg, rl = CO.steps(p, 'Crank-Nicolson', ref, pb[sub], vz[sub], fl[sub], nsteps)
    assert relerr(c[sub], ref) < RTOL and relerr(v[sub], rv) < RTOL and relerr(g[sub], rg) < RTOL and relerr(l[sub], rl) < RTOL


# ---- configs[3]: 262144 lanes over 8 GPUs -> 32768 lanes per GPU, 6 species, 1024 points ----------------------------------------
def test_config4_share_compat_mode_32768_lanes():
    B, N, nx = 32768, 6, 1024
    p, c0, pb, vz, fl = make_batch(B, N, nx, seed=10, phi_max=0.02, dt_factor=1e-4)
    nsteps = 6
    with solver_from_problem(p, 'Crank-Nicolson', batch_capacity=B) as s:
        s.set_batch(c0, pb, vz, fl)
        s.step(nsteps, 1)                          # one launch per timestep: state through HBM every step (2.1 GB of state)
        st = s.get_status()
        c, v, g, l = s.get_state()
        s.set_batch(c0, pb, vz, fl)
        s.step(nsteps, 0)                          # fused launches: same bits
        c2 = s.get_state(potential=False)
    assert (st == 0).all() and np.isfinite(c).all() and relerr(c, c2) < 1e-12      # (two kernel families: equal to rounding)
    sub = np.array([0, 1, 4097, 20000, 32767])
    ref = np.ascontiguousarray(c0[sub].reshape(len(sub), N, nx).copy())
    rv, rg, rl = CO.steps(p, 'Crank-Nicolson', ref, pb[sub], vz[sub], fl[sub], nsteps)
    assert relerr(c[sub], ref) < RTOL and relerr(v[sub], rv) < RTOL and relerr(g[sub], rg) < RTOL and relerr(l[sub], rl) < RTOL


def test_config4_share_physical_mode_100_transient_steps():
    """100 of configs[3]'s 1000 backward-Euler steps on one GPU's 32768 lanes (size-modified, Stern wall): every lane converges in
    every step; sampled lanes follow the oracle over the whole trajectory, Newton iteration totals included."""
    B, N, nx, nsteps = 32768, 6, 1024, 100
    prob, c0, pb, vz, fl = make_batch(B, N, nx, seed=12, phi_max=0.2, dt_factor=0.1)
    pb = np.nan_to_num(pb)
    radii = [4.1e-10, 3.6e-10, 3.3e-10, 3e-10, 0.0, 3e-10]
    with _capi.PnpSolver(N, nx, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Newton', batch_capacity=B) as s:
        s.set_newton(wall_bc='stern', stern_capacitance=0.2, tol=1e-9, mpb_radius=radii)
        s.set_batch(c0, pb, vz, fl)
        its = np.zeros(B, np.int64)
        for _ in range(4):                         # iteration counters are per call: 4 calls of 25 steps
            s.step(nsteps // 4)
            its += s.newton_iterations()
        st = s.get_status()
        cs, vs, es = s.get_surface()
    assert np.all(st == 0) and np.all(np.isfinite(cs)) and cs.min() > 0
    assert its.min() >= nsteps and its.max() <= 50 * nsteps
    for b in (5, 31000):
        p = PH.PhysicalProblem(D=prob.D, charges=prob.charges, beta=prob.beta, eps=prob.eps, dx=prob.dx, nx=nx,
                               c_bulk=c0[b].reshape(N, nx)[:, -1], phiM=pb[b, 0], stern_capacitance=0.2, mpb_radius=radii)
        rc, rphi, rit = PH.integrate(p, c0[b].reshape(N, nx), np.zeros(nx), prob.dt, nsteps, tol=1e-9)
        assert sum(rit) == its[b], (b, sum(rit), its[b])
        assert np.abs(cs[b] - rc[:, 0]).max() <= 2e-9 * np.abs(rc).max() and abs(vs[b] - rphi[0]) <= 2e-9 * 0.2


# ---- configs[2]: the 4096-point polarization sweep of examples/02_CO2R_Au_CatMAP -----------------------------------------------
def test_config3_co2r_sweep_4096_lanes_against_the_oracle():
    import importlib.util
    from catint_amd.calculator import Calculator
    spec = importlib.util.spec_from_file_location('co2r_physical_sweep', os.path.join(os.path.dirname(__file__), '..', 'examples',
                                                                                       'co2r_physical_sweep.py'))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    lanes = 4096
    tp, phis = ex.build(lanes, 384)
    names = list(tp.species.keys())
    rate = ex.tafel_rate(tp)
    calc = Calculator(transport=tp, calc='comsol')
    tp.newton = {'tol': 1e-9, 'maxit': 80}
    calc.set_surface_kinetics([{'species': 'CO2', 'rate': rate, 'stoichiometry': {'CO2': -1.0, 'CO': 1.0, 'OH-': 2.0}}])
    calc.run()
    assert np.all(calc.status == 0) and calc.continuation_stages >= 5
    rx = [{'lhs': [names.index(x) for x in r['reactants'][0]], 'rhs': [names.index(x) for x in r['reactants'][1]],
           'kf': r['rates'][0], 'kr': r['rates'][1]} for r in tp.reactions.values()]
    cb = np.array([tp.species[s]['bulk_concentration'] for s in names])
    nu = [0.0] * len(names)
    nu[names.index('CO2')], nu[names.index('CO')], nu[names.index('OH-')] = -1.0, 1.0, 2.0
    for lane in (0, 1365, 2730, 4095):
        c = np.repeat(cb[:, None], tp.nx, axis=1)
        phi = np.zeros(tp.nx)
        for w in np.arange(1, calc.continuation_stages + 1) / float(calc.continuation_stages):
            pm = 0.16 + (phis[lane] - 0.16) * w
            p = PH.PhysicalProblem(D=tp.D, charges=tp.charges, beta=tp.beta, eps=tp.eps, dx=tp.dx, nx=tp.nx, c_bulk=cb, phiM=pm,
                                   stern_capacitance=0.2, phi_pzc=0.16, mpb_radius=[tp.species[s].get('MPB_radius', 0.0) for s in names],
                                   reactions=rx, x=tp.xmesh,
                                   wall_kinetics=[{'species': names.index('CO2'), 'k': float(rate(np.array([pm]))[0]), 'nu': nu}])
            c, phi, it, _ = PH.newton_step(p, c, phi, c, np.inf, tol=1e-9, maxit=80)
            assert it <= 80
        d = tp.alldata[lane]
        got = np.array([d['species'][sp]['concentration'] for sp in names])
        scale = np.abs(c).max(axis=1, keepdims=True)
        assert (np.abs(got - c) / scale).max() < 1e-6, lane
        assert np.abs(np.array(d['system']['potential']) - phi).max() < 1e-7, lane
    j = calc.kinetic_flux[:, names.index('CO')]
    assert j[0] > 0 and (np.diff(j[:2000]) > 0).all()          # Tafel rise along the first half of the sweep
    assert j.max() < tp.D[names.index('CO2')] * cb[names.index('CO2')] / tp.xmesh[-1]      # below the pure-diffusion limit


# ---- configs[2]: the SCF outer loop of the 4096-point sweep (kinetics <-> transport, calculator.py:294-406) on the device --------------
def test_config3_scf_outer_loop_4096_lanes_device_loop_equals_host_loop():
    """BASELINE configs[2] is a 4096-point polarization sweep WITH the SCF outer loop.  CatMAP is not available; the kinetic model is
    the Tafel law of tests/test_gpu_calculator.py (first order in the surface CO2 concentration), on which explicit mixing with the
    reference's mix_scf = 0.02 (run.py:95) converges: 4096 voltages from -0.6 to -1.4 V as one batch through pnp_scf_cycle (mixing,
    fallback, accuracy, mix decay per lane on the device, transport solves masked to the lanes still iterating), against (a) the
    batched HOST loop -- itself pinned to the reference's run_scf_cycle, tests/test_host_scf.py -- driving the same GPU transport
    solves with the law as a Python callback, and (b) the analytic fixed point of a neutral species."""
    import collections
    from catint_amd.calculator import Calculator
    from catint_amd.transport import Transport
    B = 4096
    phis = list(np.linspace(-0.6, -1.4, B))
    rate = lambda phiM: 1e-4 * np.exp(-12.0 * (phiM + 0.6))       # noqa: E731
    kin = [{'species': 'CO2', 'rate': rate, 'stoichiometry': {'CO2': -1.0, 'CO': 1.0}}]

    def make_tp():
        species = collections.OrderedDict([('K+', {'bulk_concentration': 100.0}), ('HCO3-', {'bulk_concentration': 100.0}),
                                           ('CO2', {'bulk_concentration': 34.0}), ('CO', {'bulk_concentration': 0.0})])
        return Transport(species=species, system={'phiM': phis[0], 'boundary thickness': 4e-8, 'Stern capacitance': 20.0, 'phiPZC': 0.1},
                         nx=256, descriptors={'phiM': phis})

    tp = make_tp()
    calc = Calculator(transport=tp, calc='comsol', tau_scf=1e-6, mix_scf=0.02)
    calc.set_surface_kinetics(kin)
    dev = calc.run_scf_cycle(nel=[1, 1, 2, 2], max_iter=3000)
    assert dev['converged'].all() and not dev['failed'].any() and dev['iterations'] > 100
    L = (tp.nx - 1) * tp.dx
    kap = rate(np.array(phis)) * L / tp.D[2]
    assert np.allclose(dev['surface_concentration'][:, 2], 34.0 / (1.0 + kap), rtol=2e-4)
    assert (np.diff(-dev['flux'][:, 2]) > 0).all()                                  # the Tafel rise into the transport plateau
    tp2 = make_tp()
    host_calc = Calculator(transport=tp2, calc='comsol', tau_scf=1e-6, mix_scf=0.02)

    def cb(state):
        f = np.zeros((B, 4))
        f[:, 2] = -rate(state['phiM']) * np.maximum(state['surface_concentration'][:, 2], 0.0)
        f[:, 3] = -f[:, 2]
        return f
    host = host_calc.run_scf_cycle(cb, nel=[1, 1, 2, 2], max_iter=3000)
    assert host['converged'].all() and dev['iterations'] == host['iterations']
    assert np.array_equal(dev['mix'], host['mix'])
    assert np.allclose(dev['surface_concentration'], host['surface_concentration'], rtol=1e-9, atol=1e-12)
    assert np.allclose(dev['flux'], host['flux'], rtol=1e-9, atol=1e-18)
