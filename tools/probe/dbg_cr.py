"""dev probe: first Newton iterations of the CO2R example system, GPU vs oracle"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'examples'))
import numpy as np
import co2r_physical_sweep as E
from catint_amd import _capi
from oracle import pnp_physical as PH
for nx in (64,):
  for use_rx, use_wk, use_grid in ((2, False, False), (3, False, False), (4, False, False)):
    tp, phis = E.build(2, nx)
    names = list(tp.species.keys()); rate = E.tafel_rate(tp)
    rx = [{'lhs': [names.index(x) for x in r['reactants'][0]], 'rhs': [names.index(x) for x in r['reactants'][1]], 'kf': r['rates'][0], 'kr': r['rates'][1]} for r in list(tp.reactions.values())[use_rx - 1:use_rx]]
    if use_rx == 2:
        rx[0]['kf'] = 0.0; rx[0]['kr'] = 0.0
    if use_rx == 3:
        rx = [{'lhs': [1], 'rhs': [4], 'kf': 1.0, 'kr': 0.0}]
    if use_rx == 4:
        rx = [{'lhs': [0], 'rhs': [0], 'kf': 1.0, 'kr': 0.0}]
    cb = np.array([tp.species[s]['bulk_concentration'] for s in names])
    nu = [0.0] * 7; nu[names.index('CO2')] = -1; nu[names.index('CO')] = 1; nu[names.index('OH-')] = 2
    phiM = 0.06
    K = float(rate(np.array([phiM]))[0])
    x = tp.xmesh if use_grid else np.arange(tp.nx) * tp.dx
    radii = [tp.species[s].get('MPB_radius', 0.0) for s in names]
    print(list(tp.reactions.keys())[use_rx - 1], rx)
    for maxit in (1,):
        p = PH.PhysicalProblem(D=tp.D, charges=tp.charges, beta=tp.beta, eps=tp.eps, dx=tp.dx, nx=tp.nx, c_bulk=cb, phiM=phiM, stern_capacitance=0.2, phi_pzc=0.16,
                               mpb_radius=radii, reactions=rx, wall_kinetics=[{'species': names.index('CO2'), 'k': K, 'nu': nu}] if use_wk else [], x=x)
        c0 = np.repeat(cb[:, None], tp.nx, axis=1)
        rc, rphi, it, h = PH.newton_step(p, c0, np.zeros(tp.nx), c0, np.inf, tol=1e-9, maxit=maxit)
        s = _capi.PnpSolver(7, tp.nx, tp.dx, 1.0, tp.beta, tp.eps, tp.D, tp.charges, method='Newton', batch_capacity=1)
        s.set_newton(wall_bc='stern', stern_capacitance=0.2, phi_pzc=0.16, tol=1e-9, maxit=maxit, mpb_radius=radii)
        if use_grid: s.set_grid(x)
        if rx: s.set_reactions([(r['lhs'], r['rhs'], r['kf'], r['kr']) for r in rx])
        pb = np.zeros((1, 4)); pb[0, 0] = phiM
        s.set_batch(c0[None], pb, np.zeros(1), np.zeros((1, 7)))
        if use_wk: s.set_wall_kinetics([names.index('CO2')], [nu], [[K]])
        st = s.solve_stationary()
        c, phi, _, _ = s.get_state()
        gi = s.newton_iterations()
        s.close()
        sc = np.abs(rc).max(axis=1, keepdims=True) + 1e-30
        print('nx', tp.nx, 'rx', use_rx, 'wk', use_wk, 'grid', use_grid, 'maxit', maxit, 'its', gi, it, 'max rel diff c %.2e phi %.2e' % ((np.abs(c[0] - rc) / sc).max(), np.abs(phi[0] - rphi).max()))
