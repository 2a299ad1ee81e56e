"""Known-answer tests that pin the physical-mode CPU oracle (oracle/pnp_physical.py).

The reference delegates this solve to COMSOL (catint/comsol_wrapper.py:145,158) and holds no fixtures for it, so parity
with the reference is UNPINNED; what the reference does carry are analytic answers, used here:
Gouy-Chapman (catint/transport.py:1373-1383), Boltzmann profiles (:1325-1346), the Debye length (:439-443).
"""
import numpy as np
import pytest

from oracle import pnp_physical as PH

F = 96485.33289
RGAS = 8.3144598
T = 298.14
EPS = 78.36 * 8.854187817e-12
BETA = 1.0 / (RGAS * T)


def debye(cb, q):
    return np.sqrt(EPS / BETA / (q ** 2 * cb).sum())


def binary(phiM, ppl=8, ncell=30, cb=10.0, **kw):
    cbv = np.array([cb, cb])
    q = np.array([F, -F])
    lam = debye(cbv, q)
    nx = ncell * ppl + 1
    dx = ncell * lam / (nx - 1)
    p = PH.PhysicalProblem(D=[1.957e-9, 1.185e-9], charges=q, beta=BETA, eps=EPS, dx=dx, nx=nx, c_bulk=cbv, phiM=phiM, **kw)
    c0 = np.repeat(cbv[:, None], nx, axis=1)
    return p, c0, np.zeros(nx), lam


def test_bernoulli_series_matches_closed_form():
    u = np.array([-0.0499999, -1e-3, 0.0, 1e-9, 0.04999, 0.0500001, -0.0500001, 3.0, -40.0, 800.0])
    B, dB = PH.bernoulli(u)
    import mpmath as mp
    mp.mp.dps = 60
    for ui, Bi, dBi in zip(u, B, dB):
        x = mp.mpf(float(ui))
        Bx = mp.mpf(1) if x == 0 else x / mp.expm1(x)
        dBx = mp.mpf(-0.5) if x == 0 else (mp.expm1(x) - x * mp.exp(x)) / mp.expm1(x) ** 2
        assert abs(Bi - float(Bx)) <= 4e-16 * max(1.0, abs(float(Bx)))
        assert abs(dBi - float(dBx)) <= 2e-14 * max(1.0, abs(float(dBx)))


def test_jacobian_matches_finite_differences():
    rng = np.random.default_rng(0)
    nx, nb = 12, 4
    p = PH.PhysicalProblem(D=[1.957e-9, 1.185e-9, 1e-9], charges=[F, -F, 0.0], beta=BETA, eps=EPS, dx=3e-10, nx=nx,
                           c_bulk=[10, 10, 5], phiM=0.1, flux=[1e-4, 0, -1e-4], stern_capacitance=0.2,
                           mpb_radius=[4e-10, 3e-10, 0],
                           reactions=[{'lhs': [0, 1], 'rhs': [2], 'kf': 3e3, 'kr': 2e4},
                                      {'lhs': [2, 2], 'rhs': [0], 'kf': 1e2, 'kr': 5e2}])
    c = rng.uniform(1, 20, (3, nx)); phi = rng.uniform(-0.05, 0.1, nx); co = rng.uniform(1, 20, (3, nx))
    dt = 1e-9
    _, L, M, U = PH.residual_and_jacobian(p, c, phi, co, dt)
    J = np.zeros((nx * nb, nx * nb))
    for i in range(nx):
        for blk, o in ((L, -1), (M, 0), (U, 1)):
            if 0 <= i + o < nx:
                J[i * nb:(i + 1) * nb, (i + o) * nb:(i + o + 1) * nb] = blk[i]
    x0 = np.concatenate([c, phi[None]], 0)
    Jfd = np.zeros_like(J)
    for j in range(nx):
        for s in range(nb):
            h = 1e-5 * max(abs(x0[s, j]), 1e-3)
            xp = x0.copy(); xp[s, j] += h
            xm = x0.copy(); xm[s, j] -= h
            Jfd[:, j * nb + s] = ((PH.residual(p, xp[:3], xp[3], co, dt) - PH.residual(p, xm[:3], xm[3], co, dt)) / (2 * h)).T.reshape(-1)
    assert np.abs(J - Jfd).max() <= 2e-6 * np.abs(Jfd).max()
    col = np.abs(Jfd).max(axis=0)
    assert (np.abs(J - Jfd) / col[None, :]).max() < 1e-5


def test_jacobian_of_butler_volmer_and_langmuir_wall_kinetics_matches_finite_differences():
    """Wall rate law K c/(1 + K_sat c) exp(alpha (phiM - phi_0)) (docs/source/topics/flux_definition.rst:90-160 of the reference):
    concentration AND surface-potential derivatives of the wall rows."""
    rng = np.random.default_rng(5)
    nx, nb = 8, 4
    wk = [{'species': 2, 'k': 3e-2, 'nu': [0.0, 1.0, -1.0], 'alpha': -19.5, 'saturation': 0.08},
          {'species': 0, 'k': 2e-3, 'nu': [-1.0, 0.0, 1.0], 'alpha': 12.0},
          {'species': -1, 'k': 4e-6, 'nu': [0.0, 1.0, 0.0], 'alpha': -7.0},
          {'species': 1, 'k': 5e-3, 'nu': [0.0, -1.0, 0.0], 'saturation': 0.3}]
    p = PH.PhysicalProblem(D=[1.957e-9, 1.185e-9, 1e-9], charges=[F, -F, 0.0], beta=BETA, eps=EPS, dx=3e-10, nx=nx,
                           c_bulk=[10, 10, 5], phiM=-0.4, stern_capacitance=0.2, phi_pzc=0.05, wall_kinetics=wk)
    c = rng.uniform(1, 20, (3, nx)); phi = rng.uniform(-0.15, 0.0, nx); co = c.copy()
    F0, L, M, U = PH.residual_and_jacobian(p, c, phi, co, np.inf)
    x0 = np.concatenate([c, phi[None]], 0)
    for s_ in range(nb):                       # only the wall point's unknowns enter the wall kinetics
        h = 1e-6 * max(abs(x0[s_, 0]), 1e-3)
        xp = x0.copy(); xp[s_, 0] += h
        xm = x0.copy(); xm[s_, 0] -= h
        fd = (PH.residual(p, xp[:3], xp[3], co, np.inf) - PH.residual(p, xm[:3], xm[3], co, np.inf)) / (2 * h)
        assert np.abs(M[0][:, s_] - fd[:, 0]).max() <= 2e-6 * np.abs(fd[:, 0]).max()
    # the potential column is not a rounding-level effect: the BV terms are a sizeable part of it
    p0 = PH.PhysicalProblem(D=p.D, charges=p.q, beta=BETA, eps=EPS, dx=3e-10, nx=nx, c_bulk=[10, 10, 5], phiM=-0.4,
                            stern_capacitance=0.2, phi_pzc=0.05, wall_kinetics=[dict(w, alpha=0.0) for w in wk])
    M0 = PH.residual_and_jacobian(p0, c, phi, co, np.inf)[2]
    assert np.abs(M[0][:3, 3] - M0[0][:3, 3]).max() > 1e-3 * np.abs(M[0][:3, 3]).max()
    # alpha = saturation = 0 is the first-order table, bit for bit
    first = [{k: v for k, v in w.items() if k in ('species', 'k', 'nu')} for w in wk]
    zero = [dict(w, alpha=0.0, saturation=0.0) for w in first]
    pa = PH.PhysicalProblem(D=p.D, charges=p.q, beta=BETA, eps=EPS, dx=3e-10, nx=nx, c_bulk=[10, 10, 5], phiM=-0.4, wall_kinetics=first)
    pz = PH.PhysicalProblem(D=p.D, charges=p.q, beta=BETA, eps=EPS, dx=3e-10, nx=nx, c_bulk=[10, 10, 5], phiM=-0.4, wall_kinetics=zero)
    ra, rz = PH.residual_and_jacobian(pa, c, phi, co, np.inf), PH.residual_and_jacobian(pz, c, phi, co, np.inf)
    assert all(np.array_equal(a, b) for a, b in zip(ra, rz))


def test_block_pcr_mirror_equals_banded_lu():
    rng = np.random.default_rng(1)
    p, c0, phi0, lam = binary(0.1, ppl=4, ncell=16)
    c = c0 * rng.uniform(0.5, 2.0, c0.shape); phi = rng.uniform(-0.1, 0.1, p.nx)
    F_, L, M, U = PH.residual_and_jacobian(p, c, phi, c0, 1e-8)
    x1 = PH.solve_block_tridiagonal(L, M, U, -F_)
    x2 = PH.solve_block_pcr(L, M, U, -F_)
    assert np.abs(x1 - x2).max() <= 1e-11 * np.abs(x1).max()


@pytest.mark.parametrize("phiM", [0.025, -0.1, 0.2])
def test_stationary_is_boltzmann_and_gouy_chapman(phiM):
    p, c0, phi0, lam = binary(phiM, ppl=16 if abs(phiM) > 0.1 else 8)
    c, phi, it, hist = PH.newton_step(p, c0, np.linspace(phiM, 0, p.nx), c0, np.inf, tol=1e-11, maxit=60)
    assert it <= 12
    # exponentially fitted fluxes: Boltzmann to rounding on any grid (transport.py:1325-1346)
    for k in range(2):
        cb = p.c_bulk[k] * np.exp(-BETA * p.q[k] * phi)
        assert (np.abs(c[k] - cb) / cb).max() < 1e-11
    gc = PH.gouy_chapman(np.arange(p.nx) * p.dx, phiM, BETA, F, lam)
    tol = {0.025: 5e-4, -0.1: 4e-3, 0.2: 2e-2}[phiM]
    assert np.abs(phi - gc).max() / abs(phiM) < tol


def test_second_order_convergence_to_gouy_chapman():
    errs = []
    for ppl in (4, 8, 16):
        p, c0, phi0, lam = binary(0.05, ppl=ppl)
        c, phi, it, _ = PH.newton_step(p, c0, phi0, c0, np.inf, tol=1e-12)
        errs.append(np.abs(phi - PH.gouy_chapman(np.arange(p.nx) * p.dx, 0.05, BETA, F, lam)).max())
    assert errs[0] / errs[1] > 3.5 and errs[1] / errs[2] > 3.5


def test_newton_converges_quadratically():
    p, c0, phi0, lam = binary(0.03)
    _, _, it, hist = PH.newton_step(p, c0, phi0, c0, np.inf, tol=1e-13, dphi_max=None)
    h = [x for x in hist if x < 1e-1]
    assert it <= 7 and any(h[i + 1] < 50 * h[i] ** 2 for i in range(len(h) - 1)) and hist[-1] < 1e-13


def test_transient_conserves_mass_and_reaches_the_stationary_state():
    p, c0, phi0, lam = binary(0.05, ppl=4, ncell=12)
    dt = 0.2 * lam * (p.nx - 1) * p.dx / p.D.max()
    c, phi = c0.copy(), phi0.copy()
    its = []
    for n in range(120):
        step = dt if n < 60 else 50 * dt
        cn, phi, it, _ = PH.newton_step(p, c, phi, c, step, tol=1e-12)
        # conservative form: the half-cell-weighted mass changes only by what crosses the last edge
        for k in range(2):
            mass = lambda a: p.dx * (0.5 * a[k, 0] + a[k, 1:-1].sum())
            u = p.q[k] * p.beta * (phi[-1] - phi[-2])
            B, _ = PH.bernoulli(np.array([u]))
            jhat = -((B[0] + u) * cn[k, -1] - B[0] * cn[k, -2])
            assert abs((mass(cn) - mass(c)) + jhat * p.D[k] * step / p.dx) < 1e-9 * mass(c)
        c = cn
        its.append(it)
    cs, phis, _, _ = PH.newton_step(p, c0, phi0, c0, np.inf, tol=1e-12)
    assert max(its) <= 8 and its[-1] <= 3
    assert np.abs(phi - phis).max() < 1e-6 * abs(p.phiM)
    assert (np.abs(c - cs) / cs).max() < 1e-5


def test_wall_flux_balance_at_steady_state():
    # neutral species with a wall flux: stationary profile is linear with slope -j/D (flux closure, transport.py:1034-1095)
    nx = 41
    p = PH.PhysicalProblem(D=[2e-9, 1e-9], charges=[0.0, 0.0], beta=BETA, eps=EPS, dx=1e-6, nx=nx, c_bulk=[5.0, 1.0],
                           phiM=0.0, flux=[-1e-4, 1e-4])
    c0 = np.repeat(p.c_bulk[:, None], nx, axis=1)
    c, phi, it, _ = PH.newton_step(p, c0, np.zeros(nx), c0, np.inf)
    x = np.arange(nx) * p.dx
    for k in range(2):
        expect = p.c_bulk[k] + p.flux[k] / p.D[k] * (x[-1] - x)
        assert np.abs(c[k] - expect).max() < 1e-10 * expect.max()
    assert np.abs(phi).max() < 1e-12


def test_stern_layer_and_steric_saturation():
    a = 4.1e-10
    cmax = 1.0 / (PH.N_AVOGADRO * a ** 3)
    p, c0, phi0, lam = binary(-2.0, cb=100.0, stern_capacitance=0.2, mpb_radius=[a, 3.1e-10])
    c, phi, it, _ = PH.newton_step(p, c0, phi0, c0, np.inf, tol=1e-10, maxit=60)
    assert it <= 15
    assert 0.8 * cmax < c[0, 0] < cmax                     # cations crowd at the wall but never exceed 1/(N_A a^3)
    # Stern Robin condition: eps*E(0) = C_S (phiM - phi(0)) with E from the one-sided difference used in the residual
    lhs = p.eps * (phi[1] - phi[0]) / p.dx
    assert abs(lhs + p.CS * (p.phiM - phi[0])) < 1e-9 * abs(lhs)
    # MPB equilibrium: c_k (1-phi0)^-1 exp(q beta phi) is constant (Bikerman), transport.py:1325-1346 generalised
    free = 1.0 - (p.vol[:, None] * c).sum(axis=0)
    for k in range(2):
        inv = c[k] / free * np.exp(BETA * p.q[k] * phi)
        assert (np.abs(inv - inv[-1]) / inv[-1]).max() < 1e-9


def test_homogeneous_reaction_reaches_mass_action_equilibrium():
    # A + B <-> C, neutral, no fluxes, bulk not in equilibrium -> interior relaxes towards kf a b = kr c near the wall
    nx = 201
    kf, kr = 1e3, 1e2
    p = PH.PhysicalProblem(D=[1e-9, 1e-9, 1e-9], charges=[0.0, 0.0, 0.0], beta=BETA, eps=EPS, dx=1e-6, nx=nx,
                           c_bulk=[1.0, 2.0, 0.5], phiM=0.0, reactions=[{'lhs': [0, 1], 'rhs': [2], 'kf': kf, 'kr': kr}])
    c0 = np.repeat(p.c_bulk[:, None], nx, axis=1)
    c, phi, it, _ = PH.newton_step(p, c0, np.zeros(nx), c0, np.inf, tol=1e-12)
    assert it <= 10
    q = kf * c[0, 0] * c[1, 0] / (kr * c[2, 0])
    assert abs(q - 1.0) < 1e-6
    # element conservation with equal D and zero wall flux: a + c and b + c are harmonic -> constant
    assert np.abs((c[0] + c[2]) - 1.5).max() < 1e-9 and np.abs((c[1] + c[2]) - 2.5).max() < 1e-9


def test_unrepresentable_crowding_is_reported_not_nan():
    a = 4.1e-10
    p, c0, phi0, lam = binary(-1.0, cb=100.0, mpb_radius=[a, a])       # Dirichlet -1 V at the plane of closest approach
    c, phi, it, _ = PH.newton_step(p, c0, phi0, c0, np.inf, maxit=40)
    assert it == 41 and np.all(np.isfinite(c)) and np.all(np.isfinite(phi))


# ---- inputs pinned by the reference ------------------------------------------------------------------------
import glob      # noqa: E402
import json      # noqa: E402
import os        # noqa: E402
import re        # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLDEN, '*_n2_*.npz'))), ids=lambda p: os.path.basename(p)[:-4])
def test_gouy_chapman_against_the_references_own_arrays(path):
    """Every two-species fixture stores tp.gouy_chapman(x) as the REFERENCE evaluated it (tests/golden/make_golden.py:108): the
    analytic answer that pins the oracle is the reference's number, not the oracle's own formula."""
    d = np.load(path)
    gc = PH.gouy_chapman(d['xmesh'], float(d['phiM']), float(d['beta']), abs(float(d['charges'][0])), float(d['debye_length']))
    ref = d['gouy_chapman'][:, 0]
    assert np.abs(gc - ref).max() <= 1e-14 * max(np.abs(ref).max(), 1e-300)
    # ... and the stationary solve of the oracle on the fixture's own electrolyte, grid and wall potential approaches it
    # (the reference's two default species are a 1:1 electrolyte; the grid is the fixture's, so only a loose bound holds)
    if float(d['dx']) < 0.5 * float(d['debye_length']) and float(d['xmax']) > 8 * float(d['debye_length']):
        p = PH.PhysicalProblem(D=d['D'], charges=d['charges'], beta=float(d['beta']), eps=float(d['eps']), dx=float(d['dx']),
                               nx=int(d['nx']), c_bulk=d['c_bulk'], phiM=float(d['phiM']), x=d['xmesh'])
        c0 = np.repeat(np.asarray(d['c_bulk'], float)[:, None], int(d['nx']), axis=1)
        c, phi, it, _ = PH.newton_step(p, c0, np.zeros(int(d['nx'])), c0, np.inf, tol=1e-11, maxit=60)
        assert it <= 60 and np.abs(phi - ref).max() < 0.02 * abs(float(d['phiM']))


def _number(expr):
    """'1.957e-09[m^2/s]' -> 1.957e-09;  '(2e-05+flux_factor*0.0)[F/cm^2]' is evaluated with flux_factor = 1."""
    body = re.sub(r'\[[^\]]*\]\s*$', '', expr.strip())
    return float(eval(body, {'__builtins__': {}}, {'flux_factor': 1.0, 'PZC_factor': 1.0}))


def _runpy_transport():
    import collections
    from catint_amd.transport import Transport
    from catint_amd.units import unit_NA
    case = next(c for c in json.load(open(os.path.join(GOLDEN, 'transport_cases.json'))) if c['input']['name'] == 'co2r_runpy')['input']
    system = dict(case['system'])
    system['active site density'] = 9.61e-05 / unit_NA * (1e10) ** 2
    return Transport(species=collections.OrderedDict((n, dict(d)) for n, d in case['species']),
                     electrode_reactions=case['electrode_reactions'], electrolyte_reactions=case['electrolyte_reactions'],
                     system=system, nx=case['nx'], descriptors={'phiM': [-0.5, -0.6]})


def test_model_coefficients_are_what_the_reference_hands_to_comsol():
    """tests/golden/comsol_model_runpy.json = every parameter / variable / source expression catint/comsol_model.py emits for the
    run.py system (generated by running the reference).  The numbers that reach our solver and its oracle from
    catint_amd.transport.Transport + Calculator._physical_solver are those."""
    fx = json.load(open(os.path.join(GOLDEN, 'comsol_model_runpy.json')))['co2r_runpy']
    par, var = fx['pairs']['param'], fx['pairs']['variables_domain']
    tp = _runpy_transport()
    names = list(tp.species.keys())
    assert names == fx['species_order']
    for i, sp in enumerate(names):
        assert _number(par['D%d' % (i + 1)]) == tp.D[i] and int(par['Z%d' % (i + 1)]) == tp.species[sp]['charge']
        assert _number(par['ci%d' % (i + 1)]) == float(repr(float(tp.species[sp]['bulk_concentration'])))
    assert float(par['L_cell']) == tp.xmax and _number(par['T']) == tp.system['temperature'] and float(par['eps_r']) == tp.system['epsilon']
    assert _number(par['lambdaD']) == tp.debye_length
    # Stern Robin wall (comsol_model.py:982 rho_s = ((phiM-phiPZC)-phi)*CS; CS in F/cm^2, :1160-1167): what _physical_solver passes
    assert var['rho_s'] == '((phiM-phiPZC)-phi)*CS'
    cs_ref = _number(par['CS']) * 1e4                                 # F/cm^2 -> F/m^2 at the end of the ramp
    assert abs(cs_ref - float(tp.system['Stern capacitance']) * 1e-2) < 1e-15        # catint_amd/calculator.py:_physical_solver
    # continuation: the reference ramps phiPZC from phiM (no surface charge) to its value with flux_factor 0 -> 1 (:1147-1155);
    # ours walks phiM from phiPZC to its value -- the same family of problems from the uncharged wall to the target
    ramp = re.sub(r'\[V\]$', '', par['phiPZC'])
    at = lambda ff: eval(ramp, {'__builtins__': {}}, {'flux_factor': ff})
    assert at(0.0) == tp.system['phiM'] and abs(at(1.0) - tp.system['phiPZC']) < 1e-15
    assert fx['par_name'] == 'flux_factor' and fx['par_values'] == 'range(0,0.01,1)'
    # size-modified model (:1041-1063): phi0 = N_A sum a_i^3 c_i, gamma = 1/(1-phi0), activities c*gamma, drift -D phi0'/(1-phi0)
    assert var['phi_zero'] == 'N_A_const*(+a1^3*cp1)' and var['gamma'] == '1./(1.-phi_zero)' and var['ap3'] == 'cp3*gamma'
    assert var['tds.u5'] == '-D5*phi_zero_grad/(1-phi_zero)'
    assert _number(var['a1']) == tp.species['K+']['MPB_radius']
    # rate constants (:1064-1084), in the order of tp.reactions
    for j, rx in enumerate(tp.reactions.values()):
        assert _number(var['k%df' % (j + 1)]) == rx['rates'][0] and _number(var['k%dr' % (j + 1)]) == rx['rates'][1]


def test_reaction_source_equals_the_expressions_the_reference_emits():
    """R_cp1..R_cp7 as catint/comsol_model.py:781-867 writes them (mass action in activities, excluded species at the standard
    concentration), evaluated at random states, against oracle.pnp_physical.reaction_rates fed from our Transport's table."""
    fx = json.load(open(os.path.join(GOLDEN, 'comsol_model_runpy.json')))['co2r_runpy']
    var, tds = fx['pairs']['variables_domain'], fx['pairs']['physics_tds']
    tp = _runpy_transport()
    names = list(tp.species.keys())
    N = len(names)
    rx = [{'lhs': [names.index(x) for x in r['reactants'][0]], 'rhs': [names.index(x) for x in r['reactants'][1]],
           'kf': r['rates'][0], 'kr': r['rates'][1]} for r in tp.reactions.values()]
    radii = [tp.species[s].get('MPB_radius', 0.0) for s in names]
    p = PH.PhysicalProblem(D=tp.D, charges=tp.charges, beta=tp.beta, eps=tp.eps, dx=1e-9, nx=5, c_bulk=np.ones(N), phiM=0.0,
                           mpb_radius=radii, reactions=rx)
    rng = np.random.default_rng(0)
    c = np.exp(rng.uniform(np.log(1e-4), np.log(300.0), (N, 5)))
    R, _ = PH.reaction_rates(p, c)
    for i in range(5):
        env = {'conc_std': 1.0, 'N_A_const': PH.N_AVOGADRO}
        for k in range(N):
            env['cp%d' % (k + 1)] = c[k, i]
        env['a1'] = _number(var['a1'])
        env['phi_zero'] = eval(var['phi_zero'].replace('^', '**'), {'__builtins__': {}}, env)
        env['gamma'] = eval(var['gamma'], {'__builtins__': {}}, env)
        for k in range(N):
            env['ap%d' % (k + 1)] = eval(var['ap%d' % (k + 1)], {'__builtins__': {}}, env)
        for key in var:
            if re.fullmatch(r'k\d+[fr]', key):
                env[key] = _number(var[key])
        for k in range(N):
            ref = eval(tds['R_cp%d' % (k + 1)], {'__builtins__': {}}, env)
            scale = max(abs(ref), max(abs(env['k5r'] * env['ap6'] * env['ap7']), 1e-300) * 1e-3)
            assert abs(R[k, i] - ref) <= 1e-12 * scale, (k, i, R[k, i], ref)


def test_newton_solution_is_the_root_an_independent_solver_finds():
    """The oracle's damped Newton (its own Jacobian, its own block solver) against scipy.optimize.root (MINPACK hybrd with a
    finite-difference Jacobian) on the same discrete residual: Stern wall, steric ions, a buffer reaction, Butler-Volmer / Langmuir
    wall kinetics, graded grid, stationary and one backward-Euler step."""
    from scipy.optimize import root
    from catint_amd.host import graded_mesh
    nx = 20
    dx = 2.5e-10
    x = graded_mesh(60.0 * nx * dx, dx, nx)
    wk = [{'species': 2, 'k': 4e-3, 'nu': [0.0, 1.0, -1.0], 'alpha': -9.0, 'saturation': 0.05}]
    p = PH.PhysicalProblem(D=[1.957e-9, 1.185e-9, 1.91e-9], charges=[F, -F, 0.0], beta=BETA, eps=EPS, dx=dx, nx=nx, c_bulk=[100.0, 100.0, 34.0],
                           phiM=-0.35, stern_capacitance=0.2, phi_pzc=0.05, mpb_radius=[4.1e-10, 2e-10, 0.0],
                           reactions=[{'lhs': [2], 'rhs': [1, 1], 'kf': 3e2, 'kr': 1e-1}], wall_kinetics=wk, x=x)
    c0 = np.repeat(np.array(p.c_bulk)[:, None], nx, axis=1)
    for dt in (np.inf, 2e-7):
        c, phi, it, _ = PH.newton_step(p, c0, np.zeros(nx), c0, dt, tol=1e-12, maxit=80)
        assert it <= 80

        def fun(u):
            u = u.reshape(4, nx)
            return PH.residual(p, u[:3], u[3], c0, dt).reshape(-1)
        # start the independent solver near (not at) the oracle's answer: hybrd is a local method and the double layer is stiff
        u0 = np.concatenate([c * (1 + 1e-3 * np.sin(np.arange(nx))), (phi + 1e-4)[None]], 0).reshape(-1)
        sol = root(fun, u0, method='hybr', tol=1e-13, options={'maxfev': 40000})
        assert sol.success, sol.message
        u = sol.x.reshape(4, nx)
        assert np.abs(u[:3] - c).max() <= 1e-8 * np.abs(c).max() and np.abs(u[3] - phi).max() <= 1e-9
        assert np.abs(fun(np.concatenate([c, phi[None]], 0).reshape(-1))).max() <= 1e-9 * max(1.0, np.abs(fun(u0)).max())


def test_rounding_floor_exit_of_the_newton_iteration():
    """Fuzz case 117 (tests/golden/fuzz/newton_case117.json): after quadratic convergence the scaled update sits at ~3e-9 -- the rounding
    floor of an ill-conditioned Jacobian -- above tol = 1e-10.  The iteration leaves there (two consecutive full steps within 100 tol,
    the second not smaller than the first) instead of running until the noise dips below tol; the state it leaves is a root of the
    residual."""
    import json
    import os
    from tests.test_gpu_newton import BETA, EPS, make_lanes
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'fuzz', 'newton_case117.json')))
    N, nx = d['N'], d['nx']
    D, q, cb, dx, phiM = make_lanes(N, nx, d['B'], d['seed'], phi_lo=-0.3, phi_hi=0.3, points_per_debye=d['points_per_debye'])
    p = PH.PhysicalProblem(D=D, charges=q, beta=BETA, eps=EPS, dx=dx, nx=nx, c_bulk=cb[0], phiM=phiM[0], flux=np.array(d['flux'])[0],
                           reactions=d['reactions'], x=np.array(d['x']) * dx)
    c0 = np.repeat(cb[0][:, None], nx, axis=1)
    c, phi, it, hist = PH.newton_step(p, c0, np.zeros(nx), c0, np.inf, tol=1e-10, maxit=60)
    assert 20 <= it <= 24 and hist[16] < 1e-7 and hist[15] > 1e-5           # quadratic phase ends at iteration 17 ...
    assert all(1e-11 < h < 1e-8 for h in hist[17:])                         # ... then the floor
    assert hist[-1] >= hist[-2] and hist[-1] > 1e-10                        # left on the stagnation rule, not on update < tol
    F = PH.residual_and_jacobian(p, c, phi, c0, np.inf)[0]
    F0 = PH.residual_and_jacobian(p, c0, np.zeros(nx), c0, np.inf)[0]
    assert np.abs(F).max() < 1e-9 * np.abs(F0).max()


def test_rounding_floor_rule_rejects_linear_convergence():
    """a linearly converging Newton iteration (singular Jacobian at the root, ratio >= 0.5) must not leave through the
    rounding-floor exit -- its error is upd / (1 - ratio), far above tol.  The rule (the same formula as newton_at_rounding_floor in
    catint_amd/csrc/pnp_math.h) asks for an update that did NOT shrink; a geometric sequence shrinks at every step."""
    tol = 1e-10
    for ratio in (0.5001, 0.7, 0.9, 0.99, 0.999999):
        u = [50 * tol * ratio ** k for k in range(400)]
        assert not any(PH.at_rounding_floor(u[k], u[k - 1], tol) for k in range(1, 400)), ratio
    # noise at the floor: fluctuating updates of one size are accepted at the first step that does not shrink
    rng = np.random.default_rng(0)
    for _ in range(50):
        u = 30 * tol * rng.uniform(0.8, 1.25, 12)
        assert any(PH.at_rounding_floor(u[k], u[k - 1], tol) for k in range(1, 12))
    # outside the window, during quadratic contraction, or after a damped step (upd_prev = inf): never
    assert not PH.at_rounding_floor(3e-9, 2e-7, tol) and not PH.at_rounding_floor(3e-9, np.inf, tol)
    assert not PH.at_rounding_floor(1e-9, 3e-9, tol) and PH.at_rounding_floor(3.1e-9, 3e-9, tol)


def test_bdf2_is_second_order_in_time():
    """integrate(bdf2=True): the reference's transient study asks COMSOL for BDF with maxorder 2 (comsol_model.py:518-531).  On the
    relaxation of a double layer the error against a fine-step solution falls four-fold when dt is halved (backward Euler: two-fold)."""
    from catint_amd.units import unit_F, unit_R, unit_eps0
    nx = 64
    D, q, cb = np.array([1.957e-9, 1.185e-9]), np.array([unit_F, -unit_F]), np.array([10.0, 10.0])
    beta, eps = 1.0 / (unit_R * 298.14), 78.36 * unit_eps0
    dx = np.sqrt(eps / beta / (q ** 2 * cb).sum()) / 4.0
    p = PH.PhysicalProblem(D=D, charges=q, beta=beta, eps=eps, dx=dx, nx=nx, c_bulk=cb, phiM=-0.05)
    c0 = np.repeat(cb[:, None], nx, axis=1)
    T = 0.5 * (nx * dx) ** 2 / D.max()
    ref = PH.integrate(p, c0, np.zeros(nx), T / 512, 512, bdf2=True, tol=1e-13)[0]
    ratio = {}
    for bdf2 in (False, True):
        e = [np.abs(PH.integrate(p, c0, np.zeros(nx), T / n, n, bdf2=bdf2, tol=1e-13)[0] - ref).max() for n in (16, 32)]
        ratio[bdf2] = e[0] / e[1]
    assert 1.7 < ratio[False] < 2.4 and 3.2 < ratio[True] < 5.0, ratio


def test_jacobian_once_is_the_chord_iteration_same_fixed_point_more_iterations():
    """newton_step(jacobian_once=True): the Jacobian of the first iterate serves the whole step (comsol_model.py:526,530 jtech "once").
    Same equations, same stopping rule: the step ends on the same state to the tolerance, linearly instead of quadratically -- more
    iterations, each update a roughly constant fraction of the one before (tools/probe/jacobian_once_oracle.py measures what that costs)."""
    from catint_amd.units import unit_F, unit_R, unit_eps0
    nx = 64
    D, q, cb = np.array([1.957e-9, 1.185e-9]), np.array([unit_F, -unit_F]), np.array([10.0, 10.0])
    beta, eps = 1.0 / (unit_R * 298.14), 78.36 * unit_eps0
    dx = np.sqrt(eps / beta / (q ** 2 * cb).sum()) / 4.0
    p = PH.PhysicalProblem(D=D, charges=q, beta=beta, eps=eps, dx=dx, nx=nx, c_bulk=cb, phiM=-0.08)
    c0 = np.repeat(cb[:, None], nx, axis=1)
    dt = 0.02 * (nx * dx) ** 2 / D.max()
    c1, p1, it1, h1 = PH.newton_step(p, c0, np.zeros(nx), c0, dt, tol=1e-10)
    c2, p2, it2, h2 = PH.newton_step(p, c0, np.zeros(nx), c0, dt, tol=1e-10, jacobian_once=True)
    assert it1 <= 50 and it2 <= 50 and it2 > it1, (it1, it2)
    assert np.abs(c2 - c1).max() <= 1e-8 * np.abs(c1).max() and np.abs(p2 - p1).max() <= 1e-9
    assert h1[0] == h2[0]                                     # the first iteration is Newton's
    tail = np.array(h2[-4:])
    assert np.all(tail[1:] < tail[:-1]) and np.all(tail[1:] > 1e-4 * tail[:-1])      # linear: no update a millionth of the one before


def test_convection_velocity_analytic_profile_and_flux_closure():
    """Constant velocity v along x (tp.system['flow rate'], comsol_model.py:901-903): flux -D c' + c v.  A neutral species with a closed
    wall relaxes to c(x) = c_L exp(v (x - L) / D); the exponentially fitted edge flux reproduces it to rounding on any grid, for both
    signs of v.  With a prescribed wall flux j the stationary flux is j everywhere: c(x) = j/v + (c_L - j/v) exp(v (x - L) / D)."""
    D = np.array([1.5e-9, 2.0e-9])
    q = np.array([0.0, 0.0])
    nx, dx = 60, 2e-9
    x = np.cumsum(np.concatenate([[0.0], np.geomspace(0.5, 2.0, nx - 1)])) * dx
    L = x[-1]
    cb = np.array([3.0, 7.0])
    for v in (0.04, -0.03):
        for j in (np.zeros(2), np.array([2e-3, -1e-3])):
            p = PH.PhysicalProblem(D=D, charges=q, beta=BETA, eps=EPS, dx=dx, nx=nx, c_bulk=cb, phiM=0.0, flux=j, x=x, velocity=v)
            c0 = np.repeat(cb[:, None], nx, axis=1)
            c, phi, it, _ = PH.newton_step(p, c0, np.zeros(nx), c0, np.inf, tol=1e-12)
            assert it <= 50
            for k in range(2):
                exact = j[k] / v + (cb[k] - j[k] / v) * np.exp(v * (x - L) / D[k])
                assert np.abs(c[k] - exact).max() < 1e-10 * np.abs(exact).max()
    # and the Jacobian with the term switched on (finite differences), charged species, steric ions, Stern wall
    rng = np.random.default_rng(3)
    p = PH.PhysicalProblem(D=[1.9e-9, 1.2e-9, 5e-9], charges=[PH.F_CONST if hasattr(PH, 'F_CONST') else 96485.33212, -96485.33212, 0.0], beta=BETA,
                           eps=EPS, dx=dx, nx=24, c_bulk=[5.0, 5.0, 1.0], phiM=0.05, stern_capacitance=0.2, mpb_radius=[4e-10, 3e-10, 0.0],
                           velocity=0.05)
    c = np.array([5.0, 5.0, 1.0])[:, None] * (1 + 0.2 * rng.uniform(-1, 1, (3, 24)))
    phi = 0.02 * rng.uniform(-1, 1, 24)
    F, Lb, Mb, Ub = PH.residual_and_jacobian(p, c, phi, c, np.inf)
    i, k, h = 7, 1, 1e-6
    cp = c.copy(); cp[k, i] += h
    cm = c.copy(); cm[k, i] -= h
    dF = (PH.residual(p, cp, phi, c, np.inf) - PH.residual(p, cm, phi, c, np.inf)) / (2 * h)
    assert np.allclose(dF[:, i], Mb[i, :, k], rtol=1e-6, atol=1e-9) and np.allclose(dF[:, i + 1], Lb[i + 1, :, k], rtol=1e-6, atol=1e-9)
