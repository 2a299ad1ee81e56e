"""GPU: the Calculator counterpart as a drop-in for the reference's seam
`Calculator.integrate_pnp(dx,nx,dt,nt,ntout,method)` (calculator_old.py:210), and the batched descriptor
sweep.  Tolerance as in test_gpu_parity_golden.py (rtol 1e-9 on the max-norm; measured ~1e-13)."""
import os

import numpy as np
import pytest

from oracle import pnp_ref as R
from oracle import c_oracle as CO
from tests.test_host_transport import transport_from_fixture
from catint_amd.calculator import Calculator, make_itout

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
RTOL = 1e-9


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize('name', ['cn_dd_n2_nx50', 'cn_dd_n2_nx200', 'cn_dd_n6_nx40', 'cn_dd_n3_nx64_flux', 'cn_dd_n3_nx127_vzeta', 'ftcs_dd_n2_nx50', 'cn_dd_n2_nx50_LF',
                                  'ftcs_dd_n2_nx50_LF', 'cn_defaultpb_n2_nx50_gc', 'ftcs_defaultpb_n2_nx50_gc'])
def test_integrate_pnp_drop_in(name):
    d = np.load(os.path.join(GOLDEN, name + '.npz'))
    tp = transport_from_fixture(d)
    if 'gc' in name:
        tp.set_initial_concentrations('Gouy-Chapman')
        assert np.array_equal(tp.c0, d['c0'])
    # the remaining inputs of the path are plain attributes the caller may set (as the fixture generator did)
    tp.c0 = d['c0'].copy()
    tp.flux_bound = d['flux_bound'].copy()
    tp.system['vzeta'] = float(d['vzeta'])
    method = str(d['method'])
    ntout = next(n for n in range(1, 8) if make_itout(int(d['nt']), n) == [int(i) for i in d['itout']])
    calc = Calculator(transport=tp, calc=method, dt=float(d['dt']), tmax=float(d['tmax']), ntout=ntout)
    cout = calc.integrate_pnp(tp.dx, tp.nx, tp.dt, tp.nt, tp.ntout, calc.calc)
    assert len(cout) == len(d['cout'])
    for a, b in zip(cout, d['cout']):
        assert a.shape == b.shape and relerr(a, b) < RTOL
    assert relerr(tp.potential, d['potential']) < RTOL
    assert relerr(tp.efield, d['efield']) < RTOL
    assert relerr(tp.total_charge, d['total_charge']) < RTOL


def test_descriptor_sweep_is_batched_and_matches_oracle():
    import collections
    from catint_amd.transport import Transport
    phis = list(np.linspace(-0.03, 0.03, 13))
    species = collections.OrderedDict([('K+', {'bulk_concentration': 30.0}), ('Cl-', {'bulk_concentration': 10.0}),
                                       ('HCO3-', {'bulk_concentration': 20.0})])
    tp = Transport(species=species, system={'phiM': 0.0, 'boundary thickness': 2e-8}, nx=96,
                   pb_bound={'potential': {'wall': 'phiM', 'bulk': 0.0}}, descriptors={'phiM': phis})
    calc = Calculator(transport=tp, calc='Crank-Nicolson', dt=2e-11, tmax=3e-10, ntout=3)
    cout = calc.run()
    assert cout.shape == (len(tp.itout), len(phis), tp.nspecies * tp.nx)
    for i, phi in enumerate(phis):
        p = R.Problem(D=tp.D, charges=tp.charges, beta=tp.beta, eps=tp.eps, dx=tp.dx, nx=tp.nx, dt=tp.dt,
                      pb=np.array([phi, 0.0, np.nan, np.nan]), vzeta=phi, flux_bound=np.zeros(3))
        ref, (v, g, l) = R.integrate(p, tp.c0, tp.nt, tp.itout, 'Crank-Nicolson', solver='banded')
        assert relerr(cout[:, i], np.array(ref)) < RTOL
        sysd = tp.alldata[i]['system']
        assert sysd['phiM'] == phi and abs(sysd['surface_potential'] - phi) < 1e-15
        # phi = 0 is one of the sweep points: there the potential is pure round-off, so scale by 1 mV
        assert np.abs(sysd['potential'] - v).max() / max(np.abs(v).max(), 1e-3) < RTOL
        assert relerr(tp.alldata[i]['species']['Cl-']['concentration'], np.array(ref)[-1].reshape(3, -1)[1]) < RTOL


def test_sharded_batch_equals_whole_batch():
    """two handles on one GPU, each owning a contiguous block of lanes == one handle with all lanes"""
    from catint_amd.synthetic import make_batch
    from catint_amd.host import solver_from_problem
    from catint_amd.parallel import shard_bounds
    B = 48
    p, c0, pb, vz, fl = make_batch(B, 3, 130, seed=5, phi_max=0.02, dt_factor=1e-4)
    with solver_from_problem(p, 'Crank-Nicolson', batch_capacity=B) as s:
        s.set_batch(c0, pb, vz, fl)
        s.step(9)
        whole = s.get_state()
    parts = []
    for r in range(2):
        lo, hi = shard_bounds(B, 2, r)
        with solver_from_problem(p, 'Crank-Nicolson', batch_capacity=hi - lo) as s:
            s.set_batch(c0[lo:hi], pb[lo:hi], vz[lo:hi], fl[lo:hi])
            s.step(9)
            parts.append(s.get_state())
    for k in range(4):
        assert np.array_equal(whole[k], np.concatenate([parts[0][k], parts[1][k]], axis=0))


def test_scf_cycle_on_gpu_converges_and_is_lane_consistent():
    """SCF outer loop (calculator.py:294-406) around the GPU transport solve with an analytic kinetics callback."""
    import collections
    from catint_amd.transport import Transport
    species = collections.OrderedDict([('K+', {'bulk_concentration': 30.0}), ('Cl-', {'bulk_concentration': 30.0}),
                                       ('CO2', {'bulk_concentration': 20.0, 'diffusion': 1.91e-9, 'symbol': 'CO_2'})])
    phis = [-0.01, -0.015, -0.02, -0.025]
    tp = Transport(species=species, system={'phiM': -0.01, 'boundary thickness': 2e-8}, nx=64,
                   pb_bound={'potential': {'wall': 'phiM', 'bulk': 0.0}}, descriptors={'phiM': phis})
    calc = Calculator(transport=tp, calc='Crank-Nicolson', dt=2e-11, tmax=4e-10, ntout=1, tau_scf=1e-4, mix_scf=0.5)

    def flux_cb(state):   # first-order consumption of CO2 at the wall, potential dependent
        k = 2e-3 * np.exp(-20.0 * (state['phiM'] + 0.01))
        f = np.zeros((len(phis), 3))
        f[:, 2] = -k * np.maximum(state['surface_concentration'][:, 2], 0.0) * 1e-3
        return f

    out = calc.run_scf_cycle(flux_cb, max_iter=200)
    assert out['converged'].all() and out['iterations'] < 200
    assert (out['flux'][:, 2] < 0).all() and (np.diff(np.abs(out['flux'][:, 2])) > 0).all()   # more negative phi -> faster
    # each lane on its own gives the same answer as inside the batch
    tp1 = Transport(species=species, system={'phiM': phis[2], 'boundary thickness': 2e-8}, nx=64,
                    pb_bound={'potential': {'wall': 'phiM', 'bulk': 0.0}}, descriptors={'phiM': [phis[2]]})
    calc1 = Calculator(transport=tp1, calc='Crank-Nicolson', dt=2e-11, tmax=4e-10, ntout=1, tau_scf=1e-4, mix_scf=0.5)
    out1 = calc1.run_scf_cycle(lambda st: flux_cb({'phiM': np.array(phis), 'surface_concentration':
                                                    np.repeat(st['surface_concentration'], 4, axis=0)})[2:3], max_iter=200)
    assert np.allclose(out1['surface_concentration'][0], out['surface_concentration'][2], rtol=1e-10)


@pytest.mark.parametrize('name', ['odeint_dd_n2_nx50', 'odeint_dd_n3_nx40_flux_LF', 'dopri5_dd_n2_nx50'])
def test_method_of_lines_rhs_and_trajectories(name):
    """ode_func (calculator_old.py:827-935) on the GPU: the RHS against the reference's own RHS samples
    (rtol 1e-11 of the max-norm; the device sums in a different order), and the scipy-driven trajectories
    against the reference trajectories to the integrators' own tolerances (odeint rtol 1.5e-8, dopri5 1e-6)."""
    from catint_amd.host import solver_from_problem
    d = np.load(os.path.join(GOLDEN, name + '.npz'))
    p, c0, nt, itout, method = R.problem_from_golden(d)
    if 'rhs_states' in d:
        with solver_from_problem(p, 'FTCS', batch_capacity=3) as s:
            s.set_batch(d['rhs_states'], np.stack([p.pb] * 3), [p.vzeta] * 3, np.stack([p.flux_bound] * 3))
            f = s.mol_rhs(d['rhs_states'])
        for a, b in zip(f, d['rhs_values']):
            assert relerr(a, b) < 1e-11
    tp = transport_from_fixture(d)
    tp.c0 = d['c0'].copy()
    tp.flux_bound = d['flux_bound'].copy()
    tp.system['vzeta'] = float(d['vzeta'])
    ntout = next(n for n in range(1, 8) if make_itout(int(d['nt']), n) == [int(i) for i in d['itout']])
    calc = Calculator(transport=tp, calc=str(d['method']), dt=float(d['dt']), tmax=float(d['tmax']), ntout=ntout)
    cout = calc.integrate_pnp(tp.dx, tp.nx, tp.dt, tp.nt, tp.ntout, calc.calc)
    assert len(cout) == len(d['cout'])
    tol = 1e-5 if 'dopri5' in name else 1e-6
    for a, b in zip(cout, d['cout']):
        assert relerr(a, b) < tol


def test_co2r_polarization_example_runs_and_saturates():
    """examples/co2r_polarization_sweep.py (BASELINE configs[2] shape, reduced): the Tafel current grows with
    overpotential while CO2 depletes at the wall; every lane converges."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        'co2r', os.path.join(os.path.dirname(GOLDEN), '..', 'examples', 'co2r_polarization_sweep.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.main(['--lanes', '24', '--tmax', '0.02', '--dt', '5e-6', '--mix-scf', '0.3', '--tau-scf', '1e-3'])
    assert out['converged'].all()
    j = out['current_density'][:, 3]
    assert (j > 0).all() and (np.diff(j) >= -1e-12).all()           # more negative phiM -> larger CO current
    assert out['surface_concentration'][-1, 1] < out['surface_concentration'][0, 1]   # CO2 depletes at the wall
    assert (out['surface_concentration'][:, 2] > 6.31e-05).all()     # OH- is produced at the wall


# ---- physical mode behind the reference's calculator name 'comsol' -------------------------------------------------
def _physical_transport(phis, mpb=True, **system):
    import collections
    from catint_amd.transport import Transport
    species = collections.OrderedDict([('K+', dict({'bulk_concentration': 100.0}, **({'MPB_radius': 4.1e-10} if mpb else {}))),
                                       ('HCO3-', {'bulk_concentration': 100.0}),
                                       ('CO2', {'bulk_concentration': 34.0}), ('CO', {'bulk_concentration': 0.0})])
    sysd = {'phiM': phis[0], 'boundary thickness': 4e-8, 'Stern capacitance': 20.0, 'phiPZC': 0.1}
    sysd.update(system)
    return Transport(species=species, system=sysd, nx=256, descriptors={'phiM': list(phis)})


def test_comsol_calculator_is_the_implicit_gpu_solve_and_matches_the_oracle():
    from oracle import pnp_physical as PH
    phis = [-1.2, -0.6, 0.0, 0.4]
    tp = _physical_transport(phis)
    calc = Calculator(transport=tp, calc='comsol')
    assert calc.mode == 'stationary'
    calc.run()
    assert np.all(calc.status == 0) and calc.newton_iterations.max() <= 50
    radii = [tp.species[sp].get('MPB_radius', 0.0) for sp in tp.species]
    for i, phiM in enumerate(phis):
        p = PH.PhysicalProblem(D=tp.D, charges=tp.charges, beta=tp.beta, eps=tp.eps, dx=tp.dx, nx=tp.nx,
                               c_bulk=[tp.species[sp]['bulk_concentration'] for sp in tp.species], phiM=phiM,
                               stern_capacitance=0.2, phi_pzc=0.1, mpb_radius=radii)
        c0 = tp.c0.reshape(tp.nspecies, tp.nx)
        c, phi, it, _ = PH.newton_step(p, c0, np.zeros(tp.nx), c0, np.inf, tol=1e-8, maxit=50)
        assert it <= 50
        d = tp.alldata[i]
        got = np.array([d['species'][sp]['concentration'] for sp in tp.species])
        assert np.abs(got - c).max() <= 1e-7 * np.abs(c).max()
        assert np.abs(d['system']['potential'] - phi).max() <= 1e-7
        # most of the applied potential drops over the Stern layer; the diffuse layer sees what is left
        assert abs(d['system']['surface_potential']) < abs(phiM - 0.1)
    # derived fields of the COMSOL reader (comsol_reader.py:57-90): activity coefficient, (no H+/OH- here -> no pH)
    d0 = tp.alldata[0]
    g0 = 1.0 / (1.0 - 6.022140857e23 * 4.1e-10 ** 3 * np.array(d0['species']['K+']['concentration']))
    assert np.allclose(d0['species']['CO2']['activity_coefficient'], g0) and d0['species']['K+']['surface_activity_coefficient'] == g0[0]
    assert np.allclose(d0['system']['efield'][1:-1], -(np.array(d0['system']['potential'])[2:] - np.array(d0['system']['potential'])[:-2]) / (2 * tp.dx))
    # derived electrolyte outputs (comsol_model.py:1010-1040): no wall flux -> no current, no ohmic drop; bulk conductivity
    kb = 96485.33289 ** 2 * tp.beta * (tp.D[0] * 100.0 + tp.D[1] * 100.0)
    assert np.isclose(d0['system']['conductivity'][-1], kb, rtol=1e-6)
    assert np.abs(d0['system']['electrolyte_current_density']).max() < 1e-2 and abs(d0['system']['delta_phi_iR_inf']) < 1e-9
    assert np.isclose(d0['system']['delta_phi_inf'], -d0['system']['surface_potential'])
    cK = [tp.alldata[i]['species']['K+']['surface_concentration'] for i in range(len(phis))]
    assert cK[0] > cK[1] > cK[2] > cK[3]                   # cations pile up at negative potentials ...
    assert cK[0] < 1.0 / (6.022140857e23 * 4.1e-10 ** 3)   # ... but never beyond close packing


def test_scf_cycle_with_the_physical_mode_reaches_the_diffusion_limited_plateau():
    phis = list(np.linspace(-0.6, -1.4, 9))
    tp = _physical_transport(phis, mpb=False)      # point ions: a neutral species then diffuses freely (linear profile)
    calc = Calculator(transport=tp, calc='comsol', tau_scf=1e-6, mix_scf=0.02)     # run.py:95
    iCO2, iCO = 2, 3
    L = (tp.nx - 1) * tp.dx

    def flux_cb(state):   # Tafel kinetics, first order in the surface CO2 concentration; CO is produced 1:1
        k = 1e-4 * np.exp(-12.0 * (state['phiM'] + 0.6))
        f = np.zeros((len(phis), tp.nspecies))
        f[:, iCO2] = -k * np.maximum(state['surface_concentration'][:, iCO2], 0.0)
        f[:, iCO] = -f[:, iCO2]
        return f

    out = calc.run_scf_cycle(flux_cb, nel=[1, 1, 2, 2], max_iter=3000)
    assert out['converged'].all() and not out['failed'].any()
    j = -out['flux'][:, iCO2]
    jlim = tp.D[iCO2] * 34.0 / L                       # diffusion-limited flux of a neutral species across the layer
    assert np.all(np.diff(j) > 0) and j[-1] < jlim and j[-1] > 0.9 * jlim
    # stationary neutral species: linear profile, c_surf = c_bulk - j L / D
    assert np.allclose(out['surface_concentration'][:, iCO2], 34.0 - j * L / tp.D[iCO2], rtol=1e-4, atol=1e-4)
    assert np.allclose(out['surface_concentration'][:, iCO], j * L / tp.D[iCO], rtol=1e-4, atol=1e-4)


def test_time_dependent_physical_mode_approaches_the_stationary_one():
    phis = [-0.5, 0.3]
    tp = _physical_transport(phis)
    lam = tp.debye_length
    L = (tp.nx - 1) * tp.dx
    dt = 0.5 * L * L / tp.D.min()
    stat = Calculator(transport=tp, calc='comsol')
    stat.run()
    ref = [np.array(tp.alldata[i]['system']['potential']) for i in range(2)]
    tp2 = _physical_transport(phis)
    trans = Calculator(transport=tp2, calc='comsol', mode='time-dependent', dt=dt, tmax=40 * dt)
    trans.run()
    assert np.all(trans.status == 0)
    for i in range(2):
        assert np.abs(np.array(tp2.alldata[i]['system']['potential']) - ref[i]).max() < 1e-6


def test_implicit_surface_kinetics_give_the_scf_fixed_point_in_one_solve():
    phis = list(np.linspace(-0.6, -1.6, 11))
    tp = _physical_transport(phis, mpb=False)
    calc = Calculator(transport=tp, calc='comsol')
    rate = lambda phiM: 1e-4 * np.exp(-12.0 * (phiM + 0.6))
    calc.set_surface_kinetics([{'species': 'CO2', 'rate': rate, 'stoichiometry': {'CO2': -1.0, 'CO': 1.0}}])
    calc.run()
    assert np.all(calc.status == 0)
    L = (tp.nx - 1) * tp.dx
    kap = rate(np.array(phis)) * L / tp.D[2]
    cs = np.array([tp.alldata[i]['species']['CO2']['surface_concentration'] for i in range(len(phis))])
    assert np.allclose(cs, 34.0 / (1.0 + kap), rtol=1e-7)                      # analytic: neutral species, linear profile
    assert np.allclose(-calc.kinetic_flux[:, 2], tp.D[2] * 34.0 / L * kap / (1.0 + kap), rtol=1e-7)
    assert kap[-1] > 300 and cs[-1] < 0.2                                       # deep in the diffusion-limited plateau
    # the same answer as the SCF loop around the transport solve (where the SCF converges at all)
    tp2 = _physical_transport(phis[:9], mpb=False)
    scf = Calculator(transport=tp2, calc='comsol', tau_scf=1e-7, mix_scf=0.02)

    def flux_cb(state):
        f = np.zeros((9, tp2.nspecies))
        f[:, 2] = -rate(state['phiM']) * np.maximum(state['surface_concentration'][:, 2], 0.0)
        f[:, 3] = -f[:, 2]
        return f
    out = scf.run_scf_cycle(flux_cb, max_iter=4000)
    assert out['converged'].all()
    assert np.allclose(out['surface_concentration'][:, 2], cs[:9], rtol=2e-4)


def test_device_scf_loop_walks_the_same_iterates_as_the_host_loop():
    """pnp_scf_cycle: mixing, fallback, accuracy and mix decay per lane on the device, against the batched
    host loop with the same analytic kinetics as a Python callback -- same iteration count, same per-lane bookkeeping."""
    phis = list(np.linspace(-0.6, -1.4, 9))
    rate = lambda phiM: 1e-4 * np.exp(-12.0 * (phiM + 0.6))
    kin = [{'species': 'CO2', 'rate': rate, 'stoichiometry': {'CO2': -1.0, 'CO': 1.0}}]
    outs = []
    for device in (False, True):
        tp = _physical_transport(phis, mpb=False)
        calc = Calculator(transport=tp, calc='comsol', tau_scf=1e-6, mix_scf=0.02)
        if device:
            calc.set_surface_kinetics(kin)
            outs.append(calc.run_scf_cycle(nel=[1, 1, 2, 2], max_iter=3000))
            assert calc.surface_kinetics == kin
        else:
            def cb(state):
                f = np.zeros((len(phis), 4))
                f[:, 2] = -rate(state['phiM']) * np.maximum(state['surface_concentration'][:, 2], 0.0)
                f[:, 3] = -f[:, 2]
                return f
            outs.append(calc.run_scf_cycle(cb, nel=[1, 1, 2, 2], max_iter=3000))
    host, dev = outs
    assert host['converged'].all() and dev['converged'].all() and not dev['failed'].any()
    assert dev['iterations'] == host['iterations'] and host['iterations'] > 100
    assert np.array_equal(dev['mix'], host['mix'])                              # the same number of decays per lane
    assert np.allclose(dev['surface_concentration'], host['surface_concentration'], rtol=1e-9, atol=1e-12)
    assert np.allclose(dev['flux'], host['flux'], rtol=1e-9, atol=1e-18)
    assert np.allclose(dev['accuracy'], host['accuracy'], rtol=1e-3, atol=1e-12)
    assert np.allclose(dev['current_density'], host['current_density'], rtol=1e-9)
    # and both sit on the fixed point the implicit solve finds directly
    L = (tp.nx - 1) * tp.dx
    kap = rate(np.array(phis)) * L / tp.D[2]
    assert np.allclose(dev['surface_concentration'][:, 2], 34.0 / (1.0 + kap), rtol=2e-4)


def test_butler_volmer_kinetics_implicit_solve_and_both_scf_loops_agree():
    """Rate law beyond first order (pnp_set_wall_rate_law): Butler-Volmer factor in the Stern-layer drop phiM - phi(0) and Langmuir
    saturation, with a charged product so that the kinetics feed back on the double layer.  The implicit solve (Jacobian carries
    d/dc_s and d/dphi_0), the device SCF loop (explicit, surface potential of the previous solve) and the host SCF loop with the
    same law as a Python callback land on the same state."""
    phis = list(np.linspace(-0.5, -1.0, 6))
    kin = [{'species': 'CO2', 'rate': lambda phiM: 2e-5 * np.ones_like(phiM), 'alpha': -8.0, 'saturation': 0.02,
            'stoichiometry': {'CO2': -1.0, 'CO': 1.0, 'HCO3-': 1.0}}]
    tp = _physical_transport(phis, mpb=False)
    calc = Calculator(transport=tp, calc='comsol')
    calc.set_surface_kinetics(kin)
    calc.run()
    assert np.all(calc.status == 0)
    cs = np.array([tp.alldata[i]['species']['CO2']['surface_concentration'] for i in range(len(phis))])
    v0 = np.array([tp.alldata[i]['system']['surface_potential'] for i in range(len(phis))])
    L = (tp.nx - 1) * tp.dx
    j = -calc.kinetic_flux[:, 2]
    E = np.exp(-8.0 * (np.array(phis) - v0))
    assert np.allclose(j, 2e-5 * cs / (1.0 + 0.02 * cs) * E, rtol=1e-12)
    assert np.allclose(cs, 34.0 - j * L / tp.D[2], rtol=1e-7)                    # neutral educt: linear profile carries the flux
    assert E[-1] > 20 * E[0] and j[-1] > 5 * j[0] and cs[-1] < cs[0]       # the potential dependence is live
    outs = []
    for device in (False, True):
        tp2 = _physical_transport(phis, mpb=False)
        scf = Calculator(transport=tp2, calc='comsol', tau_scf=1e-7, mix_scf=0.05)
        if device:
            scf.set_surface_kinetics(kin)
            outs.append(scf.run_scf_cycle(nel=[1, 1, 2, 2], max_iter=4000))
        else:           # no table on this Calculator: its transport solves see prescribed fluxes only
            cb = lambda st: scf.surface_kinetic_fluxes(st['surface_concentration'], st['phiM'], clip=True, reactions=kin,   # noqa: E731
                                                       vsurf=st['surface_potential'])
            outs.append(scf.run_scf_cycle(cb, nel=[1, 1, 2, 2], max_iter=4000))
    host, dev = outs
    assert host['converged'].all() and dev['converged'].all() and not dev['failed'].any()
    assert dev['iterations'] == host['iterations'] and host['iterations'] > 50
    assert np.allclose(dev['surface_concentration'], host['surface_concentration'], rtol=1e-9, atol=1e-12)
    assert np.allclose(dev['flux'], host['flux'], rtol=1e-9, atol=1e-18)
    assert np.allclose(dev['surface_concentration'][:, 2], cs, rtol=2e-4)
    assert np.allclose(-dev['flux'][:, 2], j, rtol=2e-4)


def test_device_scf_loop_freezes_converged_lanes_and_puts_failed_ones_back():
    """pnp_scf_cycle through the C-ABI with a hand-made loop state: lane 2 asks for a flux far beyond the diffusion limit (no
    solution in the positive cone) -> its solve fails, the lane reports -1 (the reference's negative-concentration answer),
    keeps the state of its last converged solve and never converges; lanes 0 and 1 converge and leave the batch."""
    from catint_amd.units import unit_F
    phis = np.array([-0.6, -0.7, -1.4])
    tp = _physical_transport(list(phis), mpb=False)
    tp.newton = {'maxit': 30}
    calc = Calculator(transport=tp, calc='comsol', tau_scf=1e-5, mix_scf=0.5)
    B, N = 3, 4
    solver = calc._physical_solver(B)
    try:
        st = calc.solve_physical(solver, np.repeat(tp.c0[None, :], B, axis=0), phis, np.zeros((B, N)), nramp=8)
        assert (st == 0).all()
        c_start, phi_start = solver.get_state()[:2]
        calc.set_surface_kinetics([{'species': 'CO2', 'rate': lambda phiM: 1e-4 * np.exp(-12.0 * (phiM + 0.6)),
                                    'stoichiometry': {'CO2': -1.0, 'CO': 1.0}}])
        calc._apply_surface_kinetics(solver, phis)
        cs, vs, es = solver.get_surface()
        state = {'surface_concentration': cs.copy(), 'surface_concentration_old': cs.copy(), 'flux': np.zeros((B, N)),
                 'current_density_old': np.zeros((B, N)), 'mix': np.full(B, 0.5), 'accuracy': np.full(B, np.inf),
                 'surface_pH': np.full(B, 7.0), 'surface_potential': vs, 'surface_efield': es,
                 'step_to_check': np.full(B, 5), 'active': np.ones(B, np.int32), 'failed': np.zeros(B, np.int32)}
        last = solver.scf_cycle(state, istep=5, max_iter=60, tau_scf=1e-5, faraday=unit_F, nel=[1, 1, 2, 2])
        assert last == 60
        assert list(state['active']) == [0, 0, 1] and not state['failed'].any()
        assert (state['surface_concentration'][2] == -1.0).all()
        assert state['mix'][2] == 0.5 * 0.9 and state['step_to_check'][2] == 46 and (state['mix'][:2] == 0.5).all()
        c_end, phi_end = solver.get_state()[:2]
        assert np.array_equal(c_end[2], c_start[2]) and np.array_equal(phi_end[2], phi_start[2])
        # converged lanes: flux = K c_s(CO2) at the fixed point, state on the device consistent with what was reported
        K = 1e-4 * np.exp(-12.0 * (phis[:2] + 0.6))
        L = (tp.nx - 1) * tp.dx
        assert np.allclose(state['surface_concentration'][:2, 2], 34.0 / (1.0 + K * L / tp.D[2]), rtol=1e-4)
        assert np.allclose(c_end[:2, 2, 0], state['surface_concentration'][:2, 2], rtol=1e-12)
        assert np.allclose(state['flux'][:2, 2], -K * state['surface_concentration_old'][:2, 2], rtol=1e-12)
        assert (state['accuracy'][:2] <= 1e-5).all()
    finally:
        solver.close()


def test_co2r_physical_example_matches_the_oracle_along_the_polarization_curve():
    """examples/co2r_physical_sweep.py (BASELINE configs[2] in the physical mode: 7 species, 5 stiff buffer reactions, steric K+,
    Stern layer, graded mesh, implicit Tafel kinetics) at test size, against the CPU oracle walking the same continuation."""
    import importlib.util
    from oracle import pnp_physical as PH
    spec = importlib.util.spec_from_file_location('co2r_physical_sweep', os.path.join(os.path.dirname(__file__), '..', 'examples',
                                                                                       'co2r_physical_sweep.py'))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    tp, phis = ex.build(4, 160)
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
    nu = [0.0] * 7
    nu[names.index('CO2')], nu[names.index('CO')], nu[names.index('OH-')] = -1.0, 1.0, 2.0
    for lane in (0, 3):
        c = np.repeat(cb[:, None], tp.nx, axis=1); phi = np.zeros(tp.nx)
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
        assert (np.abs(got - c) / scale).max() < 1e-6
        assert np.abs(np.array(d['system']['potential']) - phi).max() < 1e-7
    j = calc.kinetic_flux[:, names.index('CO')]
    assert j[0] > 0 and j[1] > 5 * j[0]           # Tafel region, then the CO2-transport plateau


def test_run_single_step_updates_tp_like_the_comsol_reader():
    tp = _physical_transport([-0.4, -0.8], mpb=False)
    tp.system['phiM'] = -0.7
    tp.species['CO2']['flux'] = -2e-5
    tp.species['CO']['flux'] = 2e-5
    calc = Calculator(transport=tp, calc='comsol')
    assert calc.run_single_step(label='it1')
    L = (tp.nx - 1) * tp.dx
    assert np.isclose(tp.species['CO']['surface_concentration'], 2e-5 * L / tp.D[3], rtol=1e-6)       # neutral product: linear profile
    assert len(tp.species['K+']['concentration']) == tp.nx and tp.species['K+']['surface_concentration'] > 100.0
    assert tp.system['surface_potential'] < 0 and len(tp.system['potential']) == tp.nx and 'activity_coefficient' in tp.system
    assert list(tp.descriptors['phiM']) == [-0.4, -0.8] and len(tp.alldata) == 2                          # the sweep set-up is untouched


def test_gpu_sweep_results_folder_matches_the_reference_reader_manifest(tmp_path):
    """f4: a real GPU sweep of the run.py system, written with results_io.save_all and walked through the accesses of the
    reference's plotting tool (tests/results_walk.py), shows the keys and types that the REFERENCE's own reader produced for such a
    folder (tests/golden/results_manifest.json, generated by tests/golden/make_results_manifest.py); profile lengths follow nx."""
    import importlib.util
    import json
    from catint_amd.results_io import save_all, read_all
    from tests import results_walk
    spec = importlib.util.spec_from_file_location('co2r_physical_sweep', os.path.join(os.path.dirname(__file__), '..', 'examples',
                                                                                       'co2r_physical_sweep.py'))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    tp, phis = ex.build(3, 96, phimin=-0.5, phimax=-0.7)
    calc = Calculator(transport=tp, calc='comsol')
    tp.newton = {'tol': 1e-9, 'maxit': 80}
    calc.set_surface_kinetics([{'species': 'CO2', 'rate': ex.tafel_rate(tp), 'stoichiometry': {'CO2': -1.0, 'CO': 1.0, 'OH-': 2.0}}])
    calc.run()
    assert np.all(calc.status == 0)
    folder = str(tmp_path / 'CO2R_results')
    save_all(tp, folder)

    class Bare(object):
        pass
    got = results_walk.walk(read_all(Bare(), folder, only=['alldata', 'species', 'system', 'xmesh', 'descriptors', 'electrode_reactions']))
    manifest = json.load(open(os.path.join(GOLDEN, 'results_manifest.json')))
    assert sorted(got) == sorted(manifest)
    for key, (kind, n) in manifest.items():
        assert got[key][0] == kind, (key, got[key], kind)
        if n is not None and kind in ('list', 'ndarray') and key != "descriptors['phiM']":
            assert got[key][1] == tp.nx, (key, got[key])
    # the CO partial current density of the reader's formula (comsol_reader.py:241-246): j * nel * F / nprod / 10
    names = list(tp.species.keys())
    j = calc.kinetic_flux[:, names.index('CO')]
    assert np.allclose([tp.alldata[i]['species']['CO']['electrode_current_density'] for i in range(3)], j * 2 * 96485.33289 / 1 / 10.)


def test_failed_lanes_recover_on_a_finer_ramp():
    """The per-lane convergence ladder of solve_physical (the reference's rerun-with-half-the-ramp loop, calculator.py:455-531): with a
    Newton budget too small for the default stages the far lanes fail; they walk 2, 4, ... x the stages as their own batch, converge,
    and end on the same state as a run that was given enough iterations in the first place."""
    phis = [-0.4, -0.9, -1.5, -1.9]
    ref_tp = _physical_transport(phis)
    ref = Calculator(transport=ref_tp, calc='comsol')
    ref.run()
    assert np.all(ref.status == 0) and not ref.retry_log
    tp = _physical_transport(phis)
    tp.newton = {'maxit': 4, 'dphi_stage': 0.5, 'min_stages': 8, 'retry_rungs': 4}
    calc = Calculator(transport=tp, calc='comsol')
    calc.run()
    assert calc.retry_log and calc.retry_log[0]['lanes'], calc.retry_log
    assert np.all(calc.status == 0), (calc.status, calc.retry_log)
    recovered = sorted(set(sum((r['recovered'] for r in calc.retry_log), [])))
    assert recovered == sorted(calc.retry_log[0]['lanes'])
    for i in range(len(phis)):
        a = np.array([tp.alldata[i]['species'][sp]['concentration'] for sp in tp.species])
        b = np.array([ref_tp.alldata[i]['species'][sp]['concentration'] for sp in ref_tp.species])
        assert np.abs(a - b).max() <= 1e-6 * np.abs(b).max()
        assert np.abs(np.array(tp.alldata[i]['system']['potential']) - np.array(ref_tp.alldata[i]['system']['potential'])).max() <= 1e-7
    # without the ladder the same budget leaves those lanes unconverged (and says so)
    tp0 = _physical_transport(phis)
    tp0.newton = {'maxit': 4, 'dphi_stage': 0.5, 'min_stages': 8, 'retry_rungs': 0}
    c0 = Calculator(transport=tp0, calc='comsol')
    c0.run()
    assert (c0.status != 0).any() and sorted(np.flatnonzero(c0.status != 0).tolist()) == sorted(calc.retry_log[0]['lanes'])
