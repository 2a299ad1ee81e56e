"""Host logic of the physical mode (no GPU): continuation path, surface-kinetics plumbing, graded mesh, units."""
import collections

import numpy as np
import pytest

from catint_amd.calculator import Calculator, CalculatorError
from catint_amd.host import graded_mesh
from catint_amd.transport import Transport


class FakeSolver(object):
    """Records what Calculator.solve_physical asks of the C-ABI wrapper; 'converges' unless told otherwise."""

    def __init__(self, fail_direct=False):
        self.calls = []
        self.fail_direct = fail_direct
        self.B = 0

    def set_batch(self, c0, pb, vz, flux):
        self.B = len(pb)
        self.calls.append(('set_batch', pb[:, 0].copy(), np.array(flux, float).copy()))

    def set_pb(self, pb, vz):
        self.calls.append(('set_pb', pb[:, 0].copy()))

    def set_flux(self, flux):
        self.calls.append(('set_flux', np.array(flux, float).copy()))

    def set_wall_kinetics(self, species, nu, k, alpha=None, saturation=None):
        self.law = (alpha, saturation)
        self.calls.append(('kinetics', list(species), np.array(nu, float).copy(), np.array(k, float).copy()))

    def solve_stationary(self, tol=0.0, maxit=0):
        n = sum(1 for c in self.calls if c[0] == 'solve')
        self.calls.append(('solve',))
        if self.fail_direct and n == 0:
            return np.ones(self.B, np.int32)
        return np.zeros(self.B, np.int32)

    def newton_iterations(self):
        return np.full(self.B, 4, np.int32)

    def step(self, n):
        self.calls.append(('step', n))

    def get_status(self):
        return np.zeros(self.B, np.int32)


def make_tp(phis, **system):
    species = collections.OrderedDict([('K+', {'bulk_concentration': 100.0, 'MPB_radius': 4.1e-10}),
                                       ('HCO3-', {'bulk_concentration': 100.0}), ('CO2', {'bulk_concentration': 34.0})])
    sysd = {'phiM': phis[0], 'boundary thickness': 4e-8, 'Stern capacitance': 20.0, 'phiPZC': 0.16}
    sysd.update(system)
    return Transport(species=species, system=sysd, nx=64, descriptors={'phiM': list(phis)})


def test_small_potentials_are_solved_directly():
    phis = np.array([0.0, 0.3, 0.5])
    tp = make_tp(phis)
    calc = Calculator(transport=tp, calc='comsol')
    s = FakeSolver()
    st = calc.solve_physical(s, np.zeros((3, 3 * tp.nx)), phis, np.zeros((3, 3)))
    assert np.all(st == 0)
    assert [c[0] for c in s.calls] == ['set_batch', 'solve'] and np.allclose(s.calls[0][1], phis)


def test_failed_direct_solve_restarts_on_the_continuation_path():
    phis = np.array([0.0, 0.3, 0.5])
    tp = make_tp(phis)
    calc = Calculator(transport=tp, calc='comsol')
    s = FakeSolver(fail_direct=True)
    flux = np.array([[0.0, 0.0, -1e-4]] * 3)
    calc.solve_physical(s, np.zeros((3, 3 * tp.nx)), phis, flux)
    stages = calc.continuation_stages
    assert stages == 4                                  # nramp: span 0.34 V / 0.4 V = 1 stage < 4 (round 3: 8 stages of at most 0.2 V)
    batches = [c for c in s.calls if c[0] == 'set_batch']
    assert len(batches) == 2                            # direct attempt, then the restart from the bulk state
    assert np.allclose(batches[1][1], 0.16 + (phis - 0.16) / stages) and np.allclose(batches[1][2], flux / stages)
    pbs = [c[1] for c in s.calls if c[0] == 'set_pb']
    assert len(pbs) == stages - 1 and np.allclose(pbs[-1], phis)          # ends exactly at phiM with the full flux
    assert np.allclose([c for c in s.calls if c[0] == 'set_flux'][-1][1], flux)


def test_far_potentials_and_kinetics_walk_from_phi_pzc_in_400mV_stages():
    phis = np.linspace(-0.5, -2.0, 4)
    tp = make_tp(phis)
    calc = Calculator(transport=tp, calc='comsol')
    rate = lambda phiM: 1e-3 * np.exp(-10.0 * phiM)
    calc.set_surface_kinetics([{'species': 'CO2', 'rate': rate, 'stoichiometry': {'CO2': -1.0, 'HCO3-': 0.5}}])
    s = FakeSolver()
    calc.solve_physical(s, np.zeros((4, 3 * tp.nx)), phis, np.zeros((4, 3)))
    assert calc.continuation_stages == int(np.ceil(2.16 / 0.4))
    tp.newton = {'dphi_stage': 0.2}                     # (the width is an input: tp.newton['dphi_stage'])
    s2 = FakeSolver()
    calc.solve_physical(s2, np.zeros((4, 3 * tp.nx)), phis, np.zeros((4, 3)))
    assert calc.continuation_stages == int(np.ceil(2.16 / 0.2))
    tp.newton = {}
    calc.solve_physical(FakeSolver(), np.zeros((4, 3 * tp.nx)), phis, np.zeros((4, 3)))
    assert [c[0] for c in s.calls].count('solve') == calc.continuation_stages
    kin = [c for c in s.calls if c[0] == 'kinetics']
    assert len(kin) == calc.continuation_stages and kin[0][1] == [2]
    assert np.allclose(kin[0][2], [[0.0, 0.5, -1.0]])
    w = 1.0 / calc.continuation_stages
    assert np.allclose(kin[0][3][:, 0], rate(0.16 + (phis - 0.16) * w))           # rate constants at the STAGE potential
    assert np.allclose(kin[-1][3][:, 0], rate(phis))
    cs = np.array([[100.0, 100.0, 2.0]] * 4)
    f = calc.surface_kinetic_fluxes(cs, phis)
    assert np.allclose(f[:, 2], -rate(phis) * 2.0) and np.allclose(f[:, 1], 0.5 * rate(phis) * 2.0) and np.all(f[:, 0] == 0)


def test_rate_law_beyond_first_order_reaches_the_solver_and_the_flux_formula():
    """'alpha' (Butler-Volmer in the Stern-layer drop) and 'saturation' (Langmuir) of set_surface_kinetics: handed to the C-ABI
    wrapper as pnp_set_wall_rate_law arrays, and evaluated by surface_kinetic_fluxes as K c/(1 + K_sat c) exp(alpha (phiM - phi_0))
    (the reference's user-defined flux form, docs/source/topics/flux_definition.rst:90-160)."""
    phis = np.array([0.1, 0.2])
    tp = make_tp(phis)
    calc = Calculator(transport=tp, calc='comsol')
    kin = [{'species': 'CO2', 'rate': 2e-3, 'alpha': -9.0, 'saturation': 0.25, 'stoichiometry': {'CO2': -1.0, 'HCO3-': 1.0}},
           {'species': None, 'rate': 1e-6, 'stoichiometry': {'K+': 1.0}}]
    calc.set_surface_kinetics(kin)
    s = FakeSolver(); s.B = 2
    calc._apply_surface_kinetics(s, phis)
    assert s.law == ([-9.0, 0.0], [0.25, 0.0]) and s.calls[-1][1] == [2, -1]
    cs = np.array([[100.0, 100.0, 8.0], [100.0, 100.0, -1.0]])
    v0 = np.array([0.05, 0.02])
    f = calc.surface_kinetic_fluxes(cs, phis, vsurf=v0)
    g0 = 8.0 / (1.0 + 0.25 * 8.0) * np.exp(-9.0 * (0.1 - 0.05))
    assert np.isclose(f[0, 2], -2e-3 * g0, rtol=1e-15) and np.isclose(f[0, 1], 2e-3 * g0, rtol=1e-15) and f[0, 0] == 1e-6
    assert calc.surface_kinetic_fluxes(cs, phis, clip=True, vsurf=v0)[1, 2] == 0.0       # clipped surface concentration
    with pytest.raises(CalculatorError, match='surface potential'):
        calc.surface_kinetic_fluxes(cs, phis)
    with pytest.raises(CalculatorError, match='saturation'):
        calc.set_surface_kinetics([{'species': None, 'rate': 1.0, 'saturation': 0.1, 'stoichiometry': {}}])
    # first-order tables keep the old call shape (no rate-law arrays)
    calc.set_surface_kinetics([{'species': 'CO2', 'rate': 1.0, 'stoichiometry': {'CO2': -1.0}}])
    calc._apply_surface_kinetics(s, phis)
    assert s.law == (None, None)


def test_warm_and_time_dependent_paths():
    phis = np.array([-1.0, -1.2])
    tp = make_tp(phis)
    calc = Calculator(transport=tp, calc='comsol')
    s = FakeSolver(); s.B = 2
    calc.solve_physical(s, None, phis, np.ones((2, 3)), warm=True)
    assert [c[0] for c in s.calls] == ['set_flux', 'solve']
    tp2 = make_tp(phis)
    calc2 = Calculator(transport=tp2, calc='comsol', mode='time-dependent', dt=1e-6, tmax=1e-5)
    s2 = FakeSolver()
    calc2.solve_physical(s2, np.zeros((2, 3 * tp2.nx)), phis, np.zeros((2, 3)))
    assert [c[0] for c in s2.calls] == ['set_batch', 'step'] and s2.calls[1][1] == tp2.nt - 1


def test_graded_mesh_and_guards():
    x = graded_mesh(8e-5, 5e-11, 384)
    h = np.diff(x)
    assert len(x) == 384 and x[0] == 0.0 and x[-1] == 8e-5 and np.isclose(h[0], 5e-11)
    assert np.all(h > 0) and np.allclose(h[1:-1] / h[:-2], h[1] / h[0], rtol=1e-6)      # constant growth ratio
    assert np.allclose(graded_mesh(1.0, 0.5, 3), [0.0, 0.5, 1.0])                        # degenerate: uniform
    tp = make_tp([0.0])
    tp.set_graded_mesh(tp.debye_length / 10)
    assert not tp.mesh_uniform and np.isclose(tp.dx, tp.debye_length / 10) and len(tp.xmesh) == tp.nx
    with pytest.raises(CalculatorError):
        Calculator(transport=tp, calc='Crank-Nicolson', dt=1e-10, tmax=1e-9)           # FD integrators need a uniform mesh
    with pytest.raises(CalculatorError):
        Calculator(transport=make_tp([0.0]), calc='FTCS', dt=1e-10, tmax=1e-9).set_surface_kinetics([])


def test_booth_stern_field():
    from catint_amd.host import booth_permittivity, booth_stern_field
    assert booth_permittivity(5e6, 78.36) == 78.36
    assert 1.33 ** 2 < booth_permittivity(5e9, 78.36) < booth_permittivity(5e8, 78.36) < 78.36
    for e_out in (1e5, -3e8, 2e9):
        E, eps = booth_stern_field(e_out, 78.36)
        assert np.sign(E) == np.sign(e_out) and abs(E) >= abs(e_out)
        assert abs(E - e_out * 78.36 / booth_permittivity(E, 78.36)) <= 1e-9 * abs(E)       # displacement continuity
        assert np.isclose(eps, booth_permittivity(E, 78.36))
    assert booth_stern_field(0.0, 78.36) == (0.0, 78.36)


class LadderSolver(FakeSolver):
    """A fake whose lanes `stuck` never converge while the handle walks fewer than `need` stages; set_lanes / set_lane_mask record
    the patching of recovered lanes and the confirming solve restricted to them."""

    def __init__(self, B, nx, N, stuck=(), need=10 ** 9, confirm_fails=()):
        FakeSolver.__init__(self)
        self.B, self.nx, self.N, self.stuck, self.need = B, nx, N, list(stuck), need
        self.stages = 0
        self.c = np.zeros((B, N, nx)); self.phi = np.zeros((B, nx))
        self.mask = None
        self.confirm_fails = list(confirm_fails)
        self.grid = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def set_batch(self, c0, pb, vz, flux):
        FakeSolver.set_batch(self, c0, pb, vz, flux)
        self.c = np.array(c0, float).reshape(self.B, self.N, self.nx).copy()

    def set_lanes(self, lanes, c, phi=None):
        self.calls.append(('set_lanes', list(lanes)))
        self.c[np.asarray(lanes)] = np.asarray(c, float).reshape(len(lanes), self.N, self.nx)
        if phi is not None:
            self.phi[np.asarray(lanes)] = phi

    def set_lane_mask(self, mask=None):
        self.mask = None if mask is None else np.asarray(mask).copy()
        self.calls.append(('mask', None if mask is None else list(np.flatnonzero(mask))))

    def get_state(self):
        return self.c.copy(), self.phi.copy(), None, None

    def solve_stationary(self, tol=0.0, maxit=0):
        self.calls.append(('solve',))
        st = np.zeros(self.B, np.int32)
        if self.mask is not None:            # confirming solve of patched lanes: their state stays, the verdict is the handle's
            st[self.confirm_fails] = 1
            return st
        self.stages += 1
        if self.stages < self.need:
            st[self.stuck] = 1
        self.c[:] = 7.0 + self.stages           # a recognisable state: the stage count
        self.phi[:] = -self.stages
        return st


def test_failed_lanes_walk_finer_ramps_and_are_patched_back(monkeypatch):
    """The reference's convergence ladder (calculator.py:455-531: rerun with the ramp interval halved, the boundary mesh refined, ...)
    per lane: only the failed lanes go again, as their own batch, with 2, 4, 8 x the stages; what converges replaces those lanes'
    state in the main handle (pnp_set_lanes: nothing else moves) and is confirmed by a solve of the main handle restricted to them."""
    phis = np.linspace(-0.5, -2.0, 4)
    tp = make_tp(phis)
    tp.newton = {'retry_rungs': 3, 'retry_mesh_rungs': 0, 'dphi_stage': 0.2}      # (11 stages, as the ladder's numbers below assume)
    calc = Calculator(transport=tp, calc='comsol')
    main = LadderSolver(4, tp.nx, 3, stuck=[1, 3])
    subs = []

    def fake_sub(B, **kw):
        # rung 1 (22 stages): both failed lanes stay stuck; rung 2 (44): the first of the two converges; rung 3 (88): the other one
        k = len(subs)
        s = LadderSolver(B, tp.nx, 3, stuck=[[0, 1], [1], [0]][k], need=[10 ** 9, 10 ** 9, 88][k])
        subs.append(s)
        return s
    monkeypatch.setattr(calc, '_physical_solver', fake_sub)
    st = calc.solve_physical(main, np.ones((4, 3 * tp.nx)), phis, np.zeros((4, 3)))
    assert calc.continuation_stages == 11
    assert [r['stages'] for r in calc.retry_log] == [22, 44, 88] and [r['lanes'] for r in calc.retry_log] == [[1, 3], [1, 3], [3]]
    assert [r['recovered'] for r in calc.retry_log] == [[], [1], [3]]
    assert list(st) == [0, 0, 0, 0]
    # the recovered lanes carry the sub-batches' states, the others the main handle's; only those lanes travelled
    assert (main.c[1] == 7.0 + 44).all() and (main.phi[1] == -44).all() and (main.c[3] == 7.0 + 88).all() and (main.c[0] == 7.0 + 11).all()
    assert [c[1] for c in main.calls if c[0] == 'set_lanes'] == [[1], [3]]
    assert [c[1] for c in main.calls if c[0] == 'mask'] == [[1], None, [3], None]
    assert not any(c[0] == 'set_potential' for c in main.calls)


def test_mesh_rung_and_the_main_handles_verdict(monkeypatch):
    """After the ramp rungs the boundary mesh is refined by 1.5 per rung (grid_factor_bound *= 1.5 in the reference's ladder): the
    sub-batch is built on the finer graded mesh, its solution interpolated onto the batch's mesh, and what the main handle says about
    the patched lane stands -- a lane whose confirming solve fails stays failed."""
    phis = np.linspace(-0.5, -2.0, 3)
    tp = make_tp(phis)
    tp.set_graded_mesh(tp.xmesh[1] / 50.0)
    tp.newton = {'retry_rungs': 1, 'retry_mesh_rungs': 2, 'dphi_stage': 0.2}
    calc = Calculator(transport=tp, calc='comsol')
    main = LadderSolver(3, tp.nx, 3, stuck=[0, 2], confirm_fails=[2])
    grids = []

    def fake_sub(B, xmesh=None, **kw):
        grids.append(None if xmesh is None else np.array(xmesh))
        return LadderSolver(B, tp.nx, 3, stuck=[] if xmesh is not None else [0, 1])       # converges only on a refined mesh
    monkeypatch.setattr(calc, '_physical_solver', fake_sub)
    st = calc.solve_physical(main, np.ones((3, 3 * tp.nx)), phis, np.zeros((3, 3)))
    assert [r['mesh_refined'] for r in calc.retry_log] == [False, True, True]
    h0 = tp.xmesh[1] - tp.xmesh[0]
    assert grids[0] is None and np.isclose(grids[1][1] - grids[1][0], h0 / 1.5) and np.isclose(grids[2][1] - grids[2][0], h0 / 2.25)
    assert grids[1][-1] == tp.xmesh[-1] and len(grids[1]) == tp.nx
    assert calc.retry_log[1]['recovered'] == [0] and calc.retry_log[2]['lanes'] == [2] and calc.retry_log[2]['recovered'] == []
    assert list(st) == [0, 0, 1]


def test_per_lane_rate_constants_follow_the_failed_lanes_through_the_ladder(monkeypatch):
    """A rate given as an array over the full batch (or a callable returning one) is cut down to the sub-batch of failed lanes."""
    phis = np.linspace(-0.5, -2.0, 4)
    tp = make_tp(phis)
    tp.newton = {'retry_rungs': 1, 'retry_mesh_rungs': 0}
    calc = Calculator(transport=tp, calc='comsol')
    k_lane = np.array([1.0, 2.0, 3.0, 4.0])
    calc.set_surface_kinetics([{'species': 'CO2', 'rate': k_lane, 'stoichiometry': {'CO2': -1.0}},
                               {'species': 'CO2', 'rate': lambda v: 10.0 * k_lane[:len(v)] if len(v) == 4 else 10.0 * k_lane, 'stoichiometry': {'CO2': -1.0}}])
    main = LadderSolver(4, tp.nx, 3, stuck=[1, 3])
    subs = []

    def fake_sub(B, **kw):
        subs.append(LadderSolver(B, tp.nx, 3))
        return subs[-1]
    monkeypatch.setattr(calc, '_physical_solver', fake_sub)
    st = calc.solve_physical(main, np.ones((4, 3 * tp.nx)), phis, np.zeros((4, 3)))
    assert list(st) == [0, 0, 0, 0]
    kin = [c for c in subs[0].calls if c[0] == 'kinetics']
    assert kin and all(np.allclose(c[3][:, 0], [2.0, 4.0]) and np.allclose(c[3][:, 1], [20.0, 40.0]) for c in kin)


def test_flux_sign_repair_of_run_single_step():
    """err_in_flux / change_flux_sign of the reference's run_single_step (calculator.py:415-446)."""
    import collections
    species = collections.OrderedDict([('K+', {'bulk_concentration': 100.0}), ('HCO3-', {'bulk_concentration': 100.0}),
                                       ('CO2', {'bulk_concentration': 34.0}), ('CO', {'bulk_concentration': 0.0, 'flux': 1e-5}), ('OH-', {'bulk_concentration': 1e-4})])
    er = {'CO': {'reaction': 'CO2 + H2O + 2 e- -> CO + 2 OH-', 'nel': 2}}
    tp = Transport(species=species, electrode_reactions=er, system={'phiM': -0.8, 'boundary thickness': 4e-8, 'exclude species': ['H2O', 'e-']},
                   nx=32, descriptors={'phiM': [-0.8]})
    calc = Calculator(transport=tp, calc='comsol')
    assert tp.electrode_reactions['CO']['reaction'][0][0] == 'CO2' and 'CO' in tp.electrode_reactions['CO']['reaction'][1]
    # the flux closure of Transport (transport.py:929-1095) has filled in the educt and the co-product: -1e-5, +1e-5, +2e-5
    assert np.allclose([tp.species[sp]['flux'] for sp in ('CO2', 'CO', 'OH-')], [-1e-5, 1e-5, 2e-5], rtol=1e-12)
    assert not calc.flux_sign_error()
    tp.species['CO2']['flux'] = 1e-5                      # an educt that is produced: the reference flips every electrode flux
    assert calc.flux_sign_error()
    calc.change_flux_sign()
    assert np.allclose([tp.species[sp]['flux'] for sp in ('CO2', 'CO', 'OH-')], [-1e-5, -1e-5, -2e-5], rtol=1e-12)
    assert tp.species['K+'].get('flux', 0.0) in (0.0, '0.0', 0)           # species outside the electrode reactions are not touched
    tp.species['CO2']['flux'] = 0.0                       # zero fluxes carry no sign
    assert not calc.flux_sign_error()


def test_roughness_factor_scales_every_wall_flux_like_the_reference_model():
    """j_i = RF*flux_factor*flux_i (reference catint/comsol_model.py:1134): the fluxes and kinetic rate constants that reach the
    solver carry system['RF']; expected numbers from the j1..j7 / RF strings the reference's model generator emits for the run.py
    system with one numeric flux and RF = 2 (tests/golden/comsol_model_runpy.json, case co2r_numeric_flux_rf2)."""
    import json
    import os
    import re
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    par = json.load(open(os.path.join(gold, 'comsol_model_runpy.json')))['co2r_numeric_flux_rf2']['pairs']['param']
    rf = float(par['RF'])
    assert rf == 2.0
    expected = [rf * float(re.match(r'RF\*flux_factor\*(.+)\[mol/m\^2/s\]$', par['j%d' % (i + 1)]).group(1)) for i in range(7)]
    case = next(c for c in json.load(open(os.path.join(gold, 'transport_cases.json'))) if c['input']['name'] == 'co2r_numeric_flux')['input']
    species = collections.OrderedDict((name, dict(d)) for name, d in case['species'])
    system = dict(case['system'], RF=2.0)
    if system.get('active site density') == 'run.py':
        system['active site density'] = 9.61e-05 / 6.02214076e23 * (1e10) ** 2
    tp = Transport(species=species, electrode_reactions=case.get('electrode_reactions'), electrolyte_reactions=case.get('electrolyte_reactions'),
                   system=system, nx=case['nx'], descriptors={'phiM': [system['phiM']]})
    calc = Calculator(transport=tp, calc='comsol')
    assert calc.RF == 2.0
    s = FakeSolver()
    flux = np.repeat(tp.flux_bound[None, :, 0], 1, axis=0)
    calc.solve_physical(s, np.zeros((1, tp.nspecies * tp.nx)), np.array([system['phiM']]), flux, nramp=1)
    got = next(c for c in s.calls if c[0] == 'set_batch')[2][0]
    assert np.allclose(got, expected, rtol=1e-12, atol=0.0), (got, expected)
    assert np.abs(got).max() > 0
    # warm path and kinetic rate constants
    s = FakeSolver()
    calc.solve_physical(s, None, np.array([system['phiM']]), flux, warm=True)
    assert np.allclose(next(c for c in s.calls if c[0] == 'set_flux')[1][0], expected)
    calc.set_surface_kinetics([{'species': 'CO2', 'rate': 3.0e-3, 'stoichiometry': {'CO2': -1.0, 'CO': 1.0}}])
    s = FakeSolver()
    calc._apply_surface_kinetics(s, np.array([system['phiM']]))
    assert np.allclose(next(c for c in s.calls if c[0] == 'kinetics')[3], 2.0 * 3.0e-3)


def test_convection_velocity_is_carried_and_expressions_are_refused():
    """system['flow rate'] goes to the COMSOL model's convection velocity (comsol_model.py:901-903): a number reaches the solver
    (pnp_set_convection), a COMSOL expression string is refused instead of being ignored."""
    calc = Calculator(transport=make_tp([0.1], **{'flow rate': 1e-3}), calc='comsol')
    assert calc.velocity == 1e-3
    assert Calculator(transport=make_tp([0.1], **{'flow rate': '2.5e-4'}), calc='comsol').velocity == 2.5e-4
    assert Calculator(transport=make_tp([0.1], **{'flow rate': 0.0}), calc='comsol').velocity == 0.0      # a zero velocity is no convection
    with pytest.raises(CalculatorError, match='flow rate'):
        Calculator(transport=make_tp([0.1], **{'flow rate': '1e-3*x/L_cell'}), calc='comsol')
