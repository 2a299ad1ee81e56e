"""GPU: the method-of-lines integrator on the device (pnp_integrate_dopri5) against
  * the reference's own calc='dopri5' trajectory (tests/golden/dopri5_dd_n2_nx50.npz, written by the reference + scipy),
  * oracle/dopri5.py (pinned bit for bit against scipy.integrate.ode('dopri5'), tests/test_ode_oracle.py) driving the SAME device
    right-hand side from the host: same accepted / rejected step sequence and evaluation count, trajectories equal to rounding (the
    device sums its error norms in tree order, the oracle left to right -- step sizes differ in the last bits),
  * itself: lanes of a batch are independent (bitwise), failure exits freeze a lane without touching the others."""
import os

import numpy as np
import pytest

from oracle import pnp_ref as R
from oracle.dopri5 import Dopri5
from oracle.dop853 import Dop853
from catint_amd.host import solver_from_problem
from catint_amd.calculator import Calculator, make_itout
from tests.test_host_transport import transport_from_fixture

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def golden_problem(name='dopri5_dd_n2_nx50', interval=1):
    """The reference's dopri5 fixture; interval > 1 stretches the output interval (fixture: 1e-11 s, one step each -- the
    explicit stability limit dx^2/(2 D) is 2.5e-10 s) so that the controller has work to do inside every call."""
    d = np.load(os.path.join(GOLDEN, name + '.npz'))
    p, c0, nt, itout, method = R.problem_from_golden(d)
    p.dt = interval * p.dt
    return d, p, c0, nt, itout


def oracle_run(s, lane_state, B, lane, nt, order=5, **kw):
    """oracle/dopri5.py (order=8: oracle/dop853.py without scipy's redundant evaluation) on lane `lane` with the device right-hand side
    (all lanes evaluated, one kept)."""
    state = np.array(lane_state, float)

    def f(t, y):
        state[lane] = y
        return s.mol_rhs(state)[lane]
    o = (Dopri5(f, **kw) if order == 5 else Dop853(f, recompute_k1=False, **kw)).set_initial_value(lane_state[lane].copy())
    dt = s.dt_ode
    out = []
    for _ in range(nt):
        if not o.successful():
            break
        out.append(o.integrate(o.t + dt).copy())
    return o, out


def test_device_dopri5_reproduces_the_reference_trajectory():
    d, p, c0, nt, itout = golden_problem()
    tp = transport_from_fixture(d)
    tp.c0 = d['c0'].copy(); tp.flux_bound = d['flux_bound'].copy(); tp.system['vzeta'] = float(d['vzeta'])
    ntout = next(n for n in range(1, 8) if make_itout(int(d['nt']), n) == [int(i) for i in d['itout']])
    outs = []
    for on_device in (True, False):
        calc = Calculator(transport=tp, calc='dopri5', dt=float(d['dt']), tmax=float(d['tmax']), ntout=ntout)
        calc.ode_on_device = on_device
        outs.append(calc.integrate_pnp(tp.dx, tp.nx, tp.dt, tp.nt, tp.ntout, 'dopri5'))
        if on_device:
            assert calc.ode_idid == 1 and calc.ode_stats[1] >= tp.nt and calc.ode_stats[3] == 2 * tp.nt + 6 * calc.ode_stats[0]
    dev, host = outs
    assert len(dev) == len(d['cout']) == len(host)
    for a, b, c in zip(dev, host, d['cout']):
        assert relerr(a, b) < 1e-10          # same integrator, same right-hand side: device loop vs scipy loop
        assert relerr(a, c) < 1e-7           # the reference's own run (its RHS sums in another order; rtol of the integrator 1e-6)


@pytest.mark.parametrize('kw', [{}, {'rtol': 1e-9, 'atol': 1e-10}, {'first_step': 1e-13, 'max_step': 4e-12, 'safety': 0.8, 'ifactor': 4.0,
                                                                       'dfactor': 0.3, 'beta': 0.08}, {'beta': -1.0, 'rtol': 1e-4}])
def test_same_step_sequence_as_the_pinned_oracle(kw):
    d, p, c0, nt, itout = golden_problem(interval=1 if 'max_step' in kw else 100)
    nt = 12 if 'max_step' in kw else 6
    with solver_from_problem(p, 'FTCS', batch_capacity=1) as s:
        s.set_batch(c0[None, :], p.pb[None, :], [p.vzeta], p.flux_bound[None, :])
        o, ref = oracle_run(s, c0[None, :].copy(), 1, 0, nt, nsteps=10000, **kw)
        s.set_batch(c0[None, :], p.pb[None, :], [p.vzeta], p.flux_bound[None, :])
        cout, idid, stats, t_end = s.integrate_dopri5(nt, list(range(nt)), nsteps=10000, **kw)
        c_end = s.get_state()[0]
    assert idid[0] == 1 and o.idid == 1
    assert 'max_step' in kw or len(o.log) >= 3 * nt      # several steps per call
    acc = sum(1 for e in o.log if e[3])
    assert list(stats[0]) == [len(o.log), acc, stats[0][2], o.nfcn, nt - 1]
    assert stats[0][2] <= len(o.log) - acc              # DOPRI5 does not count rejections before the first accepted step of a call
    assert abs(t_end[0] - o.t) <= 1e-13 * o.t
    for n in range(nt):
        assert relerr(cout[n, 0], ref[n]) < 1e-12
    assert np.array_equal(c_end[0].reshape(-1), cout[-1, 0])      # the state on the device is the integrated one


def test_rejected_steps_are_exercised_and_match():
    """A loose first step makes DOPRI5 reject; the device takes the same decisions as the oracle."""
    d, p, c0, nt, itout = golden_problem(interval=200)      # intervals of 2 ns: eight times the explicit stability limit
    kw = {'first_step': p.dt, 'rtol': 1e-8}
    with solver_from_problem(p, 'FTCS', batch_capacity=1) as s:
        s.set_batch(c0[None, :], p.pb[None, :], [p.vzeta], p.flux_bound[None, :])
        o, ref = oracle_run(s, c0[None, :].copy(), 1, 0, 3, nsteps=10000, **kw)
        s.set_batch(c0[None, :], p.pb[None, :], [p.vzeta], p.flux_bound[None, :])
        cout, idid, stats, _ = s.integrate_dopri5(3, [0, 1, 2], nsteps=10000, **kw)
    rej = sum(1 for e in o.log if not e[3])
    assert rej >= 1 and stats[0][0] == len(o.log) and stats[0][1] == len(o.log) - rej
    assert relerr(cout[2, 0], ref[2]) < 1e-12


def test_lanes_are_independent_and_adapt_separately():
    d, p, c0, nt, itout = golden_problem(interval=100)
    B, nt = 5, 4
    rng = np.random.default_rng(3)
    cs = np.stack([c0 * rng.uniform(0.5, 1.5) for _ in range(B)])
    pb = np.stack([p.pb] * B); pb[:, 0] = np.linspace(-0.02, 0.05, B)       # wall potential differs per lane
    flux = np.stack([p.flux_bound] * B)
    with solver_from_problem(p, 'FTCS', batch_capacity=B) as s:
        s.set_batch(cs, pb, [p.vzeta] * B, flux)
        cout, idid, stats, t_end = s.integrate_dopri5(nt, [nt - 1], nsteps=10000)
        assert (idid == 1).all() and len(set(stats[:, 0])) > 1               # different step counts in one batch
        for b in (0, 3):
            s.set_batch(cs, pb, [p.vzeta] * B, flux)
            o, ref = oracle_run(s, cs.copy(), B, b, nt, nsteps=10000)
            assert stats[b][0] == len(o.log) and relerr(cout[0, b], ref[-1]) < 1e-12
    with solver_from_problem(p, 'FTCS', batch_capacity=1) as s1:              # lane 3 alone: bitwise the same
        s1.set_batch(cs[3:4], pb[3:4], [p.vzeta], flux[3:4])
        c1, _, st1, _ = s1.integrate_dopri5(nt, [nt - 1], nsteps=10000)
    assert np.array_equal(c1[0, 0], cout[0, 3]) and np.array_equal(st1[0], stats[3])


def test_failure_exits_freeze_the_lane_and_leave_the_others_alone():
    d, p, c0, nt, itout = golden_problem(interval=100)
    B, nt = 2, 4
    cs = np.stack([c0, c0 * 1.2])
    pb = np.stack([p.pb] * B)
    flux = np.stack([p.flux_bound] * B)
    with solver_from_problem(p, 'FTCS', batch_capacity=B) as s:
        s.set_batch(cs, pb, [p.vzeta] * B, flux)
        ok, _, st_ok, _ = s.integrate_dopri5(nt, list(range(nt)), nsteps=10000)
        need = int(st_ok[:, 0].max())                      # attempted steps of the whole run; the first interval needs most of them
        s.set_batch(cs, pb, [p.vzeta] * B, flux)
        o, ref = oracle_run(s, cs.copy(), B, 0, nt, nsteps=3)
        s.set_batch(cs, pb, [p.vzeta] * B, flux)
        cout, idid, stats, t_end = s.integrate_dopri5(nt, list(range(nt)), nsteps=3)
    assert need > 8 and o.idid == -2
    assert list(idid) == [-2, -2] and (stats[:, 4] == 0).all() and (stats[:, 0] == 4).all()     # NMAX + 1 attempted steps, first call
    assert abs(t_end[0] - o.t) <= 1e-13 * o.t and relerr(cout[0, 0], ref[0]) < 1e-12
    assert np.array_equal(cout[0], cout[3])                # frozen after the failed call (the reference's loop stops there)
    # stiffness exit: check every accepted step (nstiff = 1) at a tolerance that keeps the steps at the stability limit
    with solver_from_problem(p, 'FTCS', batch_capacity=1) as s:
        s.set_batch(c0[None, :], p.pb[None, :], [p.vzeta], p.flux_bound[None, :])
        o, ref = oracle_run(s, c0[None, :].copy(), 1, 0, 40, nsteps=10000, nstiff=1, rtol=1e-3, atol=1e-6)
        s.set_batch(c0[None, :], p.pb[None, :], [p.vzeta], p.flux_bound[None, :])
        cout, idid, stats, t_end = s.integrate_dopri5(40, list(range(40)), nsteps=10000, nstiff=1, rtol=1e-3, atol=1e-6)
    assert idid[0] == o.idid and stats[0][0] == len(o.log) and abs(t_end[0] - o.t) <= 1e-13 * o.t
    if o.idid == -4:
        assert stats[0][4] == len(ref) - 1 and relerr(cout[len(ref) - 1, 0], ref[-1]) < 1e-10


def test_grids_beyond_one_wave_and_argument_checks():
    from catint_amd import _capi
    from catint_amd.synthetic import make_batch
    N, nx, B = 2, 1500, 2
    prob, c0, pb, vz, flux = make_batch(B, N, nx, seed=4, dt_factor=1e-4)
    with solver_from_problem(prob, 'FTCS', batch_capacity=B) as s:
        s.set_batch(c0, pb, vz, flux)
        c_start = s.get_state()[0].reshape(B, -1)
        o, ref = oracle_run(s, c_start.copy(), B, 1, 2, nsteps=10000)
        s.set_batch(c0, pb, vz, flux)
        cout, idid, stats, _ = s.integrate_dopri5(2, [1], nsteps=10000)
        assert idid[1] == 1 and stats[1][0] == len(o.log) and relerr(cout[0, 1], ref[-1]) < 1e-12
        with pytest.raises(_capi.PnpError, match='itout'):
            s.integrate_dopri5(2, [1, 0])
        with pytest.raises(_capi.PnpError, match='negative'):
            s.integrate_dopri5(2, [1], rtol=-1.0)


def test_descriptor_sweep_with_the_integrator_on_the_device():
    """Calculator.run() with calc='dopri5': every descriptor point is a lane with its own adaptive DOPRI5; per lane equal to the pinned
    oracle integrating the reference's ode_func restatement (oracle/pnp_ref.py mol_rhs) on the CPU; the potential reported is the one
    of the state returned (the last right-hand side of the run is k7 = f(final state), calculator_old.py:816-818)."""
    import collections
    from catint_amd.transport import Transport
    phis = list(np.linspace(-0.03, 0.03, 5))
    species = collections.OrderedDict([('K+', {'bulk_concentration': 30.0}), ('Cl-', {'bulk_concentration': 10.0}),
                                       ('HCO3-', {'bulk_concentration': 20.0})])
    tp = Transport(species=species, system={'phiM': 0.0, 'boundary thickness': 2e-8}, nx=48,
                   pb_bound={'potential': {'wall': 'phiM', 'bulk': 0.0}}, descriptors={'phiM': phis})
    calc = Calculator(transport=tp, calc='dopri5', dt=2e-10, tmax=1.2e-9, ntout=2)
    cout = calc.run()
    assert (calc.status == 0).all() and (calc.ode_idid == 1).all() and len(set(calc.ode_stats[:, 0])) > 1
    out = [n for n in tp.itout if n < tp.nt]
    assert cout.shape == (len(out), len(phis), tp.nspecies * tp.nx)
    for i in (0, 3):
        p = R.Problem(D=tp.D, charges=tp.charges, beta=tp.beta, eps=tp.eps, dx=tp.dx, nx=tp.nx, dt=tp.dt,
                      pb=np.array([phis[i], 0.0, np.nan, np.nan]), vzeta=phis[i], flux_bound=np.zeros(3))
        o = Dopri5(lambda t, y: R.mol_rhs(y, p, solver='banded'), nsteps=10000).set_initial_value(tp.c0.copy())
        ref = [o.integrate(o.t + tp.dt).copy() for _ in range(tp.nt)]
        assert calc.ode_stats[i][0] == len(o.log)
        for j, n in enumerate(out):
            assert relerr(cout[j, i], ref[n]) < 1e-9
        v = R.poisson(ref[-1].reshape(tp.nspecies, tp.nx), p, solver='banded')[0]
        if out[-1] == tp.nt - 1:
            assert np.abs(tp.alldata[i]['system']['potential'] - v).max() / max(np.abs(v).max(), 1e-3) < 1e-9


@pytest.mark.parametrize('kw', [{}, {'rtol': 1e-9, 'atol': 1e-10}, {'first_step': 1e-11, 'max_step': 4e-10, 'safety': 0.8, 'ifactor': 4.0,
                                                                       'dfactor': 0.4, 'beta': 0.08}])
def test_dop853_same_step_sequence_as_the_pinned_oracle(kw):
    """pnp_integrate_dop853 against oracle/dop853.py (pinned bit for bit against scipy.integrate.ode('dop853')) driving the same device
    right-hand side: same attempted / accepted / rejected steps and evaluation count, trajectories equal to rounding."""
    d, p, c0, nt, itout = golden_problem(interval=300)      # 3 ns intervals: the 8th-order method takes long steps
    nt = 5
    with solver_from_problem(p, 'FTCS', batch_capacity=1) as s:
        s.set_batch(c0[None, :], p.pb[None, :], [p.vzeta], p.flux_bound[None, :])
        o, ref = oracle_run(s, c0[None, :].copy(), 1, 0, nt, order=8, nsteps=10000, **kw)
        s.set_batch(c0[None, :], p.pb[None, :], [p.vzeta], p.flux_bound[None, :])
        cout, idid, stats, t_end = s.integrate_dop853(nt, list(range(nt)), nsteps=10000, **kw)
        c_end = s.get_state()[0]
    assert idid[0] == 1 and o.idid == 1 and len(o.log) > 2 * nt
    acc = sum(1 for e in o.log if e[3])
    assert list(stats[0][:2]) == [len(o.log), acc] and stats[0][3] == o.nfcn and stats[0][4] == nt - 1
    assert stats[0][3] == 2 * nt + 11 * len(o.log) + acc
    assert abs(t_end[0] - o.t) <= 1e-13 * o.t
    for n in range(nt):      # (step sizes differ in the last bits -- tree sums in the norms -- and the steps sit at the stability limit,
        assert relerr(cout[n, 0], ref[n]) < 1e-9      # where rounding differences are amplified; the integrator's own tolerance is 1e-6)
    assert np.array_equal(c_end[0].reshape(-1), cout[-1, 0])


def test_dop853_against_scipy_lanes_and_failures():
    """scipy.integrate.ode('dop853') itself driving the device right-hand side (the reference's arrangement) against the device
    integrator; lanes of a batch are independent (bitwise) and adapt separately; NMAX exit per lane; Calculator(calc='dop853')."""
    import scipy.integrate as si
    d, p, c0, nt, itout = golden_problem(interval=300)
    B, nt = 4, 3
    rng = np.random.default_rng(5)
    cs = np.stack([c0 * rng.uniform(0.5, 1.5) for _ in range(B)])
    pb = np.stack([p.pb] * B); pb[:, 0] = np.linspace(-0.02, 0.05, B)
    flux = np.stack([p.flux_bound] * B)
    with solver_from_problem(p, 'FTCS', batch_capacity=B) as s:
        s.set_batch(cs, pb, [p.vzeta] * B, flux)
        cout, idid, stats, t_end = s.integrate_dop853(nt, [nt - 1], nsteps=10000)
        assert (idid == 1).all() and len(set(stats[:, 0])) > 1
        state = cs.copy()

        def f(t, y):
            state[2] = y
            return s.mol_rhs(state)[2]
        r = si.ode(f).set_integrator('dop853', nsteps=10000)
        r.set_initial_value(cs[2].copy())
        for _ in range(nt):
            r.integrate(r.t + p.dt)
        assert r.successful() and relerr(cout[0, 2], r.y) < 1e-9
        s.set_batch(cs, pb, [p.vzeta] * B, flux)
        c2, idid2, stats2, _ = s.integrate_dop853(nt, [nt - 1], nsteps=2)        # NMAX + 1 = 3 attempted steps, then IDID = -2
        assert (idid2 == -2).all() and (stats2[:, 0] == 3).all() and (stats2[:, 4] == 0).all()
    with solver_from_problem(p, 'FTCS', batch_capacity=1) as s1:
        s1.set_batch(cs[2:3], pb[2:3], [p.vzeta], flux[2:3])
        c1, _, st1, _ = s1.integrate_dop853(nt, [nt - 1], nsteps=10000)
    assert np.array_equal(c1[0, 0], cout[0, 2]) and np.array_equal(st1[0], stats[2])
    # the Calculator seam: calc='dop853' integrates on the device, ode_on_device=False hands the same RHS to scipy
    tp = transport_from_fixture(d)
    tp.c0 = d['c0'].copy(); tp.flux_bound = d['flux_bound'].copy(); tp.system['vzeta'] = float(d['vzeta'])
    outs = []
    for on_device in (True, False):
        calc = Calculator(transport=tp, calc='dop853', dt=float(d['dt']), tmax=float(d['tmax']), ntout=2)
        calc.ode_on_device = on_device
        outs.append(calc.integrate_pnp(tp.dx, tp.nx, tp.dt, tp.nt, tp.ntout, 'dop853'))
    assert len(outs[0]) == len(outs[1]) > 0
    for a, b in zip(*outs):
        assert relerr(a, b) < 1e-9


@pytest.mark.parametrize('calc', ['dopri5', 'dop853', 'odeint'])
def test_adaptive_sweep_example_runs(calc):
    import importlib.util
    spec = importlib.util.spec_from_file_location('mol_adaptive_sweep', os.path.join(os.path.dirname(GOLDEN), '..', 'examples', 'mol_adaptive_sweep.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.main(['--lanes', '9', '--calc', calc, '--tmax', '2e-9'])
    assert (out['idid'] == 1).all() and out['stats'][:, 0].min() >= 4
    k = out['surface_K']
    assert np.all(np.diff(k) < 0) and k[0] > 30.0 > k[-1]       # cations pile up at negative wall potentials, deplete at positive ones


# ---- the stiff integrator on the device: RKC (pnp_integrate_rkc), batched counterpart of odeint / ode('vode' | 'lsoda') ---------------
def rkc_oracle_run(s, lane_state, lane, nt, **kw):
    """oracle/rkc.py on lane `lane`, driving the device right-hand side from the host."""
    from oracle.rkc import Rkc
    state = np.array(lane_state, float)

    def f(t, y):
        state[lane] = y
        return s.mol_rhs(state)[lane]
    o = Rkc(f, **kw).set_initial_value(lane_state[lane].copy())
    out = []
    for n in range(nt):
        if not o.successful():
            break
        out.append(o.integrate((n + 1) * s.dt_ode).copy())
    return o, out


@pytest.mark.parametrize('interval,kw', [(100, {}), (10000, {}), (10000, {'rtol': 1e-4, 'atol': 1e-9}), (2000, {'rtol': 1e-8}),
                                         (10000, {'max_step': 2e-8})])
def test_rkc_same_steps_and_stage_counts_as_the_oracle(interval, kw):
    """Intervals of 100 ... 10 000 times the reference fixture's (1e-9 ... 1e-7 s: up to 400 times the explicit stability limit): the
    device takes the oracle's steps -- same attempted / accepted / rejected counts, same right-hand-side evaluations incl. those of the
    power iteration, same largest stage count -- and lands on the same state to rounding."""
    d, p, c0, nt, itout = golden_problem(interval=interval)
    nt = 5
    with solver_from_problem(p, 'FTCS', batch_capacity=1) as s:
        s.set_batch(c0[None, :], p.pb[None, :], [p.vzeta], p.flux_bound[None, :])
        o, ref = rkc_oracle_run(s, c0[None, :].copy(), 0, nt, **kw)
        s.set_batch(c0[None, :], p.pb[None, :], [p.vzeta], p.flux_bound[None, :])
        cout, idid, stats, t_end = s.integrate_rkc(nt, list(range(nt)), **kw)
        c_end = s.get_state()[0]
    assert idid[0] == 1 and o.idid == 1
    assert list(stats[0]) == [o.nsteps, o.naccpt, o.nrejct, o.nfe, nt - 1, o.nfesig, o.maxm]
    assert interval < 10000 or o.maxm >= 8                   # steps well beyond the explicit limit
    assert t_end[0] == nt * p.dt
    for n in range(nt):
        assert relerr(cout[n, 0], ref[n]) < 1e-10
    assert np.array_equal(c_end[0].reshape(-1), cout[-1, 0])


def test_rkc_agrees_with_odeint_within_the_tolerance():
    """scipy's odeint (LSODA: what the reference calls, calculator_old.py:946-948) on the same device right-hand side."""
    import scipy.integrate as si
    d, p, c0, nt, itout = golden_problem(interval=10000)
    nt = 4
    with solver_from_problem(p, 'FTCS', batch_capacity=1) as s:
        s.set_batch(c0[None, :], p.pb[None, :], [p.vzeta], p.flux_bound[None, :])
        ts = np.arange(nt + 1) * p.dt
        ref = si.odeint(lambda y, t: s.mol_rhs(y[None, :])[0], c0, ts, rtol=1e-10, atol=1e-14, mxstep=200000)
        for rtol, bound in ((1e-4, 2e-3), (1e-6, 5e-5), (1e-8, 2e-6)):
            s.set_batch(c0[None, :], p.pb[None, :], [p.vzeta], p.flux_bound[None, :])
            cout, idid, stats, _ = s.integrate_rkc(nt, list(range(nt)), rtol=rtol, atol=1e-12)
            assert idid[0] == 1
            err = max(relerr(cout[n, 0], ref[n + 1]) for n in range(nt))
            assert err < bound, (rtol, err)


def test_rkc_lanes_are_independent_and_fail_alone():
    d, p, c0, nt, itout = golden_problem(interval=5000)
    B, nt = 6, 3
    rng = np.random.default_rng(5)
    cs = np.stack([c0 * rng.uniform(0.3, 3.0) for _ in range(B)])
    pb = np.stack([p.pb] * B); pb[:, 0] = np.linspace(-0.03, 0.06, B)
    flux = np.stack([p.flux_bound] * B)
    with solver_from_problem(p, 'FTCS', batch_capacity=B) as s:
        s.set_batch(cs, pb, [p.vzeta] * B, flux)
        cout, idid, stats, t_end = s.integrate_rkc(nt, list(range(nt)))
        assert (idid == 1).all() and len(set(stats[:, 0])) > 1 and (t_end == nt * p.dt).all()      # different step counts in one batch
        for b in (1, 4):
            s.set_batch(cs, pb, [p.vzeta] * B, flux)
            o, ref = rkc_oracle_run(s, cs.copy(), b, nt)
            assert list(stats[b][:4]) == [o.nsteps, o.naccpt, o.nrejct, o.nfe] and relerr(cout[-1, b], ref[-1]) < 1e-10
        # a step budget that only some lanes can meet: the others freeze at their last accepted step
        s.set_batch(cs, pb, [p.vzeta] * B, flux)
        need = s.integrate_rkc(1, [0])[2][:, 0].astype(int)          # attempted steps of the first interval (the one that needs most)
        assert len(set(need)) > 1
        budget = int(np.sort(need)[B // 2 - 1])
        s.set_batch(cs, pb, [p.vzeta] * B, flux)
        c2, idid2, st2, t2 = s.integrate_rkc(nt, list(range(nt)), nsteps=budget)
    assert (idid2 == -2).any() and (idid2 == 1).any()
    for b in range(B):
        if idid2[b] == 1:
            assert np.array_equal(c2[:, b], cout[:, b])          # untouched by the neighbours' failures
        else:
            assert t2[b] < nt * p.dt and np.array_equal(c2[int(st2[b][4]), b], c2[-1, b])
    with solver_from_problem(p, 'FTCS', batch_capacity=1) as s1:              # lane 4 alone: bitwise the same
        s1.set_batch(cs[4:5], pb[4:5], [p.vzeta], flux[4:5])
        c1, _, st1, _ = s1.integrate_rkc(nt, list(range(nt)))
    assert np.array_equal(c1[:, 0], cout[:, 4]) and np.array_equal(st1[0], stats[4])


def test_rkc_descriptor_sweep_through_the_calculator():
    """calc='odeint' / 'lsoda' over a batch of operating points: the stiff integrator on the device; output indexing as the reference's
    odeint driver (state at tmesh[n], row 0 the initial state), i.e. the ode family's entry n - 1 (dopri5 on the device, loose bound)."""
    d, p, c0, nt, itout = golden_problem(interval=200)
    tp = transport_from_fixture(d)
    tp.c0 = d['c0'].copy(); tp.flux_bound = d['flux_bound'].copy(); tp.system['vzeta'] = float(d['vzeta'])
    B = 4
    cs = np.stack([tp.c0 * f for f in (1.0, 0.7, 1.3, 2.0)])
    pb = np.stack([tp.pb_array()] * B); pb[:, 0] = np.linspace(-0.02, 0.04, B)
    flux = np.stack([tp.flux_bound[:, 0]] * B)
    outs = {}
    for calc_name in ('odeint', 'lsoda', 'dopri5'):
        calc = Calculator(transport=tp, calc=calc_name, dt=200 * float(d['dt']), tmax=6 * 200 * float(d['dt']), ntout=3)
        calc.ode_options = {'rtol': 1e-8, 'atol': 1e-13, 'nsteps': 100000}
        cout, status, _ = calc.integrate_pnp_batch(cs, pb, [tp.system['vzeta']] * B, flux)
        assert (status == 0).all() and (calc.ode_idid == 1).all()
        outs[calc_name] = (cout, [int(n) for n in tp.itout if n < tp.nt])
    (co, io), (cl, il), (cd, idp) = outs['odeint'], outs['lsoda'], outs['dopri5']
    assert io == il == idp and co.shape == cd.shape == (len(io), B, tp.nspecies * tp.nx) and np.array_equal(co, cl)
    if io[0] == 0:
        assert np.array_equal(co[0], cs)                      # odeint's first row is the initial state
    hit = 0
    for j, n in enumerate(io):                                # odeint's entry n = the ode family's entry n - 1
        if n >= 1 and (n - 1) in idp:
            assert relerr(co[j], cd[idp.index(n - 1)]) < 1e-6
            hit += 1
    assert hit >= 1 or len(io) == 1


def test_rkc_grids_beyond_one_wave_empty_requests_and_argument_checks():
    """Two waves per system (1500 points: the point-wise right-hand side behind the multi-wave Poisson), no intervals / no outputs,
    argument checks, and a Newton handle refusing the call."""
    from catint_amd import _capi
    from catint_amd.synthetic import make_batch
    N, nx, B = 2, 1500, 3
    prob, c0, pb, vz, flux = make_batch(B, N, nx, seed=4, dt_factor=1e-2)
    with solver_from_problem(prob, 'FTCS', batch_capacity=B) as s:
        s.set_batch(c0, pb, vz, flux)
        c_start = s.get_state()[0].reshape(B, -1)
        o, ref = rkc_oracle_run(s, c_start.copy(), 2, 2)
        s.set_batch(c0, pb, vz, flux)
        cout, idid, stats, t_end = s.integrate_rkc(2, [1])
        assert idid[2] == 1 and list(stats[2][:4]) == [o.nsteps, o.naccpt, o.nrejct, o.nfe] and relerr(cout[0, 2], ref[-1]) < 1e-10
        assert o.maxm >= 3                                     # beyond the explicit limit
        s.set_batch(c0, pb, vz, flux)
        c_none, idid0, st0, t0 = s.integrate_rkc(0, [])        # nothing to do
        assert c_none.shape == (0, B, N * nx) and (idid0 == 1).all() and (t0 == 0).all() and (st0 == 0)[:, :4].all()
        assert np.array_equal(s.get_state()[0].reshape(B, -1), c_start)
        c_no_out, idid1, _, t1 = s.integrate_rkc(1, [])        # integrate, report nothing
        assert c_no_out.shape[0] == 0 and (idid1 == 1).all() and (t1 == prob.dt).all()
        with pytest.raises(_capi.PnpError, match='itout'):
            s.integrate_rkc(2, [1, 0])
        with pytest.raises(_capi.PnpError, match='itout'):
            s.integrate_rkc(2, [2])
        with pytest.raises(_capi.PnpError, match='negative'):
            s.integrate_rkc(2, [1], rtol=-1.0)
    with _capi.PnpSolver(N, 64, prob.dx, prob.dt, prob.beta, prob.eps, prob.D, prob.charges, method='Newton', batch_capacity=1) as sn:
        sn.set_newton()
        with pytest.raises(_capi.PnpError, match='physical mode'):
            sn.integrate_rkc(1, [0])
