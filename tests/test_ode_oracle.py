"""oracle/dopri5.py pinned against the integrator the reference calls: scipy.integrate.ode(f).set_integrator('dopri5', nsteps=10000),
one integrate(t + dt) per interval (catint/calculator_old.py:955-963).  Same sequence of right-hand-side evaluations (times),
same trajectory bit for bit, same failure exits."""
import warnings

import numpy as np
import pytest
import scipy.integrate as si

from oracle.dopri5 import Dopri5
from oracle.dop853 import Dop853


def logged(fun):
    calls = []

    def f(t, y):
        calls.append(t)
        return fun(t, y)
    return f, calls


def oscillator(t, y):
    return np.array([y[1], -y[0] * (1 + 0.3 * np.sin(y[1])), -50 * (y[2] - np.cos(t))])


def diffusion(t, y):        # a small method-of-lines problem: heat equation with a reaction term, 24 unknowns
    d = np.zeros_like(y)
    d[1:-1] = 400.0 * (y[2:] - 2 * y[1:-1] + y[:-2]) - 3.0 * y[1:-1] ** 2
    return d


@pytest.mark.parametrize('fun,y0,dt,nt,kw', [
    (oscillator, [1.0, 0.0, 0.3], 0.7, 30, {}),
    (oscillator, [1.0, 0.0, 0.3], 0.05, 40, {'rtol': 1e-9, 'atol': 1e-10}),
    (diffusion, list(np.sin(np.linspace(0, np.pi, 24)) ** 2), 0.01, 25, {}),
    (diffusion, list(np.sin(np.linspace(0, np.pi, 24)) ** 2), 0.01, 10, {'first_step': 1e-4, 'max_step': 2e-3, 'safety': 0.8, 'ifactor': 5.0,
                                                                         'dfactor': 0.3, 'beta': 0.08}),
    (diffusion, list(np.sin(np.linspace(0, np.pi, 24)) ** 2), 0.01, 10, {'beta': -1.0}),
])
def test_restatement_equals_scipy_bit_for_bit(fun, y0, dt, nt, kw):
    f, ca = logged(fun)
    g, cb = logged(fun)
    r = si.ode(f).set_integrator('dopri5', nsteps=10000, **kw)
    r.set_initial_value(y0)
    o = Dopri5(g, nsteps=10000, **kw).set_initial_value(y0)
    for _ in range(nt):
        a = r.integrate(r.t + dt)
        b = o.integrate(o.t + dt)
        assert np.array_equal(a, b) and r.t == o.t
    # DOPRI5 counts two evaluations per call; the second one (HINIT's) only happens in the first call, and only without first_step
    assert np.array_equal(ca, cb) and len(ca) == o.nfcn - nt + (0 if kw.get('first_step') else 1)


def stiff(t, y):
    return np.array([-2e4 * y[0] + np.sin(t), -y[1]])


@pytest.mark.parametrize('nsteps,idid,msg', [(100000, -4, 'stiff'), (50, -2, 'larger nsteps')])
def test_failure_exits_match_scipy(nsteps, idid, msg):
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        r = si.ode(stiff).set_integrator('dopri5', nsteps=nsteps)
        r.set_initial_value([1.0, 1.0])
        a = r.integrate(5.0)
    assert not r.successful() and msg in str(w[0].message)
    o = Dopri5(stiff, nsteps=nsteps).set_initial_value([1.0, 1.0])
    b = o.integrate(5.0)
    assert o.idid == idid and not o.successful() and o.t == r.t and np.array_equal(a, b)


def test_step_size_is_carried_between_calls_but_the_controller_memory_is_not():
    f, calls = logged(oscillator)
    o = Dopri5(f).set_initial_value([1.0, 0.0, 0.3])
    o.integrate(0.7)
    n1 = len(calls)
    h_carried = o.h
    nlog = len(o.log)
    o.integrate(1.4)
    # second call: one evaluation for k1 (no HINIT), first attempted step = the carried prediction
    assert calls[n1] == 0.7 and o.log[nlog][1] == h_carried and h_carried < 0.7
    assert (len(calls) - n1 - 1) % 6 == 0


@pytest.mark.parametrize('fun,y0,dt,nt,kw', [
    (oscillator, [1.0, 0.0, 0.3], 0.7, 30, {}),
    (oscillator, [1.0, 0.0, 0.3], 0.05, 40, {'rtol': 1e-9, 'atol': 1e-10}),
    (diffusion, list(np.sin(np.linspace(0, np.pi, 24)) ** 2), 0.01, 25, {}),
    (diffusion, list(np.sin(np.linspace(0, np.pi, 24)) ** 2), 0.01, 10, {'first_step': 1e-4, 'max_step': 2e-3, 'safety': 0.8, 'ifactor': 5.0,
                                                                         'dfactor': 0.3, 'beta': 0.08}),
])
def test_dop853_restatement_equals_scipy_bit_for_bit(fun, y0, dt, nt, kw):
    """oracle/dop853.py against scipy.integrate.ode('dop853'): same evaluation times (including scipy's redundant f(x, y) at the start
    of every step), same trajectory, rejected steps included."""
    f, ca = logged(fun)
    g, cb = logged(fun)
    r = si.ode(f).set_integrator('dop853', nsteps=10000, **kw)
    r.set_initial_value(y0)
    o = Dop853(g, nsteps=10000, **kw).set_initial_value(y0)
    for _ in range(nt):
        a = r.integrate(r.t + dt)
        b = o.integrate(o.t + dt)
        assert np.array_equal(a, b) and r.t == o.t
    assert np.array_equal(ca, cb)
    # without the redundant evaluation (what the device integrator does): the same trajectory, one evaluation less per attempted step
    g2, cc = logged(fun)
    o2 = Dop853(g2, nsteps=10000, recompute_k1=False, **kw).set_initial_value(y0)
    for _ in range(nt):
        b2 = o2.integrate(o2.t + dt)
    assert np.array_equal(b2, b) and len(cc) == len(cb) - len(o.log) and len(o2.log) == len(o.log)
    assert len(cc) == o2.nfcn - nt + (0 if kw.get('first_step') else 1)


@pytest.mark.parametrize('nsteps,idid,msg', [(100000, -4, 'stiff'), (50, -2, 'larger nsteps')])
def test_dop853_failure_exits_match_scipy(nsteps, idid, msg):
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        r = si.ode(stiff).set_integrator('dop853', nsteps=nsteps)
        r.set_initial_value([1.0, 1.0])
        a = r.integrate(5.0)
    assert not r.successful() and msg in str(w[0].message)
    o = Dop853(stiff, nsteps=nsteps).set_initial_value([1.0, 1.0])
    b = o.integrate(5.0)
    assert o.idid == idid and o.t == r.t and np.array_equal(a, b)


# ---- oracle/rkc.py: Runge-Kutta-Chebyshev with error control (what pnp_integrate_rkc runs per lane) ------------------------------------
def stiff_diffusion(t, y):        # heat equation + reaction on 64 points: spectral radius 1.6e4
    d = np.zeros_like(y)
    d[1:-1] = 4000.0 * (y[2:] - 2 * y[1:-1] + y[:-2]) - 3.0 * y[1:-1] ** 2
    return d


@pytest.mark.parametrize('rtol,bound', [(1e-4, 2e-4), (1e-6, 1e-5), (1e-8, 1e-6)])
def test_rkc_restatement_against_odeint(rtol, bound):
    from oracle.rkc import Rkc
    y0 = np.sin(np.linspace(0, np.pi, 64)) ** 2
    ts = np.arange(0, 11) * 0.01
    ref = si.odeint(stiff_diffusion, y0, ts, tfirst=True, rtol=1e-12, atol=1e-14)
    r = Rkc(stiff_diffusion, rtol=rtol, atol=1e-10).set_initial_value(y0)
    err = 0.0
    for k in range(1, 11):
        err = max(err, np.abs(r.integrate(ts[k]) - ref[k]).max())
    assert r.idid == 1 and err < bound
    # the point of the method: far fewer steps than the explicit limit 2 / rho would allow (0.1 / 1.25e-4 = 800 Euler steps)
    assert r.sprad == pytest.approx(1.6e4 * 1.2, rel=0.1) and (rtol < 1e-5 or r.nsteps < 100) and r.nrejct <= 2
    assert r.nfe == sum(e[2] for e in r.log) + 2          # m evaluations per attempted step + f(y0) + the first-step probe


def test_rkc_method_of_lines_right_hand_side_and_failure_exits():
    """The reference's ode_func (oracle/pnp_ref.mol_rhs, pinned against the reference's fixtures) under RKC against odeint."""
    import os
    from oracle import pnp_ref as R
    from oracle.rkc import Rkc
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'dopri5_dd_n2_nx50.npz'))
    p, c0, nt, itout, method = R.problem_from_golden(d)

    def f(t, y):
        return R.mol_rhs(y, p)
    dt = 1e4 * p.dt                      # 1e-7 s per interval: 400 x the explicit stability limit dx^2 / (2 D)
    ts = np.arange(0, 4) * dt
    ref = si.odeint(f, c0, ts, tfirst=True, rtol=1e-10, atol=1e-14)
    r = Rkc(f, rtol=1e-6, atol=1e-12).set_initial_value(c0)
    for k in range(1, 4):
        y = r.integrate(ts[k])
        assert np.abs(y - ref[k]).max() / np.abs(ref[k]).max() < 5e-5
    assert r.idid == 1 and r.maxm >= 8 and r.t == ts[3]
    r2 = Rkc(f, nsteps=3).set_initial_value(c0)
    r2.integrate(dt)
    assert r2.idid == -2 and not r2.successful() and r2.t < dt
    y_frozen = r2.y.copy()
    assert np.array_equal(r2.integrate(2 * dt), y_frozen)
