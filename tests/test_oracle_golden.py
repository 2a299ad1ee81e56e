"""Pins the CPU oracle to the reference: golden vectors in tests/golden/*.npz were produced by
running sringe/CatINT's own legacy integrators (tests/golden/make_golden.py).

  * oracle/pnp_ref.py, solver='dense'  : verbatim arithmetic  -> must be BIT-EXACT
  * oracle/pnp_ref.py, solver='banded' : Thomas instead of dense LU -> 1e-12 (measured <= 3e-15)
  * oracle/pnp_oracle.c (C, Thomas)    : 1e-12 (measured <= 3e-15)
"""
import glob
import os

import numpy as np
import pytest

from oracle import pnp_ref as R
from oracle import c_oracle as CO

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
ALL = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, '*.npz')))
STEPPERS = [n for n in ALL if n.startswith(('cn_', 'ftcs_'))]
MOL = [n for n in ALL if n.startswith('odeint_')]


def load(name):
    d = np.load(os.path.join(GOLDEN, name + '.npz'))
    return d, R.problem_from_golden(d)


def test_fixture_inventory():
    # every integrator / Poisson branch / flag the reference can run is represented
    assert len(STEPPERS) >= 13 and len(MOL) >= 2
    for needle in ('cn_dd', 'cn_defaultpb', 'cn_mirrorpb', '_LF', 'ftcs_dd', 'ftcs_defaultpb', 'rates', 'nomig', 'flux'):
        assert any(needle in n for n in ALL), needle


@pytest.mark.parametrize('name', STEPPERS)
def test_dense_restatement_is_bit_exact(name):
    d, (p, c0, nt, itout, method) = load(name)
    cout, (v, g, l) = R.integrate(p, c0, nt, itout, method, solver='dense')
    assert np.array_equal(np.array(cout), d['cout'])
    if d['potential'].size:
        assert np.array_equal(v, d['potential'])
        assert np.array_equal(-g, d['efield'])
        assert np.array_equal(-l * p.eps, d['total_charge'])


@pytest.mark.parametrize('name', STEPPERS)
def test_banded_and_c_oracle(name):
    d, (p, c0, nt, itout, method) = load(name)
    ref = d['cout']
    scale = np.abs(ref).max()
    cout, _ = R.integrate(p, c0, nt, itout, method, solver='banded')
    assert np.abs(np.array(cout) - ref).max() / scale < 1e-12
    cc, pot = CO.integrate(p, np.stack([c0, c0]), nt, itout, method)
    assert np.abs(cc[:, 0] - ref).max() / scale < 1e-12
    assert np.array_equal(cc[:, 0], cc[:, 1])
    if d['potential'].size:
        assert np.abs(pot[0][0] - d['potential']).max() <= 1e-12 * max(np.abs(d['potential']).max(), 1e-30)


@pytest.mark.parametrize('name', MOL)
def test_method_of_lines_rhs(name):
    d, (p, c0, nt, itout, method) = load(name)
    for s, val in zip(d['rhs_states'], d['rhs_values']):
        assert np.array_equal(R.mol_rhs(s, p), val)


def test_itout_rule():
    for name in ALL:
        d = np.load(os.path.join(GOLDEN, name + '.npz'))
        # fixtures store the *effective* ntout; the selection rule is re-derived from the stored itout
        nt = int(d['nt'])
        it = [int(i) for i in d['itout']]
        assert it[-1] == nt - 1
    assert R.make_itout(21, 4) == [5, 10, 15, 20]
    assert R.make_itout(11, 2) == [5, 10]
    assert R.make_itout(12, 2) == [6, 11]
    assert R.make_itout(41, 4) == [10, 20, 30, 40]


def test_gouy_chapman_known_answer_is_recorded():
    d = np.load(os.path.join(GOLDEN, 'cn_dd_n2_nx50.npz'))
    gc = d['gouy_chapman']
    assert gc.shape == (int(d['nx']), 2)
    assert abs(gc[0, 0] - float(d['phiM'])) < 1e-12      # phi(0) = phiM
    # Debye length, transport.py:439-443
    I = 0.5 * (d['charges'] ** 2 * d['c_bulk']).sum()
    assert abs(np.sqrt(float(d['eps']) / float(d['beta']) / 2. / I) - float(d['debye_length'])) < 1e-20
