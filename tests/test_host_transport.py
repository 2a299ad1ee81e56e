"""Host logic (CPU): the Transport / Calculator mirrors reproduce the reference's numeric plumbing.
Expected values come from the golden fixtures, i.e. from the reference's own Transport/Calculator."""
import collections
import glob
import os

import numpy as np
import pytest

from catint_amd.transport import Transport, TransportError, charge_from_symbol
from catint_amd.calculator import Calculator, CalculatorError, make_itout
from catint_amd.host import pb_mode_from_bound
from catint_amd import PB_DD, PB_VWALL_GBULK, PB_GWALL_VBULK

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
ALL = sorted(glob.glob(os.path.join(GOLDEN, '*.npz')))


def transport_from_fixture(d, **kw):
    names = [str(s) for s in d['species']]
    species = None
    if not names[0].startswith('species'):
        species = collections.OrderedDict((n, {'bulk_concentration': float(c)}) for n, c in zip(names, d['c_bulk']))
    pb = d['pb_bound']
    pbd = {'potential': {}, 'gradient': {}}
    for (k1, k2), v in zip([('potential', 'wall'), ('potential', 'bulk'), ('gradient', 'wall'), ('gradient', 'bulk')], pb):
        if not np.isnan(v):
            pbd[k1][k2] = 'phiM' if (k1, k2) == ('potential', 'wall') else float(v)
    return Transport(species=species, system={'phiM': float(d['phiM']), 'boundary thickness': float(d['xmax'])},
                     pb_bound=pbd, nx=int(d['nx_requested']), **kw)


@pytest.mark.parametrize('path', ALL, ids=[os.path.basename(p)[:-4] for p in ALL])
def test_transport_numbers_match_reference(path):
    d = np.load(path)
    tp = transport_from_fixture(d)
    assert np.array_equal(tp.D, d['D'])
    assert np.array_equal(tp.charges, d['charges'])
    assert np.array_equal(tp.mu, d['mu'])
    assert tp.beta == float(d['beta']) and tp.eps == float(d['eps'])
    assert tp.debye_length == float(d['debye_length']) and tp.ionic_strength == float(d['ionic_strength'])
    assert tp.dx == float(d['dx']) and tp.nx == int(d['nx']) and np.array_equal(tp.xmesh, d['xmesh'])
    assert np.array_equal(np.nan_to_num(tp.pb_array(), nan=-7), np.nan_to_num(d['pb_bound'], nan=-7))
    if 'gc' not in path and 'c0_perturb' not in path:
        plain = np.repeat(d['c_bulk'], int(d['nx']))
        if np.array_equal(plain, d['c0']):
            assert np.array_equal(tp.c0, d['c0'])


def test_gouy_chapman_initial_profile_matches_reference():
    d = np.load(os.path.join(GOLDEN, 'cn_defaultpb_n2_nx50_gc.npz'))
    tp = transport_from_fixture(d)
    tp.set_initial_concentrations('Gouy-Chapman')
    assert np.array_equal(tp.c0, d['c0'])
    gc = np.array([tp.gouy_chapman(x) for x in tp.xmesh])
    assert np.array_equal(gc, d['gouy_chapman'])


@pytest.mark.parametrize('path', ALL, ids=[os.path.basename(p)[:-4] for p in ALL])
def test_time_mesh_and_itout(path):
    d = np.load(path)
    tp = transport_from_fixture(d)
    method = str(d['method'])
    # the fixture stores the effective ntout; any requested ntout that maps onto the same itout is fine
    for ntout in range(1, 8):
        if make_itout(int(d['nt']), ntout) == [int(i) for i in d['itout']]:
            break
    else:
        pytest.fail('no ntout reproduces the fixture itout')
    calc = Calculator(transport=tp, calc=method, dt=float(d['dt']), tmax=float(d['tmax']), ntout=ntout)
    assert tp.nt == int(d['nt']) and tp.itout == [int(i) for i in d['itout']]
    assert calc.use_lax_friedrich == bool(d['lax_friedrich'])


def test_charge_parser_and_errors():
    assert [charge_from_symbol(s) for s in ('K^+', 'CO_3^{2-}', 'CO_2', 'Ca^{2+}', 'PO_4^{3-}', 'H^+')] == [1, -2, 0, 2, -3, 1]
    with pytest.raises(TransportError):
        Transport(species={'Xx+': {'bulk_concentration': 1.0}})
    tp = Transport()
    with pytest.raises(CalculatorError):
        Calculator(transport=tp, calc='nonsense', dt=1e-10, tmax=1e-9)
    for broken in ('vode', 'odespy'):     # not runnable in the reference either
        with pytest.raises(CalculatorError):
            Calculator(transport=tp, calc=broken, dt=1e-10, tmax=1e-9)
    phys = Calculator(transport=tp, calc='comsol')          # the production path's name selects the implicit GPU mode
    assert phys.physical and phys.mode == 'stationary'
    with pytest.raises(CalculatorError):
        Calculator(transport=None, calc='FTCS')
    with pytest.raises(ValueError):
        pb_mode_from_bound([np.nan, np.nan, 0.0, 0.0])
    assert pb_mode_from_bound([0.1, 0.0, np.nan, np.nan]) == PB_DD
    assert pb_mode_from_bound([0.1, np.nan, np.nan, 0.0]) == PB_VWALL_GBULK
    assert pb_mode_from_bound([np.nan, 0.0, 0.0, np.nan]) == PB_GWALL_VBULK


def test_descriptors_and_default_mesh():
    tp = Transport(descriptors={'phiM': [-0.1, -0.2, -0.3]}, nx=200)
    assert list(tp.descriptors.keys()) == ['phiM', 'temperature'] and len(tp.alldata_names) == 3
    # without 'boundary thickness': nx=200 -> 401 points of a 20-Debye-length cell (transport.py:456-460)
    assert tp.nx == 401 and abs(tp.xmax - 20 * tp.debye_length) < 1e-20


def test_evaluate_accuracy_signed_quirk():
    acc = Calculator.evaluate_accuracy({'a': 1.0, 'b': -2.0}, {'a': 0.9, 'b': -1.0})
    assert abs(acc - 0.1) < 1e-12   # the negative-current species contributes -0.5 and never gates
