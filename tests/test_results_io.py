"""On-disk layout of the reference's results folder (catint/catint_io.py:77-91, :103-134) written/read by catint_amd.results_io."""
import os
import pickle

import numpy as np
import pytest

from catint_amd.results_io import NAMES, read_all, save_all
from catint_amd.transport import Transport


def test_nine_pickles_round_trip(tmp_path):
    tp = Transport(descriptors={'phiM': [-0.1, -0.2]})
    tp.tmesh = np.arange(0, 1.0, 0.25)
    for i, d in enumerate(tp.alldata):
        for sp in tp.species:
            d['species'][sp] = {'concentration': np.linspace(1, 2, tp.nx) * (i + 1), 'surface_concentration': float(i + 1)}
        d['system'] = {'potential': np.zeros(tp.nx), 'surface_pH': 7.0 + i}
    folder = str(tmp_path / 'catint_results')
    save_all(tp, folder)
    assert sorted(os.listdir(folder)) == sorted(n + '.pkl' for n in NAMES)
    raw = pickle.load(open(os.path.join(folder, 'alldata.pkl'), 'rb'))
    sp0 = list(tp.species)[0]
    assert isinstance(raw, list) and isinstance(raw[1]['species'][sp0]['concentration'], list)      # lists of floats, as the reader appends
    assert raw[1]['species'][sp0]['surface_concentration'] == 2.0 and raw[1]['system']['surface_pH'] == 8.0

    class Bare(object):
        pass
    tp2 = read_all(Bare(), folder)
    assert tp2.nx == tp.nx and np.isclose(tp2.dx, tp.dx) and tp2.nt == 4 and np.isclose(tp2.dt, 0.25)
    assert list(tp2.descriptors['phiM']) == [-0.1, -0.2] and list(tp2.species) == list(tp.species)
    assert np.allclose(tp2.alldata[0]['species'][sp0]['concentration'], np.linspace(1, 2, tp.nx))
    save_all(tp, folder, only='alldata_step3')
    read_all(tp2, folder, only='alldata_step3')
    assert len(tp2.alldata_step3) == 2


def synthetic_sweep(lanes=3, nx=48, numeric_flux=False, batch=False):
    """The run.py system (reference input dictionaries) with `lanes` descriptor points whose alldata entries are filled by the SAME
    host code a GPU sweep uses (Calculator.fill_alldata) from smooth synthetic profiles -- no GPU needed."""
    import collections
    import json
    from catint_amd.calculator import Calculator
    from catint_amd.units import unit_NA
    case = next(c for c in json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'transport_cases.json')))
                if c['input']['name'] == ('co2r_numeric_flux' if numeric_flux else 'co2r_runpy'))['input']
    system = dict(case['system'])
    system['active site density'] = 9.61e-05 / unit_NA * (1e10) ** 2
    phis = [-0.5 - 0.1 * i for i in range(lanes)]
    tp = Transport(species=collections.OrderedDict((n, dict(d)) for n, d in case['species']),
                   electrode_reactions=case['electrode_reactions'], electrolyte_reactions=case['electrolyte_reactions'],
                   system=system, nx=nx - 1, descriptors={'phiM': phis}, model_name='CO2R')
    calc = Calculator(transport=tp, calc='comsol')
    x = tp.xmesh / tp.xmesh[-1]
    cb = np.array([tp.species[sp]['bulk_concentration'] for sp in tp.species])
    rows = []
    for i, phi in enumerate(phis):
        v = (phi - 0.16) * 0.3 * np.exp(-40.0 * x)
        c = cb[:, None] * np.exp(-(tp.charges * tp.beta)[:, None] * v[None, :] * 0.2) * (1.0 + 0.1 * (i + 1) * (1 - x)[None, :]) + 1e-6
        g = np.gradient(v, tp.xmesh)
        l = -(tp.charges[:, None] * c).sum(axis=0) / tp.eps
        kin = np.zeros(tp.nspecies)
        names = list(tp.species)
        kin[names.index('CO2')], kin[names.index('CO')], kin[names.index('OH-')] = -1e-5 * (i + 1), 1e-5 * (i + 1), 2e-5 * (i + 1)
        if batch:
            rows.append((c, v, g, l, tp.flux_bound[:, 0], 0, kin))
        else:
            calc.fill_alldata(i, c, v, g, l, tp.flux_bound[:, 0], 0, kin)
    if batch:          # the whole sweep in one call, as Calculator.run does
        calc.fill_alldata_batch(*[np.array([r[j] for r in rows]) for j in range(7)])
    return tp


def test_fill_alldata_batch_equals_point_by_point():
    """Calculator.run fills tp.alldata for the whole sweep in one numpy pass (fill_alldata_batch); descriptor point by descriptor point
    (fill_alldata, one lane) must give the same dictionaries: same keys, same types, same numbers to the bit."""
    one, all_ = synthetic_sweep(lanes=4), synthetic_sweep(lanes=4, batch=True)

    def same(a, b, path):
        assert type(a) is type(b), (path, type(a), type(b))
        if isinstance(a, dict):
            assert list(a) == list(b), (path, list(a), list(b))
            for k in a:
                same(a[k], b[k], path + '/' + str(k))
        elif isinstance(a, np.ndarray):
            assert a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b, equal_nan=True), path
        else:
            assert a == b or (a != a and b != b), (path, a, b)
    for i in range(4):
        same(one.alldata[i], all_.alldata[i], 'alldata[%d]' % i)
    all_.alldata[0]['system']['potential'][0] = 123.0               # rows of one batch array: a lane's entry is its own
    assert all_.alldata[1]['system']['potential'][0] != 123.0


def test_reference_reader_manifest(tmp_path):
    """tests/golden/results_manifest.json is what the REFERENCE's reader + the accesses of tools/plotting_catint.py see in a folder
    written by save_all (generated by tests/golden/make_results_manifest.py).  The same folder through our own reader gives the
    same keys, types and lengths -- lists of floats where the reference's reader appends floats, plain floats for surface values."""
    import json
    from tests import results_walk
    manifest = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'results_manifest.json')))
    tp = synthetic_sweep()
    folder = str(tmp_path / 'CO2R_results')
    save_all(tp, folder)

    class Bare(object):
        pass
    got = results_walk.walk(read_all(Bare(), folder, only=['alldata', 'species', 'system', 'xmesh', 'descriptors', 'electrode_reactions']))
    assert got == manifest
    assert manifest["alldata[i]['species']['CO']['electrode_current_density']"] == ['float', None]
    assert manifest["alldata[i]['system']['pH']"] == ['list', 49] and manifest['n_converged'] == ['int', 3]


def test_comsol_text_export_is_what_the_reference_reader_parsed(tmp_path):
    """tests/golden/comsol_export/*.txt were written by export_comsol_text and PARSED BY THE REFERENCE'S comsol_reader.Reader into exactly
    the exported arrays (tests/golden/make_comsol_export_golden.py asserts that when it generates them).  The exporter must keep producing
    those files byte for byte; a hand parser of the same layout checks the numbers on this side too."""
    from catint_amd.results_io import export_comsol_text
    tp = synthetic_sweep(numeric_flux=True)
    folder = export_comsol_text(tp, 1, str(tmp_path / 'results'))
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'comsol_export')
    for name in ('concentrations.txt', 'electrostatics.txt', 'electrode_flux.txt'):
        assert open(os.path.join(folder, name)).read() == open(os.path.join(gold, name)).read(), name
    rows = [l.split() for l in open(os.path.join(folder, 'concentrations.txt')) if not l.startswith('%')]
    names = list(tp.species)
    assert len(rows) == tp.nx and len(rows[0]) == 1 + len(names)
    got = np.array(rows, float)
    assert np.array_equal(got[:, 0], np.asarray(tp.xmesh)) and np.array_equal(got[:, 1 + names.index('CO2')], tp.alldata[1]['species']['CO2']['concentration'])


def test_resume_from_an_earlier_results_folder(tmp_path):
    """The reference's two start-up hooks of the SCF loop: initialize_surface_concentrations_from_file (calculator.py:242-258, the nine
    pickles of an earlier sweep) and system['init_folder'] (calculator.py:303-309, a COMSOL-layout text folder) -- plus the batched
    variant in which every lane starts from its own descriptor point."""
    from catint_amd.calculator import Calculator
    from catint_amd import results_io
    tp = synthetic_sweep(lanes=3, nx=40, numeric_flux=True)
    folder = str(tmp_path / 'sweep')
    results_io.save_all(tp, folder)
    names = list(tp.species)
    want = np.array([[tp.alldata[i]['species'][sp]['surface_concentration'] for sp in names] for i in range(3)])
    tp2 = synthetic_sweep(lanes=3, nx=40, numeric_flux=True)
    calc = Calculator(transport=tp2, calc='comsol')
    calc.initialize_surface_concentrations_from_file(folder, desc=tp2.descriptors['phiM'][1])          # the reference's call shape
    assert [tp2.species[sp]['surface_concentration'] for sp in names] == list(want[1])
    seen = []

    def transport_fn(flux):
        return seen[-1], np.zeros(3), np.zeros(3)

    def flux_cb(state):
        seen.append(state['surface_concentration'].copy())
        return np.zeros((3, len(names)))
    calc.run_scf_cycle(flux_cb, max_iter=1, transport_fn=transport_fn)
    assert np.array_equal(seen[0], np.repeat(want[1][None, :], 3, axis=0))                              # every lane from that point
    calc.initialize_surface_concentrations_from_file(folder)                                            # batched: lane i from point i
    seen.clear()
    calc.run_scf_cycle(flux_cb, max_iter=1, transport_fn=transport_fn)
    assert np.array_equal(seen[0], want)
    with pytest.raises(Exception):
        calc.initialize_surface_concentrations_from_file(folder, desc=123.0)
    # system['init_folder']: the COMSOL-layout export of descriptor point 2
    txt = str(tmp_path / 'comsol')
    results_io.export_comsol_text(tp, 2, txt)
    x, cols = results_io.read_comsol_text(os.path.join(txt, 'concentrations.txt'))
    assert np.array_equal(x, np.asarray(tp.xmesh, float)) and np.array_equal(cols['cp1'], tp.alldata[2]['species'][names[0]]['concentration'])
    tp3 = synthetic_sweep(lanes=3, nx=40, numeric_flux=True)
    tp3.system['init_folder'] = txt
    calc3 = Calculator(transport=tp3, calc='comsol')
    seen.clear()
    calc3.run_scf_cycle(flux_cb, max_iter=1, transport_fn=transport_fn)
    assert np.array_equal(seen[0], np.repeat(want[2][None, :], 3, axis=0))
