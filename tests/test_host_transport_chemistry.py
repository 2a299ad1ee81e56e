"""Host logic (CPU): ``catint_amd.transport.Transport`` built from the reference's own input dictionaries
(examples/02_CO2R_Au_CatMAP/run.py:6-93 and variations) reproduces what the reference's ``Transport`` computes -- species
order incl. reaction-derived additions, charges, D, Henry's law, buffer equilibria (scipy fsolve), electroneutrality closure,
pH, activity coefficients, Debye length, mesh, reaction tables, educt/product lists, wall-flux closure, boundary arrays.

Expected values: tests/golden/transport_cases.json, written by tests/golden/make_transport_golden.py from the reference
itself.  Tolerances: everything is compared exactly except bulk concentrations that come out of fsolve WITHOUT the
electroneutrality pass's 8-decimal rounding (the reference walks a hash-ordered set there, see the generator's docstring):
1e-9 relative, stated below."""
import collections
import json
import os

import numpy as np
import pytest

from catint_amd.transport import Transport, parse_reaction_table
from catint_amd.units import unit_NA

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'transport_cases.json')
CASES = json.load(open(GOLDEN))
FSOLVE_RTOL = 1e-9


def build(inp):
    species = collections.OrderedDict((name, dict(d)) for name, d in inp['species'])
    system = dict(inp['system'])
    if system.get('active site density') == 'run.py':
        system['active site density'] = 9.61e-05 / unit_NA * (1e10) ** 2
    kw = {}
    if inp.get('pb_bound') is not None:
        kw['pb_bound'] = inp['pb_bound']
    if inp.get('descriptors') is not None:
        kw['descriptors'] = collections.OrderedDict((k, list(v)) for k, v in inp['descriptors'])
    return Transport(species=species, electrode_reactions=inp.get('electrode_reactions'),
                     electrolyte_reactions=inp.get('electrolyte_reactions'), system=system, nx=inp['nx'],
                     comsol_args=inp.get('comsol_args'), catmap_args=inp.get('catmap_args'), model_name='CO2R', **kw)


def close(a, b, rtol):
    if isinstance(b, str) or isinstance(a, str):
        return a == b
    return abs(a - b) <= rtol * max(abs(b), 1e-300) or a == b


@pytest.mark.parametrize('case', CASES, ids=[c['input']['name'] for c in CASES])
def test_transport_chemistry_matches_reference(case):
    inp, exp = case['input'], case['expected']
    tp = build(inp)
    assert list(tp.species.keys()) == exp['species_order'] and tp.nspecies == exp['nspecies']
    rounded = any(d.get('bulk_concentration') == 'charge_neutrality' for _, d in inp['species'])
    for sp, ref in exp['species']:
        got = tp.species[sp]
        for key in ('charge', 'diffusion', 'name', 'symbol', 'Henry constant', 'MPB_radius', 'current density', 'flux-equation'):
            assert got.get(key) == ref.get(key), (sp, key, got.get(key), ref.get(key))
        # bulk / surface concentrations: exact when the reference rounded them (or they were inputs), fsolve tolerance otherwise
        tol = 0.0 if rounded else FSOLVE_RTOL
        assert close(got['bulk_concentration'], ref['bulk_concentration'], tol), (sp, got['bulk_concentration'], ref['bulk_concentration'])
        assert close(got['surface_concentration'], ref['surface_concentration'], tol)
        assert close(got['surface_activity_coefficient'], ref['surface_activity_coefficient'], 1e-15)
        if isinstance(ref['flux'], str):
            assert got['flux'] == ref['flux'], (sp, got['flux'], ref['flux'])
        else:
            assert got['flux'] == ref['flux'], (sp, got['flux'], ref['flux'])
    tol = 1e-15 if rounded else FSOLVE_RTOL
    assert np.array_equal(tp.charges, np.array(exp['charges'])) and np.array_equal(tp.D, np.array(exp['D']))
    assert np.array_equal(tp.mu, np.array(exp['mu'])) and tp.eps == exp['eps'] and tp.beta == exp['beta']
    assert close(tp.ionic_strength, exp['ionic_strength'], tol) and close(tp.debye_length, exp['debye_length'], tol)
    assert tp.nx == exp['nx'] and close(tp.dx, exp['dx'], tol) and close(tp.xmax, exp['xmax'], tol)
    assert np.allclose(tp.xmesh[:3], exp['xmesh_first'], rtol=max(tol, 1e-15), atol=0) and close(tp.xmesh[-1], exp['xmesh_last'], max(tol, 1e-15))
    rows = tp.c0.reshape(tp.nspecies, tp.nx)[:, [0, tp.nx // 2, tp.nx - 1]]
    assert np.allclose(rows, np.array(exp['c0_rows']), rtol=tol if tol else 0, atol=0)
    # system entries the path reads
    for key in ('bulk_pH', 'surface_pH', 'surface_potential', 'RF', 'reference_gas_concentration', 'temperature', 'epsilon', 'phiM',
                'phiPZC', 'Stern capacitance', 'pressure'):
        assert close(tp.system[key], exp['system'][key], tol), (key, tp.system[key], exp['system'][key])
    assert tp.system['exclude species'] == exp['system']['exclude species'] and tp.system['pH'] == pytest.approx(exp['system']['pH'], rel=max(tol, 1e-15))
    # boundary arrays
    if exp['flux_bound'] is None:      # symbolic fluxes: the reference creates no array; ours holds zeros for the kinetics to fill
        assert tp.flux_symbolic and not tp.flux_bound.any()
    else:
        assert not tp.flux_symbolic and np.array_equal(tp.flux_bound, np.array(exp['flux_bound']))
    assert np.array_equal(tp.dc_dt_bound, np.array(exp['dc_dt_bound'])) and list(tp.efield_bound) == exp['efield_bound']
    assert tp.pb_bound == exp['pb_bound'] and tp.boundary_type == exp['boundary_type']
    for flag in ('use_mpb', 'use_migration', 'use_convection', 'use_electrolyte_reactions', 'use_electrode_reactions'):
        assert getattr(tp, flag) == exp[flag], flag
    assert getattr(tp, 'use_catmap', False) == exp['use_catmap']
    # reaction tables
    if exp['electrolyte_reactions'] is None:
        assert tp.electrolyte_reactions is None
    else:
        assert list(tp.electrolyte_reactions.keys()) == exp['electrolyte_reaction_order']
        for key, ref in exp['electrolyte_reactions'].items():
            got = tp.electrolyte_reactions[key]
            assert got['reaction'] == ref['reaction'] and got['constant'] == ref['constant'] and got.get('rates') == ref.get('rates')
    if exp['electrode_reactions'] is None:
        assert tp.electrode_reactions is None
    else:
        assert {k: dict(v) for k, v in tp.electrode_reactions.items()} == exp['electrode_reactions']
    assert tp.product_list == exp['product_list'] and tp.educt_list == exp['educt_list'] and tp.electrolyte_list == exp['electrolyte_list']
    assert list(tp.descriptors.keys()) == exp['descriptor_keys']
    assert [len(tp.descriptors[k]) for k in tp.descriptors] == exp['descriptor_lengths']
    if exp['n_alldata'] is not None:
        assert len(tp.alldata_names) == exp['n_alldata']


def test_runpy_system_known_answers():
    """SURVEY.md App. E, as numbers: the example's bulk state after equilibria + neutrality, and what the solver is handed."""
    case = next(c for c in CASES if c['input']['name'] == 'co2r_runpy')
    tp = build(case['input'])
    names = list(tp.species.keys())
    assert names == ['K+', 'CO2', 'OH-', 'CO', 'HCO3-', 'CO32-', 'H+']
    cb = [tp.species[s]['bulk_concentration'] for s in names]
    assert cb[:6] == [93.70466795, 33.429, 6.31e-05, 0.0, 93.64969242, 0.02753546] and abs(cb[6] - 1.584893192e-4) < 1e-13
    assert abs(tp.debye_length - 9.924888854e-10) < 1e-18 and tp.nx == 201 and abs(tp.dx - 4e-7) < 1e-20
    # the mass-action table the transport solve reads: five reactions, H2O (excluded, unit activity) dropped
    assert list(tp.reactions.keys()) == ['buffer-base', 'buffer-base2', 'self-dissociation of water', 'buffer-acid', 'buffer-acid2']
    assert tp.reactions['self-dissociation of water']['reactants'] == [[], ['OH-', 'H+']]
    assert tp.reactions['buffer-acid2'] == {'reactants': [['HCO3-'], ['CO32-', 'H+']], 'rates': [59.44, 1275536480.6866953]}
    assert tp.electrode_reactions['CO']['nel'] == 2 and tp.use_reactions


def test_reaction_string_parser():
    t = parse_reaction_table({'a': {'reaction': 'CO2 + H2O + 2 e- -> CO + 2 OH-'}, 'b': {'reaction': 'H2O <-> OH- + H+'},
                              'c': {'reaction': '2 *H + 3 e- -> H2'}})
    assert t['a'] == {'reaction': [['CO2', 'H2O', 'e-', 'e-'], ['CO', 'OH-', 'OH-']], 'nel': 2}
    assert t['b']['reaction'] == [['H2O'], ['OH-', 'H+']]
    assert t['c']['reaction'] == [['*H', '*H', 'e-', 'e-', 'e-'], ['H2']] and t['c']['nel'] == 3
