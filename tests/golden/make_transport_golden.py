#!/usr/bin/env python3
"""Golden answers for the host-side ``Transport`` bulk chemistry, produced by running the *actual reference*
``catint.transport.Transport`` (read-only at /root/reference, importable as-is: SURVEY.md section 8c) in the dev
container on the input dictionaries of ``examples/02_CO2R_Au_CatMAP/run.py:6-93`` and on variations that reach the other
branches of ``initialize_species`` (:537-768), ``initialize_fluxes`` (:929-1095), ``initialize_reactions`` (:1098-1132) and
``set_boundary_conditions`` (:1424-1487).

Dev-only: never imported by tests, smoke() or bench.py.  Only numbers/strings computed by the reference are written
(``tests/golden/transport_cases.json``); no reference source is stored.  The reference iterates over a *set* of species
names while it builds the buffer-equilibrium system (:563,:587), so the unknowns' order -- and with it the last digits of
scipy's fsolve answer -- depends on string hashing: the generator pins PYTHONHASHSEED=0 and the consumer
(tests/test_host_transport_chemistry.py) compares fsolve-derived, un-rounded values to 1e-9 relative.

Usage:  python tests/golden/make_transport_golden.py
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))

DRIVER = r'''
import sys, json, logging, warnings, collections
warnings.filterwarnings('ignore')
import numpy as np
case = json.load(open(sys.argv[1]))
from catint.transport import Transport
from catint.units import unit_NA

species = collections.OrderedDict()
for name, d in case['species']:
    species[name] = dict(d)
system = dict(case['system'])
if system.get('active site density') == 'run.py':
    system['active site density'] = 9.61e-05 / unit_NA * (1e10) ** 2
kw = {}
if case.get('pb_bound') is not None:
    kw['pb_bound'] = case['pb_bound']
if case.get('descriptors') is not None:
    kw['descriptors'] = collections.OrderedDict((k, list(v)) for k, v in case['descriptors'])
tp = Transport(species=species, electrode_reactions=case.get('electrode_reactions'),
               electrolyte_reactions=case.get('electrolyte_reactions'), system=system,
               catmap_args=case.get('catmap_args', {}), comsol_args=case.get('comsol_args', {'par_method': 'internal', 'bin_version': '5.3a'}),
               model_name=case.get('model_name', 'CO2R'), nx=case['nx'], resultsdir='res_' + case['name'], **kw)
logging.disable(logging.CRITICAL)


def plain(v):
    if isinstance(v, (np.floating, float)):
        return float(v)
    if isinstance(v, (np.integer, int)) and not isinstance(v, bool):
        return int(v)
    if isinstance(v, np.ndarray):
        return [plain(x) for x in v.tolist()]
    if isinstance(v, (list, tuple)):
        return [plain(x) for x in v]
    if isinstance(v, dict):
        return {str(k): plain(x) for k, x in v.items()}
    return v

out = {
    'name': case['name'],
    'species_order': list(tp.species.keys()),
    'species': [[sp, plain({k: v for k, v in tp.species[sp].items()})] for sp in tp.species],
    'system': plain({k: v for k, v in tp.system.items() if k not in ('efield', 'potential', 'charge_density')}),
    'nspecies': tp.nspecies, 'charges': plain(tp.charges), 'D': plain(tp.D), 'mu': plain(tp.mu),
    'eps': float(tp.eps), 'beta': float(tp.beta), 'ionic_strength': float(tp.ionic_strength), 'debye_length': float(tp.debye_length),
    'nx': int(tp.nx), 'dx': float(tp.dx), 'xmax': float(tp.xmax), 'xmesh_first': plain(tp.xmesh[:3]), 'xmesh_last': float(tp.xmesh[-1]),
    'c0_rows': plain(tp.c0.reshape(tp.nspecies, tp.nx)[:, [0, tp.nx // 2, tp.nx - 1]]),
    'flux_bound': plain(tp.flux_bound) if hasattr(tp, 'flux_bound') else None,
    'dc_dt_bound': plain(tp.dc_dt_bound), 'efield_bound': plain(tp.efield_bound), 'pb_bound': plain(tp.pb_bound),
    'boundary_type': tp.boundary_type,
    'use_mpb': bool(tp.use_mpb), 'use_migration': bool(tp.use_migration), 'use_convection': bool(tp.use_convection),
    'use_electrolyte_reactions': bool(tp.use_electrolyte_reactions), 'use_electrode_reactions': bool(tp.use_electrode_reactions),
    'use_catmap': bool(getattr(tp, 'use_catmap', False)),
    'electrolyte_reactions': plain(tp.electrolyte_reactions) if tp.electrolyte_reactions is not None else None,
    'electrolyte_reaction_order': list(tp.electrolyte_reactions.keys()) if tp.electrolyte_reactions is not None else None,
    'electrode_reactions': plain(tp.electrode_reactions) if tp.electrode_reactions is not None else None,
    'product_list': list(tp.product_list), 'educt_list': list(tp.educt_list), 'electrolyte_list': list(tp.electrolyte_list),
    'descriptor_keys': list(tp.descriptors.keys()), 'descriptor_lengths': [len(tp.descriptors[k]) for k in tp.descriptors],
    'descriptor_first_last': [[float(tp.descriptors[k][0]), float(tp.descriptors[k][-1])] for k in tp.descriptors],
    'n_alldata': len(tp.alldata_names) if hasattr(tp, 'alldata_names') else None,
    'comsol_args': plain({k: tp.comsol_args[k] for k in ('par_name', 'par_values', 'par_method', 'desc_method', 'solver', 'studies')
                          if k in tp.comsol_args}),
}
json.dump(out, open(sys.argv[2], 'w'), indent=1, sort_keys=True)
print('ok', case['name'], out['species_order'])
'''

PH = 6.8
RUNPY_SYSTEM = {
    'temperature': 298, 'pressure': 1.013, 'bulk_pH': PH, 'boundary thickness': 8.E-05, 'epsilon': 78.36, 'migration': True,
    'electrode reactions': True, 'electrolyte reactions': True, 'charging_scheme': 'comsol', 'phiM': -0.5, 'phiPZC': 0.16,
    'Stern capacitance': 20., 'potential drop': 'Stern', 'active site density': 'run.py',
}
RUNPY_SPECIES = [
    ['K+', {'bulk_concentration': 'charge_neutrality', 'MPB_radius': 2 * 4.1e-10}],
    ['CO2', {'bulk_concentration': 'Henry'}],
    ['OH-', {'bulk_concentration': 10 ** (PH - 14.) * 1000.0}],
    ['CO', {'bulk_concentration': 0.0}],
]
RUNPY_COMSOL = {
    'parameter': {'grid_factor': ['100', 'Grid factor'], 'grid_factor_domain': ['100', 'Grid factor'], 'grid_factor_bound': ['200', 'Grid factor']},
    'solver_settings': {'direct': {'nliniterrefine': True}, 'ramp': {'names': ['PZC', 'CS'], 'dramp': 0.01}},
    'par_method': 'internal', 'bin_version': '5.3a',
}
PHIS = [-0.5 - 0.01 * i for i in range(151)]      # np.linspace(-0.5, -2.0, 151), run.py:45-48


def with_flux(species, **fluxes):
    out = [[n, dict(d)] for n, d in species]
    for n, d in out:
        if n in fluxes:
            d.update(fluxes[n])
    return out


CASES = [
    # A: examples/02_CO2R_Au_CatMAP/run.py verbatim (fluxes owned by CatMAP)
    dict(name='co2r_runpy', species=with_flux(RUNPY_SPECIES, CO={'flux': 'catmap'}, CO2={'flux': 'catmap'}), system=RUNPY_SYSTEM,
         electrolyte_reactions=['bicarbonate-base', 'water-diss', {'additional_cell_reactions': 'bicarbonate-acid'}],
         electrode_reactions={'CO': {'reaction': 'CO2 + H2O + 2 e- -> CO + 2 OH-'}}, nx=200, comsol_args=RUNPY_COMSOL,
         catmap_args={'desc_method': 'automatic', 'min_desc_delta': 0.2, 'max_desc_delta': 0.2, 'n_inter': 'automatic'},
         descriptors=[['phiM', PHIS]]),
    # B: the same system with ONE numeric flux: the others follow from the stoichiometry (flux closure :1034-1095)
    dict(name='co2r_numeric_flux', species=with_flux(RUNPY_SPECIES, CO={'flux': 1.25e-4}), system=RUNPY_SYSTEM,
         electrolyte_reactions=['bicarbonate-base', 'water-diss', {'additional_cell_reactions': 'bicarbonate-acid'}],
         electrode_reactions={'CO': {'reaction': 'CO2 + H2O + 2 e- -> CO + 2 OH-'}}, nx=200),
    # C: product flux given as a current density (:1012-1023), two electrode reactions sharing educts
    dict(name='co2r_h2_current_density',
         species=with_flux(RUNPY_SPECIES + [['H2', {'bulk_concentration': 0.0}]], CO={'current density': 12.5}, H2={'current density': 3.0}),
         system=RUNPY_SYSTEM, electrolyte_reactions=['bicarbonate-base', 'water-diss'],
         electrode_reactions={'CO': {'reaction': 'CO2 + H2O + 2 e- -> CO + 2 OH-'}, 'H2': {'reaction': '2 H2O + 2 e- -> H2 + 2 OH-'}},
         nx=100),
    # D: a flux given as an equation string -> every flux becomes a string expression (:1002-1011, :1080-1088)
    dict(name='co2r_flux_equation', species=with_flux(RUNPY_SPECIES, CO={'flux-equation': 'k0*[[CO2]]*exp(-alpha*phiM)'}),
         system=RUNPY_SYSTEM, electrolyte_reactions=['bicarbonate-base', 'water-diss'],
         electrode_reactions={'CO': {'reaction': 'CO2 + H2O + 2 e- -> CO + 2 OH-'}}, nx=50),
    # E: buffer equilibria closed by a counter-ion constraint instead of given OH- (:719-731)
    dict(name='bicarbonate_constraint',
         species=[['K+', {'bulk_concentration': 100.0}], ['CO2', {'bulk_concentration': 'Henry'}]],
         system={'temperature': 298.14, 'pressure': 1.0, 'boundary thickness': 5e-5, 'electrolyte reactions': True},
         electrolyte_reactions=['bicarbonate-base', {'constraints': {'counter_ion_concentration': 100.0}}], nx=64),
    # F: phosphate buffer, acid form: three unknowns from three equilibria, H+ given through bulk_pH afterwards
    dict(name='phosphate_acid',
         species=[['Na+', {'bulk_concentration': 'charge_neutrality'}], ['H+', {'bulk_concentration': 10 ** (-7.2) * 1000.}],
                  ['H2PO4-', {'bulk_concentration': 40.0}]],
         system={'temperature': 298.14, 'boundary thickness': 1e-4, 'electrolyte reactions': True, 'epsilon': 80.0,
                 'water viscosity': 0.89, 'electrolyte viscosity': 1.02},
         electrolyte_reactions=['phosphate-acid'], nx=128, pb_bound={'potential': {'wall': 'phiM', 'bulk': 0.0}}),
    # G: no reactions at all: defaults, Debye-length mesh (:455-458), default pb_bound
    dict(name='plain_electrolyte', species=[['K+', {'bulk_concentration': 10.0}], ['Cl-', {'bulk_concentration': 10.0}]],
         system={'phiM': -0.05}, nx=100, descriptors=[['phiM', [-0.05, -0.1, -0.15]], ['temperature', [298.14, 310.0]]]),
]


def main():
    tmp = tempfile.mkdtemp(prefix='catint_tgolden_')
    results = []
    try:
        with open(os.path.join(tmp, 'drv.py'), 'w') as f:
            f.write(DRIVER)
        env = dict(os.environ, PYTHONPATH=REF, PYTHONHASHSEED='0', PYTHONDONTWRITEBYTECODE='1', OMP_NUM_THREADS='1')
        for case in CASES:
            cj = os.path.join(tmp, case['name'] + '.json')
            oj = os.path.join(tmp, case['name'] + '.out.json')
            json.dump(case, open(cj, 'w'))
            r = subprocess.run([sys.executable, 'drv.py', cj, oj], cwd=tmp, env=env, capture_output=True, text=True)
            print((r.stdout.strip().splitlines() or ['<no stdout>'])[-1])
            if r.returncode != 0:
                print(r.stderr[-3000:])
                raise SystemExit('reference Transport failed for ' + case['name'])
            results.append({'input': case, 'expected': json.load(open(oj))})
        json.dump(results, open(os.path.join(HERE, 'transport_cases.json'), 'w'), indent=1, sort_keys=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
