#!/usr/bin/env python3
"""What the reference's COMSOL model generator hands to COMSOL for the examples/02_CO2R_Au_CatMAP/run.py system, as data:
every ``set("name", "expression", ...)`` pair that ``catint/comsol_model.py`` emits from its ``param`` (:1093-1182),
``variables`` (:940-1091) and ``physics.tds`` (:682-919) classes -- diffusion coefficients, charges, bulk concentrations,
cell length, temperature, Debye length, the phiPZC / Stern-capacitance continuation ramps, the Stern surface charge
``rho_s``, the size-modified volume fraction / activity coefficient / drift expressions, the rate constants and the
mass-action source term of every species.  The physical-mode oracle (oracle/pnp_physical.py) restates this model; the
fixture lets tests/test_physical_oracle.py assert its coefficients and its reaction source against the reference's own
emitted numbers and expressions instead of hand-copied ones.  (The SOLVE stays unpinned: COMSOL is not available.)

Dev-only: runs the reference (read-only at /root/reference) in a scratch dir; writes numbers/expression strings only to
tests/golden/comsol_model_runpy.json.      Usage:  python tests/golden/make_comsol_model_golden.py
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
import sys, json, logging, warnings, collections, re
warnings.filterwarnings('ignore')
import numpy as np
case = json.load(open(sys.argv[1]))
from catint.transport import Transport
from catint.comsol_model import Model
from catint.units import unit_NA

species = collections.OrderedDict((name, dict(d)) for name, d in case['species'])
system = dict(case['system'])
if system.get('active site density') == 'run.py':
    system['active site density'] = 9.61e-05 / unit_NA * (1e10) ** 2
tp = Transport(species=species, electrode_reactions=case.get('electrode_reactions'), electrolyte_reactions=case.get('electrolyte_reactions'),
               system=system, catmap_args={}, comsol_args=case['comsol_args'], model_name='CO2R', nx=case['nx'],
               descriptors=collections.OrderedDict((k, list(v)) for k, v in case['descriptors']), resultsdir='res')
logging.disable(logging.CRITICAL)
args = tp.comsol_args
args['bin_version'] = 5.31            # what comsol_wrapper.py:82 makes of '5.3a'
pairs = {}
for label, obj in (('param', Model.param(tp, comsol_args=args)),
                   ('variables_boundary', Model.variables(tp, index=1, geo='b1', comsol_args=args)),
                   ('variables_domain', Model.variables(tp, index=2, geo='d1', comsol_args=args)),
                   ('physics_tds', Model.physics(tp, methods=['tds'], comsol_args=args))):
    text = obj.get()
    found = re.findall(r'\.set\("([^"]+)",\s*"([^"]*)"', text)
    pairs[label] = collections.OrderedDict(found)
out = {'species_order': list(tp.species.keys()), 'pairs': pairs,
       'debye_length': float(tp.debye_length), 'xmax': float(tp.xmax),
       'stern_capacitance_input': tp.system['Stern capacitance'], 'phiPZC': tp.system['phiPZC'], 'phiM': tp.system['phiM'],
       'par_name': args['par_name'], 'par_values': args['par_values'], 'ramp': args['solver_settings']['ramp']}
json.dump(out, open(sys.argv[2], 'w'), indent=1)
print('ok', {k: len(v) for k, v in pairs.items()})
'''


def main():
    sys.path.insert(0, HERE)
    import make_transport_golden as T
    tmp = tempfile.mkdtemp(prefix='catint_cgolden_')
    try:
        with open(os.path.join(tmp, 'drv.py'), 'w') as f:
            f.write(DRIVER)
        env = dict(os.environ, PYTHONPATH=REF, PYTHONHASHSEED='0', PYTHONDONTWRITEBYTECODE='1', OMP_NUM_THREADS='1')
        outs = {}
        for name in ('co2r_runpy', 'co2r_numeric_flux', 'co2r_numeric_flux_rf2'):
            # (the third case: the numeric-flux system with a roughness factor of 2 -- j_i = RF*flux_factor*flux_i, comsol_model.py:1134)
            case = dict(next(c for c in T.CASES if c['name'] == name.replace('_rf2', '')))
            if name.endswith('_rf2'):
                case['name'] = name
                case['system'] = dict(case['system'], RF=2.0)
            case.setdefault('comsol_args', T.RUNPY_COMSOL)
            case.setdefault('descriptors', [['phiM', [-0.5, -0.6]]])
            cj, oj = os.path.join(tmp, name + '.json'), os.path.join(tmp, name + '.out.json')
            json.dump(case, open(cj, 'w'))
            r = subprocess.run([sys.executable, 'drv.py', cj, oj], cwd=tmp, env=env, capture_output=True, text=True)
            print((r.stdout.strip().splitlines() or ['<no stdout>'])[-1])
            if r.returncode != 0:
                print(r.stderr[-3000:])
                raise SystemExit('reference comsol_model failed for ' + name)
            outs[name] = json.load(open(oj))
        json.dump(outs, open(os.path.join(HERE, 'comsol_model_runpy.json'), 'w'), indent=1, sort_keys=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
