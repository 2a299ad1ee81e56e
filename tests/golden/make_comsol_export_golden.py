#!/usr/bin/env python3
"""f4, COMSOL-text export: catint_amd.results_io.export_comsol_text writes a descriptor point as the three text tables the reference
asks COMSOL for; here the REFERENCE's own reader (catint.comsol_reader.Reader.read_all, driven on a reference Transport of the same
system) parses them, and what it puts into tp.species / tp.system / tp.alldata is compared with the arrays that were exported
(exactly: the tables carry repr() floats).  The verified text files are committed as tests/golden/comsol_export/*.txt; the CPU test
(tests/test_results_io.py) asserts that the exporter still produces them byte for byte.

Dev-only (the reference is imported in a subprocess, read-only).    Usage:  python tests/golden/make_comsol_export_golden.py
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

READER = r'''
import sys, json, logging, warnings, collections
warnings.filterwarnings('ignore')
import numpy as np
case = json.load(open(sys.argv[1]))
folder, expected = sys.argv[2], np.load(sys.argv[3])
from catint.transport import Transport
from catint.comsol_reader import Reader
from catint.units import unit_NA, unit_F
species = collections.OrderedDict((name, dict(d)) for name, d in case['species'])
system = dict(case['system'])
if system.get('active site density') == 'run.py':
    system['active site density'] = 9.61e-05 / unit_NA * (1e10) ** 2
phis = [float(v) for v in expected['phis']]
tp = Transport(species=species, electrode_reactions=case.get('electrode_reactions'), electrolyte_reactions=case.get('electrolyte_reactions'),
               system=system, catmap_args={}, comsol_args={'par_method': 'internal', 'bin_version': '5.3a'}, model_name='CO2R',
               nx=case['nx'], descriptors=collections.OrderedDict([('phiM', phis)]), resultsdir='res')
logging.disable(logging.CRITICAL)
names = list(tp.species.keys())
assert names == [str(s) for s in expected['names']], names
idx = int(expected['index'])
tp.system['phiM'] = phis[idx]                      # the reader files the tables under the CURRENT descriptor values (comsol_reader.py:133-139)
outputs = ['concentrations', 'electrostatics', 'electrode_flux']
Reader(transport=tp, outputs=outputs, comsol_args=tp.comsol_args, results_folder=folder).read_all()
d = tp.alldata[idx]
c = expected['concentration']
for k, sp in enumerate(names):
    assert np.array_equal(np.array(d['species'][sp]['concentration']), c[k]), sp
    assert d['species'][sp]['surface_concentration'] == c[k, 0] and tp.species[sp]['surface_concentration'] == c[k, 0]
    assert d['species'][sp]['electrode_flux'] == expected['flux'][k], (sp, d['species'][sp]['electrode_flux'])
    assert np.allclose(d['species'][sp]['activity_coefficient'], expected['gamma'], rtol=1e-15)
assert np.array_equal(np.array(d['system']['potential']), expected['potential']) and np.array_equal(np.array(d['system']['efield']), expected['efield'])
assert d['system']['surface_potential'] == expected['potential'][0] and d['system']['surface_efield'] == expected['efield'][0]
assert d['system']['Stern_efield'] == expected['efield'][0] * tp.system['epsilon'] / tp.system['Stern epsilon']
nel, nprod = 2, 1                                  # CO2 + H2O + 2 e- -> CO + 2 OH-
assert d['species']['CO']['electrode_current_density'] == expected['flux'][names.index('CO')] * nel * unit_F / nprod / 10.
assert np.allclose(d['system']['pH'], expected['pH'], rtol=1e-13) and abs(d['system']['surface_pH'] - expected['pH'][0]) < 1e-12
assert np.array_equal(tp.xmesh, expected['xmesh']) and tp.nx == len(expected['xmesh'])
print('ok: the reference reader parsed', outputs, 'of descriptor point', idx)
'''


def main():
    from tests.test_results_io import synthetic_sweep
    from catint_amd.results_io import export_comsol_text
    import make_transport_golden as T
    tmp = tempfile.mkdtemp(prefix='catint_cexport_')
    try:
        tp = synthetic_sweep(numeric_flux=True)
        idx = 1
        folder = os.path.join(tmp, 'results')
        export_comsol_text(tp, idx, folder)
        names = list(tp.species.keys())
        d = tp.alldata[idx]
        np.savez(os.path.join(tmp, 'expected.npz'), names=np.array(names), index=idx, phis=np.array(tp.descriptors['phiM'], float),
                 concentration=np.array([d['species'][sp]['concentration'] for sp in names]),
                 flux=np.array([d['species'][sp]['electrode_flux'] for sp in names]),
                 gamma=np.asarray(d['system']['activity_coefficient']), potential=np.asarray(d['system']['potential']),
                 efield=np.asarray(d['system']['efield']), pH=np.asarray(d['system']['pH']), xmesh=np.asarray(tp.xmesh))
        case = dict(next(c for c in T.CASES if c['name'] == 'co2r_numeric_flux'))
        case['nx'] = tp.nx - 2
        json.dump(case, open(os.path.join(tmp, 'case.json'), 'w'))
        with open(os.path.join(tmp, 'reader.py'), 'w') as f:
            f.write(READER)
        env = dict(os.environ, PYTHONPATH=REF, PYTHONHASHSEED='0', PYTHONDONTWRITEBYTECODE='1')
        r = subprocess.run([sys.executable, 'reader.py', 'case.json', folder, 'expected.npz'], cwd=tmp, env=env, capture_output=True, text=True)
        print((r.stdout.strip().splitlines() or ['<no stdout>'])[-1])
        if r.returncode != 0:
            print(r.stderr[-3000:])
            raise SystemExit('the reference reader did not reproduce the exported arrays')
        out = os.path.join(HERE, 'comsol_export')
        shutil.rmtree(out, ignore_errors=True)
        shutil.copytree(folder, out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
