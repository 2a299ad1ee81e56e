#!/usr/bin/env python3
"""Iterates of the REFERENCE's own SCF loop -- `Calculator.run_scf_cycle` and `evaluate_accuracy` of catint/calculator.py:260-406 -- with an
analytic kinetics / transport pair standing where CatMAP and COMSOL sit, as data: per iteration the mixed surface concentrations the
kinetics saw, the fluxes they returned, what the transport answered, the mixing factor; per case the iteration count and the final state.
tests/test_host_scf.py asserts the batched host loop (catint_amd/calculator.py:run_scf_cycle) against these numbers; the device loop
(pnp_scf_cycle) is asserted equal to the host loop in tests/test_gpu_calculator.py.

`catint.calculator` imports `catmap` (third party, not available offline).  The script puts an EMPTY stand-in module of that name in
sys.modules only so that the import statement succeeds -- nothing of CatMAP is emulated: the loop's `self.catmap.run` and
`self.run_single_step` are replaced by the two analytic functions below before it runs, and no file of the reference is edited.

Dev-only: runs the reference (read-only at /root/reference) in a scratch dir; writes numbers only, to tests/golden/scf_cycle.json.
Usage:  python tests/golden/make_scf_golden.py
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
import sys, json, types, logging, warnings, collections
warnings.filterwarnings('ignore')
import numpy as np
# stand-ins so that `from catmap.model import ReactionModel` / `from catmap import analyze` (catmap_wrapper.py:11-12) import
cm = types.ModuleType('catmap'); cmm = types.ModuleType('catmap.model'); cma = types.ModuleType('catmap.analyze')
cmm.ReactionModel = object
cm.model, cm.analyze = cmm, cma
sys.modules.update({'catmap': cm, 'catmap.model': cmm, 'catmap.analyze': cma})
from catint.transport import Transport
from catint.calculator import Calculator

case = json.load(open(sys.argv[1]))
species = collections.OrderedDict((name, dict(d)) for name, d in case['species'])
tp = Transport(species=species, system=dict(case['system']), nx=case['nx'], pb_bound=case['pb_bound'],
               electrode_reactions=case.get('electrode_reactions'),
               descriptors=collections.OrderedDict((k, list(v)) for k, v in case['descriptors']),
               comsol_args={'par_method': 'internal', 'bin_version': '5.3a'}, resultsdir='res')
logging.disable(logging.CRITICAL)
tp.use_catmap = False                 # (no CatMAP object is built; the loop's kinetics call is replaced below)
names = list(tp.species.keys())
D = np.array(case['D'])
L = case['system']['boundary thickness']
cb = np.array([tp.species[sp]['bulk_concentration'] for sp in names])
kin = case['kinetics']


def kinetics(sc, phiM):
    j = kin['k0'] * max(sc[kin['educt']], 0.0) * np.exp(-kin['alpha'] * (phiM - kin['phi0']))
    return np.array(kin['nu']) * j


out = {'name': case['name'], 'species_order': names, 'bulk': cb.tolist(), 'lanes': []}
for phiM in case['phis']:
    calc = Calculator(transport=tp, calc='Crank-Nicolson', dt=1e-3, tmax=1e-2, ntout=1, tau_scf=case['tau_scf'], mix_scf=case['mix_scf'])
    tp.system['phiM'] = phiM
    tp.system[list(tp.descriptors.keys())[1]] = tp.descriptors[list(tp.descriptors.keys())[1]][0]
    for k, sp in enumerate(names):
        tp.species[sp]['surface_concentration'] = float(case['sc0'][k]) if case.get('sc0') else float(cb[k])
        tp.species[sp]['flux'] = 0.0
    trace = []

    def catmap_run():
        sc = np.array([tp.species[sp]['surface_concentration'] for sp in names], float)
        fl = kinetics(sc, phiM)
        for k, sp in enumerate(names):
            tp.species[sp]['flux'] = float(fl[k])
        trace.append({'sc_in': sc.tolist(), 'flux': fl.tolist(), 'mix': float(calc.mix_scf), 'surface_pH': float(tp.system.get('surface_pH', np.nan))})

    def run_single_step(label=''):
        fl = np.array([tp.species[sp]['flux'] for sp in names], float)
        sc = cb + fl * L / D                 # diffusion-layer algebra standing in for the PDE solve: may go negative
        for k, sp in enumerate(names):
            tp.species[sp]['surface_concentration'] = float(sc[k])
        trace[-1]['sc_out'] = sc.tolist()
        if len(trace) > case['max_iter']:
            raise RuntimeError('SCF did not converge')

    calc.catmap = types.SimpleNamespace(run=catmap_run)
    calc.run_single_step = run_single_step
    converged = True
    try:
        calc.run_scf_cycle(label='x')
    except RuntimeError:              # (the iteration cap of this script: the reference's loop has none)
        converged = False
    out['lanes'].append({'phiM': phiM, 'iterations': len(trace), 'converged': converged, 'final_mix': float(calc.mix_scf),
                         'final_sc': [float(tp.species[sp]['surface_concentration']) for sp in names],
                         'final_flux': [float(tp.species[sp]['flux']) for sp in names],
                         'final_surface_pH': float(tp.system['surface_pH']),
                         'trace_head': trace[:6], 'trace_tail': trace[-2:], 'trace_41_42': trace[40:42],
                         'mix_changes': [[i + 1, t['mix']] for i, t in enumerate(trace) if i == 0 or t['mix'] != trace[i - 1]['mix']],
                         'negative_iterations': [i + 1 for i, t in enumerate(trace) if min(t['sc_out']) < 0]})
json.dump(out, open(sys.argv[2], 'w'), indent=1)
print('ok', case['name'], [l['iterations'] for l in out['lanes']])
'''

SPECIES = [['K+', {'bulk_concentration': 100.0}], ['OH-', {'bulk_concentration': 1e-4}],
           ['CO2', {'bulk_concentration': 33.0, 'diffusion': 1.91e-9, 'symbol': 'CO_2'}],
           ['CO', {'bulk_concentration': 0.0, 'diffusion': 2.23e-9, 'symbol': 'CO', 'flux': 'catmap'}]]      # (as in run.py:36)
BASE = dict(species=SPECIES, system={'phiM': -0.5, 'boundary thickness': 8e-5, 'bulk_pH': 6.8}, nx=50,
            pb_bound={'potential': {'wall': 'phiM', 'bulk': 0.0}}, D=[1.957e-9, 5.273e-9, 1.91e-9, 2.23e-9],
            electrode_reactions={'CO': {'reaction': 'CO2 + H2O + 2 e- -> CO + 2 OH-'}},
            kinetics={'k0': 1e-7, 'educt': 2, 'alpha': 8.0, 'phi0': -0.5, 'nu': [0.0, 2.0, -1.0, 1.0]}, max_iter=5000)
CASES = [
    # A: the Tafel ladder of tests/test_host_scf.py: seven potentials, lanes converge after different iteration counts
    dict(BASE, name='tafel_mix03', phis=[-0.5, -0.6, -0.7, -0.8, -0.9, -1.0, -1.1], descriptors=[['phiM', [-0.5, -0.6, -0.7, -0.8, -0.9, -1.0, -1.1]]],
         tau_scf=1e-6, mix_scf=0.3),
    # B: run.py's mixing (0.02) and tolerance (0.008): slow enough that the 40-step mix decay (calculator.py:319-323) fires
    dict(BASE, name='tafel_runpy_mixing', phis=[-0.9, -1.1], descriptors=[['phiM', [-0.9, -1.1]]], tau_scf=0.008, mix_scf=0.02),
    # C: kinetics fast enough that the transport answers with a NEGATIVE CO2 surface concentration: the fallback to the previous
    #    iterate (:328-338), the 1e-20 clamp of the first two iterations (:341-344) and the loop condition `any(sc < 0)` (:316)
    dict(BASE, name='negative_surface_concentration', phis=[-1.2, -1.25, -1.3], descriptors=[['phiM', [-1.2, -1.25, -1.3]]], tau_scf=1e-5,
         mix_scf=0.3),
    # D: slow mixing and a tight tolerance: more than 40 iterations, the mixing factor decays by 0.9 every 40 steps (:319-323)
    dict(BASE, name='slow_mixing_decay', phis=[-1.0], descriptors=[['phiM', [-1.0]]], tau_scf=1e-9, mix_scf=0.05),
    # E: a lane whose transport answer is negative from the third iteration on: the loop falls back to the previous iterate (:332-334),
    #    which reproduces the same answer -- it never leaves; recorded up to 100 iterations (two decays of the mixing factor)
    dict(BASE, name='stuck_on_the_negative_fallback', phis=[-1.4], descriptors=[['phiM', [-1.4]]], tau_scf=1e-5, mix_scf=0.3, max_iter=100),
]


def main():
    tmp = tempfile.mkdtemp(prefix='catint_scfgolden_')
    try:
        with open(os.path.join(tmp, 'drv.py'), 'w') as f:
            f.write(DRIVER)
        env = dict(os.environ, PYTHONPATH=REF, PYTHONHASHSEED='0', PYTHONDONTWRITEBYTECODE='1', OMP_NUM_THREADS='1')
        outs = []
        for case in CASES:
            cj, oj = os.path.join(tmp, case['name'] + '.json'), os.path.join(tmp, case['name'] + '.out.json')
            json.dump(case, open(cj, 'w'))
            r = subprocess.run([sys.executable, 'drv.py', cj, oj], cwd=tmp, env=env, capture_output=True, text=True)
            print((r.stdout.strip().splitlines() or ['<no stdout>'])[-1])
            if r.returncode != 0:
                print(r.stderr[-3000:])
                raise SystemExit('reference run_scf_cycle failed for ' + case['name'])
            outs.append({'input': case, 'reference': json.load(open(oj))})
        json.dump(outs, open(os.path.join(HERE, 'scf_cycle.json'), 'w'), indent=1, sort_keys=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
