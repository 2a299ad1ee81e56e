#!/usr/bin/env python3
"""Generate golden input/output vectors for the batched 1D PNP path by running the
*actual reference* (sringe/CatINT, read-only at /root/reference) in the dev container.

Dev-only tool: it is never imported by tests, smoke() or bench.py.  Only *numeric data*
(inputs + expected outputs) is written below ``tests/golden/``; no reference source,
patched copy or byte-code ever lands in the repository -- the py3-loadable copy of
``catint/calculator_old.py`` is produced by six mechanical line substitutions (SURVEY.md
App. B) into a throw-away temp dir that is deleted on exit.

What is driven (file:line relative to /root/reference):
  * catint/transport.py:40        Transport(...)  -> D, charges, mu, beta, eps, mesh, c0, flux_bound, pb_bound
  * catint/calculator_old.py:210  Calculator.integrate_pnp(dx,nx,dt,nt,ntout,method)
        'Crank-Nicolson' (:457-564), 'FTCS' (:976-1029), 'odeint'/'dopri5' (:821-973),
        Poisson get_potential_and_gradient (:680-819), get_rates (:159-208), itout (:140-152)
  * catint/transport.py:1373      gouy_chapman known answers, :439-443 Debye length

Usage:  python tests/golden/make_golden.py            (re-creates every fixture)
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))

# -- six mechanical py2->py3 line edits; nothing else is touched ---------
SUBS = [
    (r'^from units import \*', 'from catint.units import *'),
    (r'^from io import sync_mpi,reduce_dict_mpi', 'from catint.catint_io import sync_mpi,reduce_dict_mpi'),
    (r'^from comsol_wrapper import Comsol', 'Comsol=None'),
    (r'^from catmap_wrapper import CatMAP', 'CatMAP=None'),
    (r"print 'Check', np.allclose\(np.dot\(A, x\), b\)", 'pass'),
    (r"print 'going', np.shape\(c\),np.shape\(t\)", 'pass'),
]

# -- the driver that runs inside the scratch dir (our code; talks to the reference API) ------
DRIVER = r'''
import sys, json, logging, warnings
warnings.filterwarnings('ignore')
import numpy as np
case = json.load(open(sys.argv[1]))
out = sys.argv[2]
from catint.transport import Transport
import calc_old_py3 as co

species = None
if case.get('species'):
    import collections
    species = collections.OrderedDict()
    for name, conc in case['species']:
        species[name] = {'bulk_concentration': conc}
system = {'phiM': case['phiM'], 'boundary thickness': case['L']}
system.update(case.get('system', {}))
pb = case['pb_bound']
tp = Transport(species=species, system=system, pb_bound=pb, nx=case['nx'],
               resultsdir='res_' + case['name'],
               comsol_args={'par_method': 'internal', 'bin_version': '5.3a'})
logging.disable(logging.CRITICAL)
# shims for attributes the legacy integrators read but today's Transport no longer sets
tp.use_catmap = False
tp.system['vzeta'] = case.get('vzeta', tp.system['phiM'])
tp.reactions = case.get('reactions', {})
tp.use_reactions = bool(case.get('reactions'))
method = case['method']
tp.calc = method.split('--')[0]
if case.get('init') == 'Gouy-Chapman':
    tp.set_initial_concentrations('Gouy-Chapman')
if case.get('c0_perturb'):
    rng = np.random.default_rng(case['c0_perturb']['seed'])
    tp.c0 = tp.c0 * (1.0 + case['c0_perturb']['amp'] * rng.uniform(-1, 1, size=tp.c0.shape))
for k, v in case.get('flux_bound', []):
    tp.flux_bound[k, 0] = v
if 'use_migration' in case:
    tp.use_migration = case['use_migration']

calc = co.Calculator(transport=tp, calc=method, dt=case['dt'], tmax=case['tmax'],
                     ntout=case['ntout'], desc_method='external')

rhs_samples = None
if case.get('capture_rhs'):
    # grab the method-of-lines RHS closure (calculator_old.py:827) through scipy's odeint seam
    grabbed = {}
    real_odeint = co.integrate.odeint
    def fake_odeint(func, y0, t, args=(), **kw):
        grabbed['f'] = func; grabbed['args'] = args
        return real_odeint(func, y0, t, args=args, **kw)
    co.integrate.odeint = fake_odeint

cout = calc.integrate_pnp(tp.dx, tp.nx, tp.dt, tp.nt, tp.ntout, calc.calc)
cout = np.array(cout)

extra = {}
if case.get('capture_rhs'):
    rng = np.random.default_rng(7)
    states = [tp.c0.copy(), tp.c0 * (1 + 0.2 * rng.uniform(-1, 1, size=tp.c0.shape)), cout[-1].copy()]
    extra['rhs_states'] = np.array(states)
    extra['rhs_values'] = np.array([grabbed['f'](s, 0.0, *grabbed['args']) for s in states])

def pbv(a, b):
    v = tp.pb_bound[a][b]
    return np.nan if v is None else float(v)

gc = np.array([tp.gouy_chapman(x) for x in tp.xmesh]) if tp.nspecies == 2 else np.zeros((0, 2))
np.savez_compressed(out,
    name=case['name'], method=method, nspecies=tp.nspecies,
    species=np.array(list(tp.species.keys())),
    D=tp.D, charges=tp.charges, mu=tp.mu, beta=tp.beta, eps=tp.eps,
    z=np.array([tp.species[sp]['charge'] for sp in tp.species], dtype=np.int64),
    c_bulk=np.array([tp.species[sp]['bulk_concentration'] for sp in tp.species]),
    temperature=tp.system['temperature'], epsilon_r=tp.system['epsilon'],
    debye_length=tp.debye_length, ionic_strength=tp.ionic_strength,
    dx=tp.dx, nx=tp.nx, xmesh=tp.xmesh, xmax=tp.xmax, nx_requested=case['nx'],
    dt=tp.dt, tmax=tp.tmax, nt=tp.nt, ntout=tp.ntout, itout=np.array(tp.itout, dtype=np.int64),
    c0=tp.c0, flux_bound=tp.flux_bound, vzeta=tp.system['vzeta'], phiM=tp.system['phiM'],
    pb_bound=np.array([pbv('potential', 'wall'), pbv('potential', 'bulk'),
                       pbv('gradient', 'wall'), pbv('gradient', 'bulk')]),
    use_migration=tp.use_migration, lax_friedrich=calc.use_lax_friedrich,
    reactions_json=json.dumps(case.get('reactions', {})),
    cout=cout,
    potential=np.array(getattr(tp, 'potential', np.zeros(0))),
    efield=np.array(getattr(tp, 'efield', np.zeros(0))),
    total_charge=np.array(getattr(tp, 'total_charge', np.zeros(0))),
    gouy_chapman=gc, **extra)
print('ok', case['name'], 'nx', tp.nx, 'nt', tp.nt, 'itout', tp.itout, 'cout', cout.shape)
'''

DD = {'potential': {'wall': 'phiM', 'bulk': 0.0}}
DEFAULT_PB = None  # -> transport.py:207-210: wall potential + zero bulk gradient
MIRROR = {'potential': {'bulk': 0.0}, 'gradient': {'wall': 0.0}}

K_CL_HCO3 = [['K+', 30.0], ['Cl-', 10.0], ['HCO3-', 20.0]]
SIX = [['K+', 20.0], ['Na+', 25.0], ['Cl-', 15.0], ['HCO3-', 20.0], ['CO32-', 3.0], ['OH-', 4.0]]

CASES = [
    # --- Crank-Nicolson, Dirichlet-Dirichlet Poisson (primary branch, App. A.1/A.2) ---------
    dict(name='cn_dd_n2_nx50', method='Crank-Nicolson', species=None, phiM=-0.025, L=5e-8, nx=50,
         pb_bound=DD, dt=1e-10, tmax=2e-9, ntout=4),
    dict(name='cn_dd_n2_nx200', method='Crank-Nicolson', species=None, phiM=-0.025, L=1e-7, nx=200,
         pb_bound=DD, dt=1e-9, tmax=1e-8, ntout=2),
    dict(name='cn_dd_n3_nx64_flux', method='Crank-Nicolson', species=K_CL_HCO3, phiM=0.04, L=2e-8, nx=64,
         pb_bound=DD, dt=2e-11, tmax=4e-10, ntout=5, flux_bound=[[0, 1e-3], [2, -2e-3]],
         c0_perturb={'seed': 3, 'amp': 0.05}),
    dict(name='cn_dd_n3_nx127_vzeta', method='Crank-Nicolson', species=K_CL_HCO3, phiM=-0.03, L=3e-8, nx=127,
         pb_bound=DD, dt=5e-12, tmax=5e-11, ntout=2, vzeta=-0.02),
    dict(name='cn_dd_n6_nx40', method='Crank-Nicolson', species=SIX, phiM=-0.03, L=1e-8, nx=40,
         pb_bound=DD, dt=1e-11, tmax=1.2e-10, ntout=3, c0_perturb={'seed': 5, 'amp': 0.02}),
    dict(name='cn_dd_n2_nx50_LF', method='Crank-Nicolson--LF', species=None, phiM=-0.025, L=5e-8, nx=50,
         pb_bound=DD, dt=1e-10, tmax=1e-9, ntout=2),
    # --- Crank-Nicolson, prefix-sum Poisson branches (calculator_old.py:787-803) -------------
    dict(name='cn_defaultpb_n2_nx50_gc', method='Crank-Nicolson', species=None, phiM=-0.02, L=5e-8, nx=50,
         pb_bound=DEFAULT_PB, dt=1e-10, tmax=1e-9, ntout=2, init='Gouy-Chapman'),
    dict(name='cn_mirrorpb_n2_nx50_gc', method='Crank-Nicolson', species=None, phiM=-0.02, L=5e-8, nx=50,
         pb_bound=MIRROR, dt=1e-10, tmax=1e-9, ntout=2, init='Gouy-Chapman'),
    # --- FTCS (App. F.1) ---------------------------------------------------------------------
    dict(name='ftcs_dd_n2_nx50', method='FTCS', species=None, phiM=-0.025, L=5e-8, nx=50,
         pb_bound=DD, dt=1e-12, tmax=4e-11, ntout=4),
    dict(name='ftcs_dd_n3_nx64_flux', method='FTCS', species=K_CL_HCO3, phiM=0.03, L=2e-8, nx=64,
         pb_bound=DD, dt=2e-13, tmax=6e-12, ntout=3, flux_bound=[[1, 5e-4]], c0_perturb={'seed': 11, 'amp': 0.05}),
    dict(name='ftcs_dd_n2_nx50_LF', method='FTCS--LF', species=None, phiM=-0.025, L=5e-8, nx=50,
         pb_bound=DD, dt=1e-12, tmax=2e-11, ntout=2),
    dict(name='ftcs_defaultpb_n2_nx50_gc', method='FTCS', species=None, phiM=-0.02, L=5e-8, nx=50,
         pb_bound=DEFAULT_PB, dt=1e-12, tmax=2e-11, ntout=2, init='Gouy-Chapman'),
    dict(name='ftcs_dd_n3_nx40_rates', method='FTCS', species=K_CL_HCO3, phiM=0.01, L=2e-8, nx=40,
         pb_bound=DD, dt=2e-13, tmax=4e-12, ntout=2, c0_perturb={'seed': 13, 'amp': 0.05},
         reactions={'r1': {'reactants': [['K+', 'Cl-'], ['HCO3-']], 'rates': [3.0e6, 2.0e8]},
                    'r2': {'reactants': [['HCO3-', 'H2O'], ['Cl-']], 'rates': [1.0e8, 5.0e7]},
                    'r3': {'reactants': [['K+'], ['Cl-']]}}),
    dict(name='ftcs_nomig_n2_nx50', method='FTCS', species=None, phiM=-0.025, L=5e-8, nx=50,
         pb_bound=DD, dt=1e-12, tmax=2e-11, ntout=2, use_migration=False,
         c0_perturb={'seed': 17, 'amp': 0.1}),
    # --- method of lines (App. F.2): trajectory + RHS samples -------------------------------
    dict(name='odeint_dd_n2_nx50', method='odeint', species=None, phiM=-0.025, L=5e-8, nx=50,
         pb_bound=DD, dt=1e-10, tmax=1e-9, ntout=2, capture_rhs=True),
    dict(name='dopri5_dd_n2_nx50', method='dopri5', species=None, phiM=-0.025, L=5e-8, nx=50,
         pb_bound=DD, dt=1e-11, tmax=1e-10, ntout=2),
    dict(name='odeint_dd_n3_nx40_flux_LF', method='odeint--LF', species=K_CL_HCO3, phiM=0.02, L=2e-8, nx=40,
         pb_bound=DD, dt=2e-11, tmax=2e-10, ntout=2, flux_bound=[[0, 1e-3]], capture_rhs=True,
         c0_perturb={'seed': 19, 'amp': 0.05}),
]


def main():
    only = set(sys.argv[1:])
    tmp = tempfile.mkdtemp(prefix='catint_golden_')
    try:
        src = open(os.path.join(REF, 'catint', 'calculator_old.py'), encoding='utf8').read()
        n_applied = 0
        for pat, rep in SUBS:
            src, n = re.subn(pat, rep, src, flags=re.M)
            n_applied += n
        assert n_applied == len(SUBS), n_applied
        with open(os.path.join(tmp, 'calc_old_py3.py'), 'w') as f:
            f.write(src)
        with open(os.path.join(tmp, 'drv.py'), 'w') as f:
            f.write(DRIVER)
        env = dict(os.environ, PYTHONPATH=REF + ':' + tmp, OMP_NUM_THREADS='1', PYTHONDONTWRITEBYTECODE='1')
        for case in CASES:
            if only and case['name'] not in only:
                continue
            cj = os.path.join(tmp, case['name'] + '.json')
            json.dump(case, open(cj, 'w'))
            out = os.path.join(HERE, case['name'] + '.npz')
            r = subprocess.run([sys.executable, 'drv.py', cj, out], cwd=tmp, env=env,
                               capture_output=True, text=True)
            tail = (r.stdout.strip().splitlines() or ['<no stdout>'])[-1]
            print(tail)
            if r.returncode != 0:
                print(r.stderr[-3000:])
                raise SystemExit('reference run failed for ' + case['name'])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
