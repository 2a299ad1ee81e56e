import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, importlib.util
spec = importlib.util.spec_from_file_location('co2r', os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'),'examples/co2r_polarization_sweep.py')); mod=importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
out = mod.main(['--lanes','8','--tmax','0.01','--dt','5e-6','--mix-scf','0.3','--tau-scf','1e-3','--max-iter','60'])
print('acc', out['accuracy'])
print('sc min per lane', out['surface_concentration'].min(axis=1))
print('sc lane0', out['surface_concentration'][0])
print('hist tail', out['history'][-3:])
