"""Where do device and oracle part ways on fuzz case 117 (3 species, 3 reactions, graded grid, stationary)?  State after k Newton iterations."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import tests.test_gpu_newton as T
from oracle import pnp_physical as PH
d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests', 'golden', 'fuzz', 'newton_case117.json')))
for k in (1, 2, 3, 5, 8, 12, 16, 20, 25, 31, 40):
    kw = dict(d['newton_kw'], maxit=k)
    for kern in ('', 'generic', 'team'):
        os.environ['CATINT_NEWTON_KERNEL'] = kern
        got, ref = T.run_both(d['N'], d['nx'], B=d['B'], seed=d['seed'], newton_kw=kw, reactions=d['reactions'], flux=np.array(d['flux']),
                              x=np.array(d['x']), stationary=True, phi_lo=-0.3, phi_hi=0.3, points_per_debye=d['points_per_debye'])
        c, phi, its, st = got
        rc, rphi, rit = ref
        print('maxit %2d kernel %-8s dc %.2e dphi %.2e its %s ref %s st %s' % (k, kern or 'default', np.abs(c - rc).max() / np.abs(rc).max(), np.abs(phi - rphi).max(), its, rit, st))
