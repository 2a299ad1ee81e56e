import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'examples'))
import numpy as np
import co2r_physical_sweep as E
from catint_amd.calculator import Calculator
ref = None
for dphi in (0.1, 0.2, 0.3, 0.5):
    tp, phis = E.build(1024, 384)
    calc = Calculator(transport=tp, calc='comsol')
    tp.newton = {'tol': 1e-8, 'maxit': 80, 'dphi_stage': dphi}
    calc.set_surface_kinetics([{'species': 'CO2', 'rate': E.tafel_rate(tp), 'stoichiometry': {'CO2': -1.0, 'CO': 1.0, 'OH-': 2.0}}])
    t0 = time.time(); calc.run(); t1 = time.time()
    j = calc.kinetic_flux[:, 3]
    if ref is None: ref = j.copy()
    print('dphi_stage %.1f: stages %d, converged %d/1024, %.2f s, max |j - j_ref|/max j = %.1e' % (dphi, calc.continuation_stages, (calc.status == 0).sum(), t1 - t0, np.abs(j - ref).max() / np.abs(ref).max()))
