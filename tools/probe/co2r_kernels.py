#!/usr/bin/env python3
"""BASELINE configs[2] (the reference's CO2R example as a 4096-voltage batch, examples/co2r_physical_sweep.py) under each kernel family
of the physical mode: seconds of transport solves, continuation stages, lanes converged, retries.  One JSON line per family.

    python tools/probe/co2r_kernels.py [lanes] [nx] [kernel ...]
"""
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'examples'))


def main():
    lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    nx = int(sys.argv[2]) if len(sys.argv) > 2 else 384
    kernels = sys.argv[3:] or ['team', 'lane4', 'lane2', 'lane', '']
    spec = importlib.util.spec_from_file_location('co2r_physical_sweep', os.path.join(ROOT, 'examples', 'co2r_physical_sweep.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from catint_amd.calculator import Calculator
    stats = []
    if os.environ.get('CO2R_ITER_STATS'):       # per continuation stage: seconds, mean / max Newton iterations over the lanes
        orig = Calculator._continuation

        def traced(self, solver, c0, pb, vz, flux, phiM, start, nst, lanes=None):
            inner = solver.solve_stationary

            def solve(*a, **k):
                t0 = time.perf_counter()
                st = inner(*a, **k)
                dt_ = time.perf_counter() - t0
                it = solver.newton_iterations()
                stats.append((round(dt_ * 1e3, 2), round(float(it.mean()), 2), int(it.max()), int(np.percentile(it, 90))))
                return st
            solver.solve_stationary = solve
            try:
                return orig(self, solver, c0, pb, vz, flux, phiM, start, nst, lanes=lanes)
            finally:
                solver.solve_stationary = inner
        Calculator._continuation = traced
    ref = None
    for kern in kernels + kernels[:1]:
        os.environ['CATINT_NEWTON_KERNEL'] = kern
        tp, phis = mod.build(lanes, nx)
        calc = Calculator(transport=tp, calc='comsol')
        tp.newton = {'tol': 1e-8, 'maxit': 80}
        calc.set_surface_kinetics([{'species': 'CO2', 'rate': mod.tafel_rate(tp), 'stoichiometry': {'CO2': -1.0, 'CO': 1.0, 'OH-': 2.0}}])
        t0 = time.perf_counter()
        calc.run()
        t_all = time.perf_counter() - t0
        j = np.array([tp.alldata[i]['species']['CO']['electrode_current_density'] for i in range(lanes)], dtype=float) if hasattr(tp, 'alldata') else None
        rec = {'kernel': kern or 'auto', 'lanes': lanes, 'nx': int(tp.nx), 'lanes_converged': int((calc.status == 0).sum()),
               'continuation_stages': int(calc.continuation_stages), 'transport_solve_seconds': float(calc.solve_seconds),
               'seconds_total': t_all, 'retry_log': str(getattr(calc, 'retry_log', None))[:300]}
        if j is not None and np.all(np.isfinite(j)):
            if ref is None:
                ref = j
            rec['max_rel_dev_of_CO_current_from_first'] = float(np.abs(j - ref).max() / np.abs(ref).max())
        if stats:
            rec['stages_ms_mean_max_p90'] = list(stats)
            rec['sum_of_max_iterations'] = int(sum(x[2] for x in stats))
            del stats[:]
        print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
