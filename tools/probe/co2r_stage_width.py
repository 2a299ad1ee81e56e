#!/usr/bin/env python3
"""BASELINE configs[2]: width of the continuation stages (tp.newton['dphi_stage']) and minimum number of stages (nramp)."""
import json
import os
import sys

import numpy as np

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'examples'))
import co2r_physical_sweep as ex
from catint_amd.calculator import Calculator


def run(lanes, dphi, nramp):
    tp, phis = ex.build(lanes, 384)
    kin = [{'species': 'CO2', 'rate': ex.tafel_rate(tp), 'stoichiometry': {'CO2': -1.0, 'CO': 1.0, 'OH-': 2.0}}]
    calc = Calculator(transport=tp, calc='comsol')
    tp.newton = {'tol': 1e-8, 'maxit': 80, 'dphi_stage': dphi}
    calc.set_surface_kinetics(kin)
    orig = calc.solve_physical
    calc.solve_physical = lambda s, c0, phiM, flux, **kw: orig(s, c0, phiM, flux, nramp=nramp, **kw)
    best = None
    for _ in range(2):
        calc.newton_iterations_total = 0
        calc.newton_iterations_slowest = 0
        calc.run()
        best = calc.solve_seconds if best is None else min(best, calc.solve_seconds)
    names = list(tp.species)
    cs = np.array([[tp.alldata[i]['species'][sp]['surface_concentration'] for sp in names] for i in range(lanes)])
    return {'dphi_stage': dphi, 'nramp': nramp, 'stages': calc.continuation_stages, 'transport_solve_seconds': best,
            'iterations_total': calc.newton_iterations_total, 'iterations_slowest_lane_summed': calc.newton_iterations_slowest,
            'converged': int((calc.status == 0).sum()), 'retries': len(getattr(calc, 'retry_log', []))}, cs


def main():
    lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    ref, cs0 = run(lanes, 0.2, 8)
    print(json.dumps(ref), flush=True)
    for dphi, nramp in ((0.3, 8), (0.3, 4), (0.4, 4), (0.5, 4), (0.7, 2), (1.0, 2)):
        r, cs = run(lanes, dphi, nramp)
        r['max_rel_diff_surface_concentration'] = float((np.abs(cs - cs0) / (np.abs(cs0) + 1e-30)).max())
        print(json.dumps(r), flush=True)


if __name__ == '__main__':
    main()
