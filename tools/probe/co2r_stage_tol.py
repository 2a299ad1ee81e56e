#!/usr/bin/env python3
"""BASELINE configs[2] (the CO2R example, 4096 voltages): continuation stages before the operating point solved to a looser tolerance
(tp.newton['stage_tol']), the last stage to the full one -- time of the transport solves, Newton iterations, and the difference of the
final answer to the run with every stage at full tolerance."""
import json
import os
import sys

import numpy as np

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'examples'))
import co2r_physical_sweep as ex
from catint_amd.calculator import Calculator


def run(lanes, stage_tol):
    tp, phis = ex.build(lanes, 384)
    kin = [{'species': 'CO2', 'rate': ex.tafel_rate(tp), 'stoichiometry': {'CO2': -1.0, 'CO': 1.0, 'OH-': 2.0}}]
    calc = Calculator(transport=tp, calc='comsol')
    tp.newton = {'tol': 1e-8, 'maxit': 80}
    if stage_tol:
        tp.newton['stage_tol'] = stage_tol
    calc.set_surface_kinetics(kin)
    best = None
    for _ in range(2):
        calc.newton_iterations_total = 0
        calc.newton_iterations_slowest = 0
        calc.run()
        if best is None or calc.solve_seconds < best:
            best = calc.solve_seconds
    names = list(tp.species)
    cs = np.array([[tp.alldata[i]['species'][sp]['surface_concentration'] for sp in names] for i in range(lanes)])
    return {'stage_tol': stage_tol, 'transport_solve_seconds': best, 'iterations_total': calc.newton_iterations_total,
            'iterations_slowest_lane_summed': calc.newton_iterations_slowest, 'converged': int((calc.status == 0).sum()),
            'stages': calc.continuation_stages}, cs, calc.kinetic_flux.copy()


def main():
    lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    ref, cs0, kf0 = run(lanes, None)
    print(json.dumps(ref), flush=True)
    for tol in (1e-6, 1e-4, 1e-3, 1e-2):
        r, cs, kf = run(lanes, tol)
        r['max_rel_diff_surface_concentration'] = float((np.abs(cs - cs0) / (np.abs(cs0) + 1e-30)).max())
        r['max_rel_diff_kinetic_flux'] = float((np.abs(kf - kf0) / (np.abs(kf0).max() + 1e-300)).max())
        print(json.dumps(r), flush=True)


if __name__ == '__main__':
    main()
