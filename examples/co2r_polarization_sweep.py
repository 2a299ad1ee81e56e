#!/usr/bin/env python3
"""Polarization sweep in the shape of BASELINE.json configs[2] / the reference's
examples/02_CO2R_Au_CatMAP/run.py: CO2 reduction on Au in 0.1 M KHCO3, every voltage of the sweep is one
GPU lane, the SCF outer loop (reference catint/calculator.py:294-406) couples a kinetics callback to the
transport solve.

What differs from the reference example, and why:
  * CatMAP is not available offline: a Tafel law stands in for it behind the same seam
    (`flux_callback`, the place where `self.catmap.run()` sits, calculator.py:373);
  * COMSOL is not available: transport is the reference's own finite-difference FTCS integrator
    (calculator_old.py:976-1029) on the MI355X path with `migration: False` (transport.py:316-319), i.e.
    diffusion with flux boundary conditions in an electroneutral electrolyte.  The reference's FD schemes
    lag the potential, which limits dt to the dielectric relaxation time (~1e-10 s) whenever migration is
    on (and `vzeta = 0` makes the wall condition the plain flux condition, calculator_old.py:1003-1006)
    -- useless for a one-second mass-transport transient on the example's 80-micron grid
    (dx ~ 400 Debye lengths), where the double layer is not resolved anyway;
  * inputs are the run.py dictionaries (examples/co2r_inputs.py); the bulk concentrations come out of `Transport`'s own
    Henry / buffer-equilibrium / electroneutrality pass (transport.py:537-768), as in the reference.

    python examples/co2r_polarization_sweep.py --lanes 4096
"""
import argparse
import collections
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from catint_amd.transport import Transport        # noqa: E402
from catint_amd.calculator import Calculator      # noqa: E402
from catint_amd.units import unit_F, unit_R      # noqa: E402
import co2r_inputs                               # noqa: E402  (examples/co2r_inputs.py: the run.py dictionaries)


def build(lanes, nx=200):
    """Transport of the run.py system from the reference's input dictionaries (examples/co2r_inputs.py): species list, bulk
    equilibria and electroneutrality as transport.py:537-768 computes them; diffusion only (see the module docstring)."""
    phis = co2r_inputs.voltages(lanes, -0.5, -0.74)   # beyond ~-0.76 V this Tafel law outruns CO2 transport (negative c)
    tp = Transport(species=co2r_inputs.species(steric=False), electrode_reactions=co2r_inputs.electrode_reactions(),
                   electrolyte_reactions=co2r_inputs.electrolyte_reactions(),
                   system=co2r_inputs.system(phiM=phis[0], migration=False, vzeta=0.0), nx=nx, descriptors={'phiM': phis},
                   model_name='CO2R')
    # the legacy explicit integrator has no use for the buffer's rate constants (up to 1.3e9 1/s against dt = 5e-6 s, and
    # get_rates keeps only the last reaction per species, calculator_old.py:173,:193): the equilibria set the bulk state only
    tp.reactions = collections.OrderedDict()
    tp.use_reactions = False
    return tp, np.array(phis)


def tafel_kinetics(tp):
    """CO2 + H2O + 2e- -> CO + 2 OH-   (educt flux negative, calculator.py:415-432)"""
    names = list(tp.species.keys())
    i_co2, i_co, i_oh = names.index('CO2'), names.index('CO'), names.index('OH-')
    alpha, k0, phi0 = 0.5, 2e-9, -0.11          # transfer coefficient, m/s, V
    beta = 1.0 / (unit_R * tp.system['temperature'])

    def flux(state):
        sc = state['surface_concentration']
        rate = k0 * np.maximum(sc[:, i_co2], 0.0) * np.exp(-alpha * unit_F * beta * (state['phiM'] - phi0))
        f = np.zeros_like(sc)
        f[:, i_co2] = -rate
        f[:, i_co] = rate
        f[:, i_oh] = 2.0 * rate
        return f
    return flux


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--lanes', type=int, default=4096)
    ap.add_argument('--nx', type=int, default=200)
    ap.add_argument('--dt', type=float, default=5e-6)          # explicit: dt <= dx^2/(2 D_max) = 8.6e-6 s
    ap.add_argument('--tmax', type=float, default=0.25)
    ap.add_argument('--tau-scf', type=float, default=0.008)     # run.py:95
    ap.add_argument('--mix-scf', type=float, default=0.02)
    ap.add_argument('--max-iter', type=int, default=150)
    args = ap.parse_args(argv)
    tp, phis = build(args.lanes, args.nx)
    tp.set_calculator('FTCS')
    calc = Calculator(transport=tp, dt=args.dt, tmax=args.tmax, ntout=1, tau_scf=args.tau_scf, mix_scf=args.mix_scf)
    names = list(tp.species.keys())
    nel = np.ones(len(names)); nel[names.index('CO')] = 2
    t0 = time.time()
    out = calc.run_scf_cycle(tafel_kinetics(tp), nel=nel, max_iter=args.max_iter)
    dt = time.time() - t0
    j_co = out['current_density'][:, names.index('CO')]
    print('# %d lanes x %d species x %d points, %d steps per transport solve, %d SCF iterations, %.2f s'
          % (args.lanes, tp.nspecies, tp.nx, tp.nt - 1, out['iterations'], dt))
    print('# converged lanes: %d / %d, failed: %d' % (out['converged'].sum(), args.lanes, out['failed'].sum()))
    print('# phiM [V]   j_CO [mA/cm^2]   surface pH   c_CO2(x=0) [mol/m^3]')
    for i in np.linspace(0, args.lanes - 1, min(args.lanes, 13)).astype(int):
        print('%9.4f  %14.6e  %10.4f  %12.5f' % (phis[i], j_co[i], out['surface_pH'][i],
                                                  out['surface_concentration'][i, names.index('CO2')]))
    return out


if __name__ == '__main__':
    main()
