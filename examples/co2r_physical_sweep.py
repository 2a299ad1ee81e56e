#!/usr/bin/env python3
"""BASELINE.json configs[2] in the physical mode: the system of the reference's examples/02_CO2R_Au_CatMAP/run.py
(CO2 reduction on Au in 0.1 M KHCO3) solved the way its production path asks COMSOL to -- stationary size-modified
Poisson-Nernst-Planck with the bicarbonate/water buffer reactions, a Stern layer and flux boundary conditions -- with every
voltage of the polarization sweep as one GPU lane (4096 lanes in the BASELINE configuration).

  * inputs: the run.py dictionaries themselves (examples/co2r_inputs.py); `Transport` derives species, bulk concentrations
    (Henry's law, buffer equilibria, electroneutrality), ion size, Stern capacitance, phiPZC, boundary layer thickness and the
    homogeneous reactions with their rate constants exactly as the reference does (transport.py:537-768, catint/data.py);
    H2O is an excluded species (constant activity), so `H2O <-> OH- + H+` has a constant forward rate;
  * CatMAP is not available offline: first-order Tafel kinetics `CO2 + H2O + 2e- -> CO + 2 OH-` stand in for it.  They are
    coupled implicitly (Calculator.set_surface_kinetics): the whole polarization curve is ONE batched Newton solve instead of
    hundreds of kinetics<->transport SCF iterations (the SCF loop is available too: --scf);
  * the mesh is graded towards the electrode (first cell lambda_D/20, 80 micron domain), as the reference's COMSOL mesh is
    (comsol_model.py:588,593) -- a uniform 201-point grid has dx = 400 Debye lengths.

    python examples/co2r_physical_sweep.py --lanes 4096
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


def build(lanes, nx, phimin=-0.5, phimax=-2.0):
    """Transport of the run.py system: the reference's input dictionaries (examples/co2r_inputs.py) in, species list, bulk
    equilibria, electroneutrality, reaction table out (transport.py:537-768); then the wall-graded mesh."""
    phis = co2r_inputs.voltages(lanes, phimin, phimax)
    tp = Transport(species=co2r_inputs.species(), electrode_reactions=co2r_inputs.electrode_reactions(),
                   electrolyte_reactions=co2r_inputs.electrolyte_reactions(), system=co2r_inputs.system(phiM=phis[0]), nx=nx - 1,
                   descriptors={'phiM': phis}, model_name='CO2R')
    tp.set_graded_mesh(tp.debye_length / 20.0)
    return tp, np.array(phis)


def tafel_rate(tp, alpha=0.5, k0=4e-9, phi0=-0.11):
    beta = 1.0 / (unit_R * tp.system['temperature'])
    return lambda phiM: k0 * np.exp(-alpha * unit_F * beta * (phiM - phi0))      # m/s, first order in c_CO2(x=0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lanes', type=int, default=256)
    ap.add_argument('--nx', type=int, default=384)
    ap.add_argument('--scf', action='store_true', help='also run the SCF outer loop (kinetics callback on the host) for comparison')
    ap.add_argument('--scf-device', action='store_true', help='also run the SCF outer loop on the device (pnp_scf_cycle)')
    a = ap.parse_args()
    tp, phis = build(a.lanes, a.nx)
    rate = tafel_rate(tp)
    kin = [{'species': 'CO2', 'rate': rate, 'stoichiometry': {'CO2': -1.0, 'CO': 1.0, 'OH-': 2.0}}]
    calc = Calculator(transport=tp, calc='comsol')
    tp.newton = {'tol': 1e-8, 'maxit': 80}
    calc.set_surface_kinetics(kin)
    t0 = time.time()
    calc.run()
    t1 = time.time()
    names = list(tp.species.keys())
    j = calc.kinetic_flux[:, names.index('CO')] * 2 * unit_F / 10.0            # mA/cm^2 (nel = 2), comsol_reader.py:241-246
    ok = calc.status == 0
    jlim = tp.D[names.index('CO2')] * 33.429 / tp.xmesh[-1] * 2 * unit_F / 10.0
    print('%d lanes x %d species x %d points (graded, h0 = %.2e m ... %.2e m): %d converged, %d continuation stages, '
          '%.2f s in the transport solves, %.2f s in total (incl. the per-lane result dictionaries)'
          % (a.lanes, tp.nspecies, tp.nx, tp.xmesh[1], tp.xmesh[-1] - tp.xmesh[-2], ok.sum(), calc.continuation_stages,
             calc.solve_seconds, t1 - t0))
    print('phiM [V]   j_CO [mA/cm2]   c_CO2(0)   pH(0)   phi(0) [V]   c_K+(0)')
    for i in np.linspace(0, a.lanes - 1, min(a.lanes, 9)).astype(int):
        d = tp.alldata[i]
        ph = 14 + np.log10(max(d['species']['OH-']['surface_concentration'], 1e-300) / 1000.0)
        print('%7.3f   %12.5f   %8.4f   %5.2f   %9.4f   %8.1f' % (phis[i], j[i], d['species']['CO2']['surface_concentration'], ph,
                                                                d['system']['surface_potential'], d['species']['K+']['surface_concentration']))
    print('diffusion-limited CO2 current without buffer regeneration: %.3f mA/cm2' % jlim)
    if a.scf_device:
        tp3, _ = build(a.lanes, a.nx)
        tp3.newton = tp.newton
        scf = Calculator(transport=tp3, calc='comsol', tau_scf=0.008, mix_scf=0.02)         # run.py:95
        scf.set_surface_kinetics(kin)
        t0 = time.time()
        out = scf.run_scf_cycle(nel=[1, 2, 1, 2, 1, 1, 1], max_iter=3000)
        print('SCF loop on the device: %d iterations, %d/%d lanes converged, %.2f s; max |j_scf - j_implicit| / j = %.2e'
              % (out['iterations'], out['converged'].sum(), a.lanes, time.time() - t0,
                 np.abs(out['flux'][out['converged'], names.index('CO')] * 2 * unit_F / 10.0 - j[out['converged']]).max()
                 / max(np.abs(j).max(), 1e-300)))
    if a.scf:
        tp2, _ = build(a.lanes, a.nx)
        tp2.newton = tp.newton
        scf = Calculator(transport=tp2, calc='comsol', tau_scf=0.008, mix_scf=0.02)         # run.py:95

        def flux_cb(state):
            f = np.zeros((a.lanes, tp2.nspecies))
            r = rate(state['phiM']) * np.maximum(state['surface_concentration'][:, names.index('CO2')], 0.0)
            f[:, names.index('CO2')] = -r
            f[:, names.index('CO')] = r
            f[:, names.index('OH-')] = 2 * r
            return f
        t0 = time.time()
        out = scf.run_scf_cycle(flux_cb, nel=[1, 2, 1, 2, 1, 1, 1], max_iter=3000)
        print('SCF loop: %d iterations, %d/%d lanes converged, %.2f s; max |j_scf - j_implicit| / j = %.2e'
              % (out['iterations'], out['converged'].sum(), a.lanes, time.time() - t0,
                 np.abs(out['flux'][out['converged'], names.index('CO')] * 2 * unit_F / 10.0 - j[out['converged']]).max()
                 / max(np.abs(j).max(), 1e-300)))


if __name__ == '__main__':
    main()
