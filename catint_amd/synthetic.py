"""Deterministic synthetic operating-point batches for the BASELINE.json configurations
(SURVEY.md section 8(d)): species from the reference's data/diffusion_constants.txt, per-lane
log-uniform bulk concentrations with the first cation closing charge neutrality
(mirrors transport.py:757-765), per-lane wall potential, uniform grid of L = 40 Debye lengths."""
from dataclasses import dataclass, field

import numpy as np

from .units import unit_R, unit_F, unit_eps0

# name: (z, D [m^2/s])  -- reference data/diffusion_constants.txt
SPECIES_TABLE = {
    'K+': (1, 1.957e-9), 'Na+': (1, 1.334e-9), 'Cs+': (1, 2.056e-9), 'H+': (1, 9.311e-9),
    'Cl-': (-1, 2.032e-9), 'HCO3-': (-1, 1.185e-9), 'CO32-': (-2, 0.923e-9), 'OH-': (-1, 5.273e-9),
    'ClO4-': (-1, 1.792e-9),
}
SPECIES_SETS = {
    2: ['K+', 'HCO3-'],
    3: ['K+', 'Cl-', 'HCO3-'],
    4: ['K+', 'Na+', 'Cl-', 'HCO3-'],
    5: ['K+', 'Na+', 'Cl-', 'HCO3-', 'OH-'],                       # (5, 7: not in SURVEY section 8d's list -- probes of the kernel thresholds)
    6: ['K+', 'Na+', 'Cl-', 'HCO3-', 'CO32-', 'OH-'],
    7: ['K+', 'Na+', 'Cl-', 'HCO3-', 'CO32-', 'OH-', 'Cs+'],
    8: ['K+', 'Na+', 'Cl-', 'HCO3-', 'CO32-', 'OH-', 'Cs+', 'ClO4-'],
}


@dataclass
class SyntheticProblem:
    """Problem-wide fields in the reference's units (same attribute names as oracle.pnp_ref.Problem)."""
    D: np.ndarray
    charges: np.ndarray
    beta: float
    eps: float
    dx: float
    nx: int
    dt: float
    pb: np.ndarray
    vzeta: float = 0.0
    flux_bound: np.ndarray = None
    lax_friedrich: bool = False
    use_migration: bool = True
    reactions: list = field(default_factory=list)
    species: list = field(default_factory=list)

    @property
    def N(self):
        return len(self.D)


def make_batch(B, nspecies, nx, seed=0, phi_max=0.05, temperature=298.14, eps_r=78.36, dt_factor=0.1):
    """Returns (problem, c0[B][N*nx], pb[B][4], vzeta[B], flux[B][N]).

    Poisson BCs: Dirichlet wall (= phiM) / Dirichlet bulk 0 -- the double-layer-forming branch of
    the reference.  c(t=0) = c_bulk (transport.py:1280-1285)."""
    names = SPECIES_SETS[nspecies]
    z = np.array([SPECIES_TABLE[s][0] for s in names], dtype=np.float64)
    D = np.array([SPECIES_TABLE[s][1] for s in names])
    charges = z * unit_F
    beta = 1.0 / (temperature * unit_R)
    eps = eps_r * unit_eps0
    rng = np.random.default_rng(seed)
    cb = np.exp(rng.uniform(np.log(0.1), np.log(100.0), size=(B, nspecies)))
    # first cation closes electroneutrality; keep it positive by boosting it when needed
    rest = (cb[:, 1:] * z[1:]).sum(axis=1)
    need = -rest / z[0]
    bad = need <= 0.05
    if bad.any():
        # add anions' worth of the first anion to flip the sign deterministically
        ia = int(np.where(z < 0)[0][0])
        cb[bad, ia] += (0.1 - need[bad]) * z[0] / (-z[ia])
        rest = (cb[:, 1:] * z[1:]).sum(axis=1)
        need = -rest / z[0]
    cb[:, 0] = need
    # lane-independent reference ionic strength sets the grid (50 mol/m^3 1:1 electrolyte)
    I_ref = 50.0 * unit_F ** 2
    debye = np.sqrt(eps / beta / 2.0 / I_ref)
    L = 40.0 * debye
    dx = L / (nx - 1)
    dt = dt_factor * debye * L / D.max()
    phiM = rng.uniform(-phi_max, phi_max, size=B)
    pb = np.zeros((B, 4))
    pb[:, 0] = phiM
    pb[:, 2:] = np.nan
    c0 = np.repeat(cb[:, :, None], nx, axis=2).reshape(B, nspecies * nx)
    flux = np.zeros((B, nspecies))
    prob = SyntheticProblem(D=D, charges=charges, beta=beta, eps=eps, dx=dx, nx=nx, dt=dt, pb=pb[0].copy(),
                            flux_bound=np.zeros(nspecies), species=names)
    return prob, c0, pb, phiM.copy(), flux
