"""Host-side mirror of the part of the reference's ``Transport`` state bag that feeds the
finite-difference transport path (reference: catint/transport.py, class Transport :38-1521).

Only what the legacy integrators read is reproduced; reaction
networks, CatMAP/COMSOL plumbing, logging folders and pickling are out of scope.  Attribute
names, units and arithmetic order follow the reference so that the numbers are identical:

  charges = z*F                       transport.py:1240-1276 (symbol_reader)
  eps, beta                           :311-312
  D, mu = D*charges*beta              :423-436
  ionic_strength, debye_length        :439-443
  mesh (xmax, dx, xmesh, nx)          :449-460   (nx is re-read from the mesh, as in the reference)
  c0 (flat species-major)             :1396-1412
  flux_bound, pb_bound                :1435-1475, :1296-1311
  descriptors / alldata_names         :1135-1195
"""
import collections
import itertools

import numpy as np

from .units import unit_R, unit_F, unit_eps0, unit_NA

# name: (symbol, D [m^2/s]) -- CRC-handbook values as tabulated by the reference (data/diffusion_constants.txt)
SPECIES_DATA = {
    'H2': ('H_2', 5.11e-9), 'CO2': ('CO_2', 1.91e-9), 'CO': ('CO', 2.23e-9), 'O2': ('O_2', 2.42e-9),
    'H3PO4': ('H_3PO_4', 8.8e-10), 'H2PO4-': ('H_2PO_4^-', 9.59e-10), 'HPO42-': ('HPO_4^{2-}', 7.59e-10),
    'PO43-': ('PO_4^{3-}', 8.24e-10), 'HCO3-': ('HCO_3^-', 1.185e-9), 'CO32-': ('CO_3^{2-}', 0.923e-9),
    'Cs+': ('Cs^+', 2.056e-9), 'D+': ('D^+', 6.655e-9), 'H+': ('H^+', 9.311e-9), 'K+': ('K^+', 1.957e-9),
    'Na+': ('Na^+', 1.334e-9), 'NH4+': ('NH_4^+', 1.957e-9), 'Li+': ('Li^+', 1.029e-9), 'Ca2+': ('Ca^{2+}', 0.792e-9),
    'OH-': ('OH^-', 5.273e-9), 'Cl-': ('Cl^-', 2.032e-9), 'I-': ('I^-', 2.045e-9), 'Br-': ('Br^-', 2.080e-9),
    'ClO4-': ('ClO_4^-', 1.792e-9), 'CH4': ('CH_4', 1.49e-9), 'C2H4': ('C_2H_4', 1.51e-9),
}

SYSTEM_DEFAULTS = {           # transport.py:213-230 (the keys the FD path reads)
    'epsilon': 78.36, 'temperature': 298.14, 'phiM': 0.0, 'phiPZC': 0.0, 'Stern capacitance': 18e-2,
    'pressure': 1, 'exclude species': ['H2O', 'e-'],
}


class TransportError(ValueError):
    """Raised where the reference logs an error and calls sys.exit()."""


def charge_from_symbol(symbol):
    """transport.py:1240-1276: 'K^+' -> 1, 'CO_3^{2-}' -> -2, 'CO_2' -> 0."""
    parts = symbol.split('^')
    if len(parts) == 1:
        return 0
    s = parts[-1].replace('{', '').replace('}', '')
    if s[-1] == '-':
        return -int(s[:-1]) if len(s) > 1 else -1
    if s[-1] == '+':
        return int(s[:-1]) if len(s) > 1 else 1
    return int(s)


class Transport(object):
    def __init__(self, species=None, system=None, pb_bound=None, nx=100, descriptors=None, model_name=None):
        if species is None:     # reference defaults, transport.py:185-194
            species = collections.OrderedDict([
                ('species1', {'symbol': r'K^+', 'name': 'potassium', 'diffusion': 1.96e-9, 'kind': 'electrolyte',
                              'bulk_concentration': 0.001 * 1000.}),
                ('species2', {'symbol': r'HCO_3^-', 'name': 'bicarbonate', 'diffusion': 1.2e-9, 'kind': 'electrolyte',
                              'bulk_concentration': 0.001 * 1000.})])
        self.species = collections.OrderedDict((k, dict(v)) for k, v in species.items())
        self.system = dict(SYSTEM_DEFAULTS)
        if system is not None:
            self.system.update(system)
        for es in self.system['exclude species']:
            self.species.pop(es, None)
        for sp in self.species:
            d = self.species[sp]
            if 'diffusion' not in d:
                if sp not in SPECIES_DATA:
                    raise TransportError('No diffusion constant for {}. Provide it as an input'.format(sp))
                d['diffusion'] = SPECIES_DATA[sp][1]
            if 'symbol' not in d:
                if sp not in SPECIES_DATA:
                    raise TransportError('No symbol (charge) known for {}'.format(sp))
                d['symbol'] = SPECIES_DATA[sp][0]
            d.setdefault('bulk_concentration', 0.0)
            d['charge'] = charge_from_symbol(d['symbol'])
            d.setdefault('flux', 0.0)
            d.setdefault('surface_concentration', d['bulk_concentration'])
        self.nspecies = len(self.species)
        self.charges = np.array([self.species[sp]['charge'] * unit_F for sp in self.species])
        self.eps = self.system['epsilon'] * unit_eps0
        self.beta = 1. / (self.system['temperature'] * unit_R)
        self.use_migration = bool(self.system.get('migration', True))
        self.D = np.array([self.species[sp]['diffusion'] for sp in self.species])
        self.mu = self.D * self.charges * self.beta
        self.ionic_strength = 0.0
        for isp, sp in enumerate(self.species):
            self.ionic_strength += self.charges[isp] ** 2 * self.species[sp]['bulk_concentration']
        self.ionic_strength *= 0.5
        with np.errstate(divide='ignore'):
            self.debye_length = np.sqrt(self.eps / self.beta / 2. / self.ionic_strength)
        # mesh, transport.py:449-460
        self.nx = nx
        if 'boundary thickness' in self.system:
            self.xmax = self.system['boundary thickness']
            self.dx = self.xmax / (self.nx * 1.)
        else:
            nx_mod = max(1., np.ceil(self.nx / 10.))
            self.xmax = self.debye_length * nx_mod
            self.dx = self.debye_length / nx_mod
        self.xmesh = np.arange(0, self.xmax + self.dx, self.dx)
        self.nx = len(self.xmesh)
        self.mesh_uniform = True
        # initial / boundary conditions
        self.c0 = np.repeat([self.species[sp]['bulk_concentration'] for sp in self.species], self.nx).astype(float)
        if any(isinstance(self.species[sp]['flux'], str) for sp in self.species):
            # the reference creates no flux_bound when a flux is symbolic ('catmap', equations): SURVEY App. E
            raise TransportError('symbolic fluxes need a flux callback (Calculator.run_scf_cycle), not the FD path')
        self.flux_bound = np.zeros([self.nspecies, 2])
        self.flux_bound[:, 0] = [self.species[sp]['flux'] for sp in self.species]
        if pb_bound is None:    # transport.py:207-210
            pb_bound = {'potential': {'wall': 'phiM'}, 'gradient': {'bulk': 0.0}}
        self.pb_bound = {}
        for key1 in ['potential', 'gradient']:
            self.pb_bound[key1] = {}
            for key2 in ['wall', 'bulk']:
                v = pb_bound.get(key1, {}).get(key2, None)
                self.pb_bound[key1][key2] = self.system['phiM'] if isinstance(v, str) and v == 'phiM' else v
        self._pb_symbolic = {k1: {k2: pb_bound.get(k1, {}).get(k2, None) for k2 in ('wall', 'bulk')}
                             for k1 in ('potential', 'gradient')}
        # vzeta is read by the legacy integrators' wall condition (calculator_old.py:529, :1003) but is not a key of
        # today's Transport: unless the caller sets it, it follows phiM (also per lane in a phiM sweep)
        self.vzeta_follows_phiM = 'vzeta' not in self.system
        self.system.setdefault('vzeta', self.system['phiM'])
        self.reactions = {}
        self.use_reactions = False
        self.calc = None
        self.initialize_descriptors(descriptors)
        self.alldata = [{'species': {}, 'system': {}} for _ in self.alldata_names]

    # -- transport.py:1135-1195 ------------------------------------------------------------------
    def initialize_descriptors(self, descriptors):
        if descriptors is None:
            descriptors = {'phiM': [self.system['phiM']]}
        descriptors = collections.OrderedDict(descriptors)
        if len(descriptors) > 2:
            raise TransportError('Only two descriptors are supported')
        if len(descriptors) == 1:   # dummy second descriptor, :1157-1163
            descriptors['temperature'] = [self.system['temperature']]
        self.descriptors = descriptors
        keys = list(descriptors.keys())
        self.alldata_names = [[v1, v2] for v1, v2 in itertools.product(descriptors[keys[0]], descriptors[keys[1]])]

    def pb_array(self, system=None):
        """[potential wall, potential bulk, gradient wall, gradient bulk] with NaN for None; 'phiM' entries
        follow the (per-lane) system dict."""
        system = self.system if system is None else system
        out = []
        for k1 in ('potential', 'gradient'):
            for k2 in ('wall', 'bulk'):
                v = self._pb_symbolic[k1][k2]
                if isinstance(v, str) and v == 'phiM':
                    v = system['phiM']
                out.append(np.nan if v is None else float(v))
        return np.array(out)

    # -- transport.py:1373-1383 ------------------------------------------------------------------
    def gouy_chapman(self, x, phiM=None):
        if phiM is None:
            phiM = self.system['phiM']

        def func(x):
            term1 = 1. + np.tanh(phiM * self.beta * unit_F / 4.) * np.exp(-1. / self.debye_length * x)
            term2 = 1. - np.tanh(phiM * self.beta * unit_F / 4.) * np.exp(-1. / self.debye_length * x)
            return 2. / (self.beta * abs(self.charges[0])) * np.log(term1 / term2)
        grad = (func(x + 1e-10) - func(x - 1e-10)) / (2 * 1e-10)
        return func(x), grad

    def set_initial_concentrations(self, func, phiM=None):
        """transport.py:1325-1346 ('Gouy-Chapman' Boltzmann profile)."""
        if func != 'Gouy-Chapman' or self.nspecies != 2:
            raise TransportError('Gouy-Chapman limit only implemented for two species')
        c0 = np.zeros(self.nspecies * self.nx)
        for k, sp in enumerate(self.species):
            for i in range(self.nx):
                c0[k * self.nx + i] = self.species[sp]['bulk_concentration'] * \
                    np.exp(-self.beta * self.charges[k] * self.gouy_chapman(self.xmesh[i], phiM=phiM)[0])
        self.c0 = c0

    def set_graded_mesh(self, h0):
        """Replace the uniform mesh by a geometric one with the same end point and number of points and first spacing h0
        (physical mode only; the legacy FD integrators assume a constant dx).  tp.dx becomes h0."""
        from .host import graded_mesh
        self.xmesh = graded_mesh(self.xmesh[-1], h0, self.nx)
        self.dx = float(self.xmesh[1] - self.xmesh[0])
        self.mesh_uniform = False

    def set_calculator(self, calc=None):   # transport.py:1514
        self.calc = calc
