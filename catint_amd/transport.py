"""Host-side mirror of the reference's ``Transport`` state bag (reference: catint/transport.py, class Transport :38-1521):
everything the transport solve reads from ``tp`` -- species list incl. reaction-derived additions, bulk concentrations from
buffer equilibria / Henry's law / electroneutrality, charges, D, mu, Debye length, mesh, initial state, wall fluxes closed by
the electrode-reaction stoichiometry, Poisson boundary slots, reaction tables, descriptors.  Same constructor arguments,
attribute names, units and arithmetic order as the reference, so the numbers are identical (tests/test_host_transport*.py
against reference-generated fixtures).  Out of scope: results folders, log files, MPI, COMSOL/CatMAP argument plumbing
(``comsol_args`` / ``catmap_args`` are stored untouched).

  species completion                   transport.py:537-601   (_complete_species)
  Henry's law, charges                 :603-639, :1240-1276
  buffer equilibria (fsolve)           :641-735               (_solve_buffer_equilibria)
  electroneutrality closure            :743-768
  pH, activity coefficients            :264-304
  eps, beta, D, mu, Debye length, mesh :311-460
  reaction tables                      :325-395, :1098-1132   (parse_reaction_table)
  educt / product / electrolyte lists  :397-420
  wall fluxes                          :929-1095              (_initialize_fluxes)
  c0, pb_bound, flux/dc_dt/efield bounds  :1277-1487
  descriptors / alldata                :1135-1195
"""
import collections
import itertools
import re

import numpy as np

from .units import unit_R, unit_F, unit_eps0, unit_NA, unit_T
from .data import SPECIES_TABLE, HENRY_CONSTANTS, ELECTROLYTE_REACTIONS

# name: (symbol, D [m^2/s]) -- kept for callers that only need the diffusion table
SPECIES_DATA = collections.OrderedDict((k, (v[1], v[2])) for k, v in SPECIES_TABLE.items())

SPECIES_KEYS = ['bulk_concentration', 'diffusion', 'name', 'symbol', 'flux', 'current density', 'flux-equation', 'MPB_radius',
                'catmap_symbol', 'Henry constant',
                'kind', 'charge', 'surface_concentration']       # last three: tolerated extras (defaults dict / round trips)
SYSTEM_KEYS = ['phiM', 'Stern capacitance', 'Stern epsilon', 'bulk_pH', 'phiPZC', 'temperature', 'pressure', 'water viscosity',
               'electrolyte viscosity', 'epsilon', 'migration', 'field dependence', 'electrode reactions', 'electrolyte reactions',
               'boundary thickness', 'exclude species', 'active site density', 'current density', 'flow rate', 'RF',
               'potential drop', 'Stern_efield', 'charging_scheme', 'use_activities', 'Stern_potential', 'init_folder',
               'vzeta', 'wall potential']                        # last two: this package's extensions (legacy FD wall term; Dirichlet wall)
SYSTEM_DEFAULTS = collections.OrderedDict([      # transport.py:213-230
    ('epsilon', 78.36), ('Stern epsilon', 2.0), ('Stern capacitance', 18e-2), ('temperature', 298.14), ('phiM', 0.0),
    ('phiPZC', 0.0), ('electrode reactions', False), ('electrolyte reactions', False), ('exclude species', ['H2O', 'e-']),
    ('pressure', 1), ('field dependence', None), ('init_folder', None), ('Stern_efield', 0.0), ('charging_scheme', 'comsol'),
    ('use_activities', True), ('Stern_potential', 0.0), ('potential drop', 'Stern'),
])

_SPECIES_IN_TERM = re.compile(r'([a-zA-Z]{1,10}[a-zA-Z-+0-9]+)')        # transport.py:552,:580
_SPECIES_IN_EQUILIBRIUM = re.compile(r'([a-zA-Z]{0,10}[a-zA-Z-+0-9]+)')  # :679
_FLUX_KEYS = ('flux', 'current density', 'flux-equation')


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


def species_of_reaction_string(text):
    """Every species named in 'A + 2 B <-> C' / 'A + 2 e- -> B' in order of appearance (transport.py:550-553, :578-581)."""
    terms = sum([side.split(' + ') for side in text.split('->')], [])
    return [_SPECIES_IN_TERM.findall(t.strip())[0] for t in terms]


def parse_reaction_table(reactions):
    """initialize_reactions (transport.py:1098-1132): the 'reaction' string of every entry becomes [[educts], [products]] with
    stoichiometric coefficients expanded into repeats ('2 OH-' -> 'OH-', 'OH-'); 'N e-' also sets entry['nel'] = N."""
    for key in reactions:
        text = reactions[key]['reaction']
        if '<->' in text:
            sides = sum([part.split('->') for part in text.split('<->')], [])
        else:
            sides = text.split('->')
        parsed = []
        for side in (s.strip() for s in sides):
            names = []
            for term in side.split(' + '):
                nel = re.findall(r'([0-9]+)[ ]+e-', term)
                if nel:
                    reactions[key]['nel'] = int(nel[0])
                count = re.findall(r'([0-9]+)[ ]+[*A-Za-z]+', term)
                if not count:
                    names.append(term.strip())
                else:
                    n = int(count[0])
                    names.extend([term[len(str(n)) + 1:].strip()] * n)
            parsed.append(names)
        reactions[key]['reaction'] = parsed
    return reactions


class Transport(object):
    def __init__(self, catint_path=None, species=None, electrode_reactions=None, electrolyte_reactions=None, system=None,
                 pb_bound=None, nx=100, descriptors=None, model_name=None, comsol_args=None, catmap_args=None, only_plot=False,
                 resultsdir=None):
        if only_plot:
            return
        self.catint_path = catint_path
        self.model_name = 'catint' if model_name is None else model_name
        self.resultsdir = resultsdir
        self.mpi_rank, self.mpi_size = 0, 1
        # ---- dictionaries (transport.py:143-262) ------------------------------------------------------------------
        if species is None:     # reference defaults, :185-194
            species = collections.OrderedDict([
                ('species1', {'symbol': r'K^+', 'name': 'potassium', 'diffusion': 1.96e-9, 'kind': 'electrolyte',
                              'bulk_concentration': 0.001 * 1000.}),
                ('species2', {'symbol': r'HCO_3^-', 'name': 'bicarbonate', 'diffusion': 1.2e-9, 'kind': 'electrolyte',
                              'bulk_concentration': 0.001 * 1000.})])
        else:
            for sp in species:
                for key in species[sp]:
                    if key not in SPECIES_KEYS:
                        raise TransportError('No such key "' + key + '" in species list.')
        self.species = collections.OrderedDict((k, dict(v)) for k, v in species.items())
        if system is not None:
            for key in system:
                if key not in SYSTEM_KEYS:
                    raise TransportError('No such key "' + key + '" in system list. Current system list = {}'.format(SYSTEM_KEYS))
        self.system = dict(system) if system is not None else {}
        for key, value in SYSTEM_DEFAULTS.items():
            if key not in self.system:
                self.system[key] = list(value) if isinstance(value, list) else value
        self.system['exclude species'] = list(self.system['exclude species'])
        for es in ('e-', 'H2O'):                                   # :236-243
            if es not in self.system['exclude species']:
                self.system['exclude species'] += [es]
        for es in self.system['exclude species']:
            self.species.pop(es, None)

        electrolyte_reactions = self._complete_species(electrolyte_reactions, electrode_reactions)
        self.nspecies = len(self.species)
        for sp in self.species:
            self.species[sp].setdefault('bulk_concentration', 0.0)

        # ---- pH (:264-284) -----------------------------------------------------------------------------------------
        if 'bulk_pH' in self.system:
            if 'H+' in self.species:
                self.species['H+']['bulk_concentration'] = 10 ** (-self.system['bulk_pH']) * 1000.
            elif 'OH-' in self.species:
                self.species['OH-']['bulk_concentration'] = 10 ** (-(14 - self.system['bulk_pH'])) * 1000.
        else:
            if 'H+' in self.species:
                self.system['bulk_pH'] = -np.log10(self.species['H+']['bulk_concentration'] / 1000.)
            elif 'OH-' in self.species:
                self.system['bulk_pH'] = 14 + np.log10(self.species['OH-']['bulk_concentration'] / 1000.)
            else:
                self.system['bulk_pH'] = 7.0
        self.system['surface_pH'] = self.system['bulk_pH']
        self.system['surface_potential'] = self.system['phiM']
        self.system['pH'] = [self.system['bulk_pH']]
        for sp in self.species:
            self.species[sp].setdefault('surface_concentration', self.species[sp]['bulk_concentration'])
        # activity coefficients of the size-modified model from the surface concentrations (:299-304)
        phi_zero = 0.
        for sp in self.species:
            if 'MPB_radius' in self.species[sp]:
                phi_zero += self.species[sp]['MPB_radius'] ** 3 * self.species[sp]['surface_concentration'] * unit_NA
        for sp in self.species:
            self.species[sp]['surface_activity_coefficient'] = 1. / (1. - phi_zero)

        self.eps = self.system['epsilon'] * unit_eps0
        self.beta = 1. / (self.system['temperature'] * unit_R)
        self.use_migration = not ('migration' in self.system and not self.system['migration'])
        self.use_convection = 'flow rate' in self.system

        # ---- reaction tables (:325-395) ------------------------------------------------------------------------------
        if electrolyte_reactions is not None:
            self.electrolyte_reactions = collections.OrderedDict()
            for group in electrolyte_reactions:
                for key, entry in ELECTROLYTE_REACTIONS[group].items():
                    self.electrolyte_reactions[key] = {k: (list(v) if isinstance(v, list) else v) for k, v in entry.items()}
        else:
            self.electrolyte_reactions = None
        self.use_electrolyte_reactions = bool(self.system['electrolyte reactions']) if 'electrolyte reactions' in self.system else True
        if self.electrolyte_reactions is not None and self.use_electrolyte_reactions:
            if any('rates' in r for r in self.electrolyte_reactions.values()):
                self.electrolyte_reactions = parse_reaction_table(self.electrolyte_reactions)
        if self.electrolyte_reactions is None and self.use_electrolyte_reactions:
            raise TransportError('Electrolyte reactions were requested by input, but no electrolyte reaction was defined.')
        if self.use_electrolyte_reactions:
            for el in self.electrolyte_reactions:
                for sp in sum(self.electrolyte_reactions[el]['reaction'], []):
                    if sp not in self.species and sp not in self.system['exclude species']:
                        raise TransportError('Species {} has not been defined, but is used in the electrolyte reactions'.format(sp))
        self.electrode_reactions = None if electrode_reactions is None else \
            collections.OrderedDict((k, dict(v)) for k, v in electrode_reactions.items())
        self.use_electrode_reactions = bool(self.system.get('electrode reactions', False))
        if self.electrode_reactions is not None:
            self.use_electrode_reactions = True
            self.electrode_reactions = parse_reaction_table(self.electrode_reactions)
        elif self.use_electrode_reactions:
            raise TransportError('Electrode reactions were requested by input, but no electrode reaction was defined.')
        if self.use_electrode_reactions:
            for el in self.electrode_reactions:
                for sp in sum(self.electrode_reactions[el]['reaction'], []):
                    if sp not in self.species and sp not in self.system['exclude species'] and not sp.startswith('*'):
                        raise TransportError('Species {} has not been defined, but is used in the electrode reactions'.format(sp))
        # products / educts / spectators (:397-413)
        self.product_list, self.educt_list, self.electrolyte_list = [], [], []
        if self.use_electrode_reactions:
            excl = self.system['exclude species']
            for sp in self.electrode_reactions:
                self.product_list.append(sp)
                lhs, rhs = self.electrode_reactions[sp]['reaction'][0], self.electrode_reactions[sp]['reaction'][1]
                for rr in lhs:
                    if rr != sp and rr not in ['e-'] and rr not in excl and rr not in self.educt_list:
                        self.educt_list.append(rr)
                for rr in rhs:
                    if rr != sp and rr not in ['e-'] and rr not in excl and rr not in self.product_list:
                        self.product_list.append(rr)
        for sp in self.species:
            if sp not in self.product_list and sp not in self.educt_list:
                self.electrolyte_list.append(sp)

        # ---- transport coefficients, Debye length, mesh (:423-460) ---------------------------------------------------
        self.D = np.array([self.species[sp]['diffusion'] if 'diffusion' in self.species[sp] else 0.0 for sp in self.species])
        if all(a in self.system for a in ['water viscosity', 'electrolyte viscosity']):   # Stokes-Einstein rescaling
            self.D = np.array([d * float(self.system['water viscosity']) / float(self.system['electrolyte viscosity']) for d in self.D])
        self.mu = self.D * self.charges * self.beta
        self.ionic_strength = 0.0
        for isp, sp in enumerate(self.species):
            self.ionic_strength += self.charges[isp] ** 2 * self.species[sp]['bulk_concentration']
        self.ionic_strength *= 0.5
        with np.errstate(divide='ignore'):
            self.debye_length = np.sqrt(self.eps / self.beta / 2. / self.ionic_strength)
        self.nx = nx
        if 'boundary thickness' in self.system:
            self.boundary_thickness = self.system['boundary thickness']
            self.xmax = self.boundary_thickness
            self.dx = self.xmax / (self.nx * 1.)
        else:
            nx_mod = max(1., np.ceil(self.nx / 10.))
            self.xmax = self.debye_length * nx_mod
            self.dx = self.debye_length / nx_mod
        self.xmesh = np.arange(0, self.xmax + self.dx, self.dx)
        self.nx = len(self.xmesh)
        self.xmesh_init, self.nx_init, self.xmax_init = self.xmesh, self.nx, self.xmax
        self.mesh_uniform = True
        self.external_charge = np.zeros([len(self.xmesh)])
        self.count = 1

        self._initialize_fluxes()
        self._set_boundary_and_initial_conditions(pb_bound)
        # vzeta is read by the legacy integrators' wall condition (calculator_old.py:529, :1003) but is not a key of
        # today's Transport: unless the caller sets it, it follows phiM (also per lane in a phiM sweep)
        self.vzeta_follows_phiM = 'vzeta' not in self.system
        self.system.setdefault('vzeta', self.system['phiM'])
        self.system['efield'] = np.zeros([self.nx])
        self.system['potential'] = np.zeros([self.nx])
        self.system['charge_density'] = np.zeros([self.nx])
        # the legacy integrators' reaction table (calculator_old.py:159-208: tp.reactions[r]['reactants'], ['rates']) and the
        # physical mode's mass-action table: the electrolyte reactions with rates, excluded species (unit activity) dropped
        self.reactions = collections.OrderedDict()
        if self.use_electrolyte_reactions and self.electrolyte_reactions is not None:
            excl = self.system['exclude species']
            for key, rx in self.electrolyte_reactions.items():
                if 'rates' in rx and isinstance(rx['reaction'], list):
                    self.reactions[key] = {'reactants': [[s for s in side if s not in excl] for side in rx['reaction']],
                                           'rates': list(rx['rates'])}
        self.use_reactions = len(self.reactions) > 0
        self.calc = None
        self.initialize_descriptors(descriptors)
        self.catmap_args = {} if catmap_args is None else catmap_args
        self.comsol_args = {} if comsol_args is None else comsol_args
        if 'RF' not in self.system:                                  # initialize_comsol, :853-858
            par = self.comsol_args.get('parameter', {})
            self.system['RF'] = float(par['RF'][0]) if 'RF' in par else 1.0

    # -- transport.py:537-768 -------------------------------------------------------------------------------------------
    def _complete_species(self, electrolyte_reactions, electrode_reactions):
        """initialize_species: reaction-derived species, tabulated D / name / symbol / Henry constant, charges, bulk
        concentrations from Henry's law, buffer equilibria and electroneutrality.  Returns the list of electrolyte reaction
        groups that take part in the dynamics (incl. 'additional_cell_reactions')."""
        excl = self.system['exclude species']
        self.use_mpb = any('MPB_radius' in self.species[sp] for sp in self.species)
        reacting = []
        if electrode_reactions is not None:
            for e in electrode_reactions:
                for rx in species_of_reaction_string(electrode_reactions[e]['reaction']):
                    reacting.append(rx)
                    if rx not in self.species and rx not in excl:
                        self.species[rx] = {}
        constraints, additional, buffer_species, groups = None, None, [], None
        if electrolyte_reactions is not None:
            groups = []
            for e in electrolyte_reactions:        # dictionaries carry options, strings name a buffer system of the table
                if isinstance(e, dict):
                    if 'constraints' in e:
                        constraints = e['constraints']
                    if 'additional_cell_reactions' in e:
                        additional = e['additional_cell_reactions']
                else:
                    groups.append(e)
            for g in groups:
                if g not in ELECTROLYTE_REACTIONS:
                    raise TransportError('Unknown electrolyte reaction set {}'.format(g))
                for entry in ELECTROLYTE_REACTIONS[g].values():
                    for rx in species_of_reaction_string(entry['reaction']):
                        if rx not in buffer_species:
                            buffer_species.append(rx)
                        if rx not in self.species and rx not in excl:
                            self.species[rx] = {}
        for sp in self.species:
            d = self.species[sp]
            if 'diffusion' not in d:
                if sp not in SPECIES_TABLE:
                    raise TransportError('No diffusion constant for {}. Provide it as an input'.format(sp))
                d['diffusion'] = SPECIES_TABLE[sp][2]
            if 'name' not in d:
                d['name'] = SPECIES_TABLE[sp][0] if sp in SPECIES_TABLE else sp
            if 'symbol' not in d:
                if sp not in SPECIES_TABLE:
                    raise TransportError('No symbol (charge) known for {}'.format(sp))
                d['symbol'] = SPECIES_TABLE[sp][1]
        for sp, h in HENRY_CONSTANTS.items():
            if sp in self.species:
                self.species[sp]['Henry constant'] = float(h) * 1e5           # mol/m^3/bar
        for sp in self.species:
            if 'Henry constant' not in self.species[sp] and sp in set(reacting) and sp not in excl and sp not in ['OH-', 'H+']:
                raise TransportError('No Henry constant found for {}'.format(sp))
        for sp in self.species:
            d = self.species[sp]
            if sp not in excl and 'bulk_concentration' in d and isinstance(d['bulk_concentration'], str) and \
                    d['bulk_concentration'] == 'Henry':
                if 'Henry constant' not in d:
                    raise TransportError('Henry constant was selected for the bulk concentration of {}, but none is tabulated'.format(sp))
                d['bulk_concentration'] = d['Henry constant'] * self.system['pressure']
        for sp in self.species:
            self.species[sp]['charge'] = charge_from_symbol(self.species[sp]['symbol'])
        self.charges = np.array([self.species[sp]['charge'] * unit_F for sp in self.species])

        if groups is not None:
            self._solve_buffer_equilibria(groups, buffer_species, constraints)
            if additional is not None:
                groups = groups + [additional]
        # electroneutrality (:743-765): concentrations are rounded to 8 decimals first so that no residual bulk charge is left
        if any(isinstance(self.species[sp].get('bulk_concentration'), str) and self.species[sp]['bulk_concentration'] == 'charge_neutrality'
               for sp in self.species):
            for sp in self.species:
                if not isinstance(self.species[sp]['bulk_concentration'], str):
                    self.species[sp]['bulk_concentration'] = round(self.species[sp]['bulk_concentration'], 8)
        closed = 0
        for sp in self.species:
            d = self.species[sp]
            if isinstance(d.get('bulk_concentration'), str) and d['bulk_concentration'] == 'charge_neutrality':
                total = 0.
                for sp2 in self.species:
                    d2 = self.species[sp2]
                    if 'bulk_concentration' in d2 and not isinstance(d2['bulk_concentration'], str):
                        total += d2['charge'] * d2['bulk_concentration']
                d['bulk_concentration'] = -total / d['charge']
                closed += 1
        if closed > 1:
            raise TransportError('Only a single species can be evaluated by charge neutrality')
        for sp in self.species:
            self.species[sp].setdefault('bulk_concentration', 0.0)
        self.system['reference_gas_concentration'] = 10 ** 5 / unit_R / unit_T
        return groups

    def _solve_buffer_equilibria(self, groups, buffer_species, constraints):
        """Bulk concentrations the input leaves open, from the buffer equilibria of the selected systems and, optionally, a
        counter-ion constraint (transport.py:641-735): scipy's fsolve from the all-ones start, as the reference does.  The
        reference walks the buffer species as a *set* (hash order); here the order of first appearance is used, so fsolve's
        answer agrees with it to fsolve's own tolerance, and exactly after the electroneutrality pass's 8-decimal rounding."""
        from scipy.optimize import fsolve
        excl = self.system['exclude species']
        unknowns = [sp for sp in buffer_species if sp not in excl and 'bulk_concentration' not in self.species[sp]]
        if not unknowns:
            return
        entries = [entry for g in groups for entry in ELECTROLYTE_REACTIONS[g].values()]
        n_constraints = len(constraints) if constraints is not None else 0
        if len(unknowns) != len(entries) + n_constraints:
            raise TransportError('Number of unknown concentrations {} does not match the number of buffer equilibria equations {}. '
                                 'These are the unknowns = {}'.format(len(unknowns), len(entries) + n_constraints, unknowns))
        if len(unknowns) > 4:
            raise TransportError('More than 4 unknowns in the buffer concentrations are not implemented yet')
        sides = []
        for entry in entries:
            lhs = [_SPECIES_IN_EQUILIBRIUM.findall(b.strip())[0] for b in entry['reaction'].split('->')[0].split(' + ')]
            rhs = [_SPECIES_IN_EQUILIBRIUM.findall(b.strip())[0] for b in entry['reaction'].split('->')[1].split(' + ')]
            sides.append((lhs, rhs, entry['constant']))

        def value(sp, var):
            return self.species[sp]['bulk_concentration'] if 'bulk_concentration' in self.species[sp] else var[sp]

        def equations(p):
            var = dict(zip(unknowns, np.atleast_1d(p)))
            eq = ()
            for lhs, rhs, K in sides:
                prod_rhs = 1
                for s in rhs:
                    if s not in excl:
                        prod_rhs *= value(s, var)
                prod_lhs = 1
                for s in lhs:
                    if s not in excl:
                        prod_lhs *= value(s, var)
                eq += (prod_rhs / prod_lhs - K,)
            if constraints is not None:
                charge_sum = 0.0
                for s in buffer_species:
                    if s not in excl:
                        charge_sum += value(s, var) * self.species[s]['charge']
                for con in constraints:
                    if con == 'counter_ion_concentration':
                        eq += (constraints[con] + charge_sum,)
            return eq

        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            sol = fsolve(equations, (1,) * len(unknowns))
        for sp, v in zip(unknowns, sol):
            self.species[sp]['bulk_concentration'] = v

    # -- transport.py:929-1095 ------------------------------------------------------------------------------------------
    def _initialize_fluxes(self):
        """Wall fluxes of every species from the ONE flux / current density / flux equation given per electrode reaction and
        the reaction's stoichiometry.  Numbers when everything is numeric, expression strings as soon as one is an equation,
        'catmap' for every species when CatMAP owns the kinetics."""
        if not self.use_electrode_reactions:
            for sp in self.species:
                self.species[sp]['flux'] = 0.0
            return
        self.use_catmap = any(self.species[sp].get('flux') == 'catmap' for sp in self.species if isinstance(self.species[sp].get('flux'), str))
        if self.use_catmap:
            for sp in self.species:
                self.species[sp].setdefault('flux', 'catmap')
            return
        excl = self.system['exclude species']
        ers = self.electrode_reactions
        for sp in self.species:
            if sum(1 for key in self.species[sp] if key in _FLUX_KEYS) > 1:
                raise TransportError('Flux of species {} has been defined by more than one method.'.format(sp))

        def real_species(er):
            return [ep for ep in sum(ers[er]['reaction'], []) if ep not in excl and '*' not in ep]
        for er in ers:
            given = sum(1 for ep in real_species(er) if any(a in _FLUX_KEYS for a in self.species[ep]))
            if given > 1:
                raise TransportError('More than one flux has been defined for equation {}.'.format(ers[er]['reaction']))
            if given == 0:
                raise TransportError('No flux defined in equation {}. Define one flux.'.format(ers[er]['reaction']))
        symbolic = any('flux-equation' in self.species[sp] for sp in self.species)
        e_in_educts = None
        for er in ers:                                             # reduction or oxidation (the LAST reaction decides, :993-1002)
            if 'e-' in ers[er]['reaction'][0]:
                e_in_educts = True
            elif 'e-' in ers[er]['reaction'][1]:
                e_in_educts = False
            else:
                raise TransportError('No electron found in the reactions.')
        # current densities -> fluxes  (j / nel / F * nprod, negative for a reduction; :1004-1032)
        for sp in self.species:
            d = self.species[sp]
            if symbolic and 'flux' in d:
                d['flux'] = str(d['flux'])
            elif 'current density' in d:
                for reac in ers:
                    if sp == reac.split('-')[0]:
                        nprod = len([a for a in ers[reac]['reaction'][1] if a == sp])
                        if symbolic:
                            d['flux'] = ('(-1)' if e_in_educts else '1') + '*' + str(d['current density'] / ers[reac]['nel'] / unit_F * nprod)
                        else:
                            d['flux'] = (-1 if e_in_educts else 1) * d['current density'] / ers[reac]['nel'] / unit_F * nprod
            elif symbolic and 'flux-equation' in d:
                d['flux'] = d['flux-equation']
        # species without a flux of their own, in order of appearance (:1034-1043)
        missing = []
        for er in ers:
            for reac in sum(ers[er]['reaction'], []):
                if reac not in ers and reac not in ['e-'] and reac not in excl and reac not in missing and not reac.startswith('*'):
                    missing += [reac]
        reference = {}
        for er in ers:
            for ep in real_species(er):
                if any(a in _FLUX_KEYS for a in self.species[ep]):
                    reference[er] = ep
                    break
        for er in ers:
            educts, products = ers[er]['reaction'][0], ers[er]['reaction'][1]
            sp = reference[er]
            for m in missing:
                if m not in educts and m not in products:
                    continue
                count_m = max(educts.count(m), products.count(m)) * 1.
                count_ref = max(educts.count(sp), products.count(sp)) * 1.
                same_side = (m in educts and sp in educts) or (m in products and sp in products)
                if symbolic:
                    self.species[m].setdefault('flux', '0')
                    self.species[m]['flux'] += '+' + ('1' if same_side else '(-1)') + '*' + self.species[sp]['flux'] + '*' + str(count_m / count_ref)
                else:
                    self.species[m].setdefault('flux', 0.0)
                    self.species[m]['flux'] += (1 if same_side else (-1.)) * self.species[sp]['flux'] * count_m / count_ref
        for sp in self.species:
            self.species[sp].setdefault('flux', '0.0' if symbolic else 0.0)

    # -- transport.py:1277-1321, :1396-1487 ---------------------------------------------------------------------------------
    def _set_boundary_and_initial_conditions(self, pb_bound):
        self.c0 = np.repeat([self.species[sp]['bulk_concentration'] for sp in self.species], self.nx).astype(float)
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
        self.phiM_init = None
        self.boundary_type = 'flux'
        # The reference creates flux_bound only when every flux is a number (:1471-1475); with symbolic fluxes ('catmap',
        # equations) the numbers come from the kinetics each SCF iteration.  Here the array always exists (zeros for the
        # symbolic entries) because the batched Calculator fills it from its flux callback / implicit wall kinetics.
        self.flux_symbolic = any(isinstance(self.species[sp]['flux'], str) for sp in self.species)
        self.flux_bound = np.zeros([self.nspecies, 2])
        if not self.flux_symbolic:
            self.flux_bound[:, 0] = [self.species[sp]['flux'] for sp in self.species]
        self.dc_dt_bound = np.zeros([self.nspecies, 2]) * 10 ** 3           # dc_dt_boundary {'all': {'r': 0.0}} in mol/l/s -> mol/m^3/s
        self.efield_bound = np.array([0.0 * 1e10, None])                    # efield_boundary {'l': 0.0} V/Ang -> V/m

    # -- transport.py:1135-1195 ------------------------------------------------------------------
    def initialize_descriptors(self, descriptors):
        if descriptors is None:
            descriptors = collections.OrderedDict([('phiM', [self.system['phiM']]), ('temperature', [self.system['temperature']])])
        else:
            if any(not isinstance(descriptors[d], (list, np.ndarray)) for d in descriptors):
                raise TransportError('Descriptors must be given as list.')
            descriptors = collections.OrderedDict(descriptors)
        if len(descriptors) == 1:   # dummy second descriptor, :1157-1163
            if 'temperature' not in descriptors:
                descriptors['temperature'] = [self.system['temperature']]
            else:
                descriptors['phiM'] = [self.system['phiM']]
        if len(descriptors) != 2:
            raise TransportError('Cannot use other than 2 descriptors')
        self.descriptors = descriptors
        keys = list(descriptors.keys())
        # (the reference leaves alldata unset when no descriptors are given, :1150-1152; the single point is listed here)
        self.alldata_names = [[v1, v2] for v1, v2 in itertools.product(descriptors[keys[0]], descriptors[keys[1]])]
        self.alldata = [{'species': {sp: {} for sp in self.species}, 'system': {}} for _ in self.alldata_names]

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

    def gouy_chapman(self, x, phiM=None):
        """Potential of the z:z diffuse layer and its numerical slope, as transport.py:1373-1383 evaluates them."""
        if phiM is None:
            phiM = self.system['phiM']
        gamma = np.tanh(phiM * self.beta * unit_F / 4.)
        scale = 2. / (self.beta * abs(self.charges[0]))

        def potential(pos):
            decay = np.exp(-1. / self.debye_length * pos)
            return scale * np.log((1. + gamma * decay) / (1. - gamma * decay))
        slope = (potential(x + 1e-10) - potential(x - 1e-10)) / (2 * 1e-10)
        return potential(x), slope

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
