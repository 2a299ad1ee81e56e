"""Constant tables the reference's ``Transport`` reads at construction time (numbers only; transcribed from the reference's data
files, which never travel with this package):

  SPECIES_TABLE            data/diffusion_constants.txt   name, symbol (carries the charge), D [m^2/s] at 25 C
  HENRY_CONSTANTS          data/henry_constants.txt       mol/m^3/Pa (Transport multiplies by 1e5 -> mol/m^3/bar, transport.py:617)
  ELECTROLYTE_REACTIONS    catint/data.py:7-122           buffer systems: reaction string, equilibrium constant, [k_f, k_r]

Units follow the reference: concentrations mol/m^3, equilibrium constants in powers of mol/m^3, rates in the matching units.
"""
import collections

# formula: (name, symbol, D)
SPECIES_TABLE = collections.OrderedDict([
    ('H2', ('hydrogen', 'H_2', 5.11e-9)), ('CO2', ('carbon_dioxide', 'CO_2', 1.91e-9)), ('CO', ('carbon_monoxide', 'CO', 2.23e-9)),
    ('O2', ('oxygen', 'O_2', 2.42e-9)),
    ('H3PO4', ('phosphoric_acid', 'H_3PO_4', 8.8e-10)), ('H2PO4-', ('h2_phosphate', 'H_2PO_4^-', 9.59e-10)),
    ('HPO42-', ('h_phosphate', 'HPO_4^{2-}', 7.59e-10)), ('PO43-', ('phophate', 'PO_4^{3-}', 8.24e-10)),
    ('H3Cit', ('h3_citrate', 'H_3Cit', 8.87e-10)), ('H2Cit-', ('h2_citrate', 'H_2Cit^-', 7.99e-10)),
    ('HCit2-', ('h_citrate', 'HCit^{2-}', 7.e-10)), ('Cit3-', ('citrate', 'Cit^{3-}', 6.23e-10)),
    ('HCO3-', ('bicarbonate', 'HCO_3^-', 1.185e-9)), ('CO32-', ('carboxylate', 'CO_3^{2-}', 0.923e-9)),
    ('Cs+', ('cesium', 'Cs^+', 2.056e-9)), ('D+', ('deuterium', 'D^+', 6.655e-9)), ('H+', ('hydronium', 'H^+', 9.311e-9)),
    ('K+', ('potassium', 'K^+', 1.957e-9)), ('Na+', ('sodium', 'Na^+', 1.334e-9)), ('NH4+', ('ammonium', 'NH_4^+', 1.957e-9)),
    ('Li+', ('lithium', 'Li^+', 1.029e-9)), ('Ca2+', ('calcium', 'Ca^{2+}', 0.792e-9)),
    ('OH-', ('hydroxide', 'OH^-', 5.273e-9)), ('Cl-', ('chloride', 'Cl^-', 2.032e-9)), ('I-', ('iodide', 'I^-', 2.045e-9)),
    ('Br-', ('bromide', 'Br^-', 2.080e-9)), ('ClO4-', ('perchlorate', 'ClO_4^-', 1.792e-9)),
    ('CH4', ('methane', 'CH_4', 1.49e-9)), ('C2H4', ('ethylene', 'C_2H_4', 1.51e-9)),
    ('CH3CO2H', ('acetic_acid', 'CH_3CO_2H', 1.29e-9)), ('CH3CH2OH', ('ethanol', 'CH_3CH_2OH', 1.24e-9)),
])

HENRY_CONSTANTS = collections.OrderedDict([
    ('CH4', 1.4e-5), ('C2H6', 1.9e-5), ('CH3OH', 2.0), ('CH3CH2OH', 1.9), ('CO', 9.7e-6), ('CO2', 3.3e-4), ('N2', 6.4e-6),
    ('H2', 7.8e-6), ('NH3', 5.9e-1), ('O2', 1.2e-5), ('CH2O', 3.2e1), ('NO', 1.9e-5),
])


def _rx(reaction, constant, rates=None):
    d = collections.OrderedDict([('reaction', reaction), ('constant', constant)])
    if rates is not None:
        d['rates'] = list(rates)
    return d


ELECTROLYTE_REACTIONS = collections.OrderedDict([
    ('bicarbonate-base', collections.OrderedDict([
        ('buffer-base', _rx('CO2 + OH- <-> HCO3-', 44400.0, [5.93, 0.00013355855855855855])),
        ('buffer-base2', _rx('HCO3- + OH- <-> CO32- + H2O', 4.66, [1.0e5, 21459.227467811157])),
    ])),
    ('bicarbonate-acid', collections.OrderedDict([
        ('buffer-acid', _rx('CO2 + H2O <-> HCO3- + H+', 0.000444, [3.7e-2, 83.33333333333333])),
        ('buffer-acid2', _rx('HCO3- <-> CO32- + H+', 4.66e-8, [59.44, 1275536480.6866953])),
    ])),
    ('phosphate-acid', collections.OrderedDict([
        ('phosphate-1', _rx('H3PO4 + H+ <-> H2PO4-', 0.00629 * 1000., [5.6e8, 8.9e10 / 1000.])),
        ('phosphate-2', _rx('H2PO4- + H+ <-> HPO42-', 6.32e-8 * 1000., [6.32e2, 1e10 / 1000.])),
        ('phosphate-3', _rx('HPO42- + H+ <-> PO43-', 4.47e-13 * 1000., [4.47e-3, 1e10 / 1000.])),
    ])),
    ('citrate-acid', collections.OrderedDict([
        ('citrate-1', _rx('H3Cit + H+ <-> H2Cit-', 0.000745 * 1000, [7.45e6, 1e10 / 1000.])),
        ('citrate-2', _rx('H2Cit- + H+ <-> HCit2-', 1.73e-5 * 1000, [1.73e5, 1e10 / 1000.])),
        ('citrate-3', _rx('HCit2- + H+ <-> Cit3-', 4.02e-7 * 1000, [4.02e3, 1e10 / 1000.])),
    ])),
    ('borate-base', collections.OrderedDict([
        ('borate-1', _rx('H3BO3 + OH- <-> H2BO3- + H2O', 5.75e-10 * 1000)),
        ('borate-2', _rx('H2BO3- + OH- <-> HBO32- + H2O', 3.98e-13 * 1000)),
        ('borate-3', _rx('HBO32- + OH- <-> BO33- + H2O', 5.01e-14 * 1000)),
    ])),
    ('water-diss', collections.OrderedDict([
        ('self-dissociation of water', _rx('H2O <-> OH- + H+', 1e-8, [2.4e-5 * 1000., 2.4e-5 / 1e-14 / 1000.])),
    ])),
])
