"""Physical constants -- values identical to the reference's catint/units.py:4-25 (they enter
beta, eps, charges and must match to the last digit for parity)."""
unit_R = 8.3144598
unit_e = 1.6021766208e-19
unit_eps0 = 8.854187817e-12
unit_NA = 6.022140857e23
unit_F = 96485.33289
unit_kB = 1.38064852e-23
unit_T = 298.14
