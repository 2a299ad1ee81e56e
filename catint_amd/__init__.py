"""catint_amd -- MI355X-native batched 1D Poisson-Nernst-Planck transport path, a drop-in for the
legacy finite-difference solve under sringe/CatINT's Calculator (see DESIGN.md / INTEGRATION.md)."""
from ._capi import (PnpSolver, PnpError, PnpLibraryError, load_library, LIB_PATH,  # noqa: F401
                    PB_DD, PB_VWALL_GBULK, PB_GWALL_VBULK, PB_VWALL_GWALL, PB_VBULK_GBULK,
                    STATUS_OK, STATUS_NAN, STATUS_NEGATIVE)

__version__ = '0.1.0'
