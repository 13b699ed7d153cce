'''``import mod16`` resolved to the MI355X build: every name of ``mod16_amd`` under the reference
package's own import name, so ``from mod16 import MOD16, psychrometric_constant, radiation_net,
svp_slope, latent_heat_vaporization`` (reference tests/tests.py:8) and ``mod16.utils`` /
``mod16.models`` work without an edited import. Put the repository root in front of the
reference on ``sys.path``; nothing else lives here.'''
from mod16_amd import *          # noqa: F401,F403
from mod16_amd import (          # noqa: F401  (names a star import leaves out or tests name explicitly)
    MOD16, PFT_VALID, STEFAN_BOLTZMANN, SPECIFIC_HEAT_CAPACITY_AIR, MOL_WEIGHT_WET_DRY_RATIO_AIR,
    TEMP_LAPSE_RATE, GRAV_ACCEL, GAS_LAW_CONST, AIR_MOL_WEIGHT, STD_TEMP_K, STD_PRESSURE_PASCALS,
    AIR_PRESSURE_RATE, latent_heat_vaporization, psychrometric_constant, radiation_net, svp,
    svp_slope, evapotranspiration_raster, evapotranspiration_raw, pinned_empty, __version__)
from . import utils, models      # noqa: F401,E402
