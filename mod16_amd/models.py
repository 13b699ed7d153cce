'''
Ready-made MOD16 model variants, interface-compatible with the reference's
``mod16.models`` (reference mod16/models.py).
'''
import os

import numpy as np

from . import MOD16, PFT_VALID
from .utils import DATA_DIR, restore_bplut

MOD16_DIR = os.path.dirname(os.path.abspath(__file__))
COLLECTION61_BPLUT = os.path.join(
    DATA_DIR, 'MOD16_BPLUT_C5.1_05deg_MCD43B_Albedo_MERRA_GMAO.csv')

# MCD12Q1 LC_Type2 classes that are Plant Functional Types: (code, name, acronym).
# PFT_ALL maps the reference's display names to the codes (same keys as
# reference mod16/models.py:13-25, including the trailing blank of the MF key).
_LC_TYPE2_PFT = (
    (1, 'Evergreen Needleleaf Forest', 'ENF'), (2, 'Evergreen Broadleaf Forest', 'EBF'),
    (3, 'Deciduous Needleleaf Forest', 'DNF'), (4, 'Deciduous Broadleaf Forest', 'DBF'),
    (5, 'Mixed Forest', 'MF'), (6, 'Closed Shrublands', 'CSH'), (7, 'Open Shrublands', 'OSH'),
    (8, 'Woody Savannas', 'WSV'), (9, 'Savannas', 'SAV'), (10, 'Grasslands', 'GRS'),
    (12, 'Croplands', 'CRO'))
PFT_ALL = dict(('%s (%s)%s' % (name, abbr, ' ' if abbr == 'MF' else ''), code)
               for code, name, abbr in _LC_TYPE2_PFT)


class MOD16Collection61(MOD16):
    '''
    The MOD16 Collection 6.1 model for one Plant Functional Type: parameters
    from the Collection 6.1 BPLUT, ``beta = 250`` where the table has none
    (reference mod16/models.py:27-51).

    Parameters
    ----------
    pft : int
        The numeric code of the Plant Functional Type of interest.
    '''
    def __init__(self, pft, device=0):
        assert pft in PFT_VALID, \
            'Not a recognized numeric PFT code; should be one of: %s' \
            % ','.join(map(str, PFT_VALID))
        table = restore_bplut(COLLECTION61_BPLUT)
        params = dict((key, table[key][pft]) for key in table.keys())
        if np.isnan(params['beta']):
            params['beta'] = 250
        super().__init__(params=params, device=device)
