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

# MCD12Q1 LC_Type2 names -> numeric PFT code (reference mod16/models.py:13-25)
PFT_ALL = {
    'Evergreen Needleleaf Forest (ENF)': 1,
    'Evergreen Broadleaf Forest (EBF)': 2,
    'Deciduous Needleleaf Forest (DNF)': 3,
    'Deciduous Broadleaf Forest (DBF)': 4,
    'Mixed Forest (MF) ': 5,
    'Closed Shrublands (CSH)': 6,
    'Open Shrublands (OSH)': 7,
    'Woody Savannas (WSV)': 8,
    'Savannas (SAV)': 9,
    'Grasslands (GRS)': 10,
    'Croplands (CRO)': 12
}


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
