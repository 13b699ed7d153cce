'''
BPLUT (Biome Properties Look-Up Table) utilities, interface-compatible with the
reference's ``mod16.utils`` (reference mod16/utils.py).
'''
import csv
import io
import os
from collections import Counter

import numpy as np

# CSV row label -> parameter name (reference mod16/utils.py:15-27)
BPLUT_FIELD_LOOKUP = {
    'Tmin_min(C)':      'tmin_close',
    'Tmin_max(C)':      'tmin_open',
    'VPD_min(Pa)':      'vpd_open',
    'VPD_max(Pa)':      'vpd_close',
    'gl_sh((m/s)':      'gl_sh',
    'gl_e_wv(m/s)':     'gl_wv',
    'g_cuticular(m/s)': 'g_cuticular',
    'Cl(m/s)':          'csl',
    'RBL_MIN(s/m)':     'rbl_min',
    'RBL_MAX(s/m)':     'rbl_max',
    'beta':             'beta'
}

# Column order of the CSV -> MCD12Q1 LC_Type2 code (mod16/utils.py:101)
_PFT_LOOKUP = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12]
_PFT_VALID = (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12)

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')
BPLUT_HEADER = [
    'UMD_VEG_LC', 'ENF=0', 'EBF=1', 'DNF=2', 'DBF=3', 'MF=4', 'CShrub=5',
    'OShrub=6', 'Wsavannas=7', 'Savannas=8', 'Grass=9', 'Crop=10']


def restore_bplut(path_or_buffer, nrows=11):
    '''
    Reads a BPLUT CSV into a dict of 11 arrays of length 13, indexed by the
    numeric PFT code; entries of codes that are not PFTs (0, 11) and of fields
    absent from the file (``beta`` in Collection 5.x tables) stay NaN.
    Same contract as the reference's ``restore_bplut`` (mod16/utils.py:81-117).

    Parameters
    ----------
    path_or_buffer : str or file-like
    nrows : int
        Number of data rows to read

    Returns
    -------
    dict
    '''
    if hasattr(path_or_buffer, 'read'):
        text = path_or_buffer.read()
        if isinstance(text, bytes):
            text = text.decode('utf-8')
        handle = io.StringIO(text)
    else:
        handle = open(path_or_buffer, 'r', newline='')
    with handle:
        rows = [r for r in csv.reader(handle) if r]
    output = dict(
        (name, np.full((13,), np.nan)) for name in BPLUT_FIELD_LOOKUP.values())
    for row in rows[1:1 + nrows]:      # rows[0] is the header
        label = row[0]
        if label not in BPLUT_FIELD_LOOKUP:
            raise KeyError(label)
        output[BPLUT_FIELD_LOOKUP[label]][_PFT_LOOKUP] = \
            [float(v) for v in row[1:1 + len(_PFT_LOOKUP)]]
    return output


def write_bplut(params_dict, output_path):
    '''
    Writes a BPLUT parameters dictionary to a CSV file in the layout
    ``restore_bplut`` reads (reference mod16/utils.py:120-145).
    '''
    with open(output_path, 'w', newline='') as file:
        writer = csv.writer(file)
        writer.writerow(BPLUT_HEADER)
        for label, key in BPLUT_FIELD_LOOKUP.items():
            writer.writerow(
                (label, *[params_dict[key][pft] for pft in _PFT_VALID]))


def pft_dominant(pft_map, site_list=None, valid_pft=_PFT_VALID):
    '''
    Dominant (modal) valid PFT among the sub-grid cells of each site, with
    the Cal/Val protocol's fixed assignments (reference mod16/utils.py:29-78):
    CA-SF2, CA-SF3, US-NGC are PFT 3 and US-A10 is excluded (0).

    Parameters
    ----------
    pft_map : numpy.ndarray
        (N x M) PFT codes, N sites by M sub-grid cells
    site_list : list
        (Optional) site names, needed for the fixed assignments
    valid_pft : sequence
        Codes that count as PFTs

    Returns
    -------
    numpy.ndarray
        (N,) float32 array of dominant PFT codes (0 = none valid)
    '''
    pft_map = np.asarray(pft_map)
    dominant = np.zeros(pft_map.shape[0], np.float32)
    for i, cells in enumerate(pft_map):
        counts = Counter(c for c in cells.tolist() if c in valid_pft)
        if counts:
            dominant[i] = counts.most_common()[0][0]
    if site_list is not None:
        site_list = list(site_list)
        if 'US-A10' in site_list:
            dominant[site_list.index('US-A10')] = 0
        for sid in ('CA-SF2', 'CA-SF3', 'US-NGC'):
            if sid in site_list:
                dominant[site_list.index(sid)] = 3
    return dominant


def bplut_table(bplut, beta=None):
    '''
    dict of 11 arrays(13) (what ``restore_bplut`` returns) -> float64 [13][11]
    table in ``MOD16.required_parameters`` column order, the layout the HIP
    library takes (``mod16_set_bplut_f64``). ``beta``, if given, fills the
    ``beta`` column of valid classes where the file left it NaN (the
    reference's callers patch ``beta = 250`` the same way, models.py:49-50).
    '''
    from . import MOD16
    table = np.stack(
        [np.asarray(bplut[k], np.float64) for k in MOD16.required_parameters],
        axis=1)
    if table.shape != (13, 11):
        raise ValueError('BPLUT arrays must have 13 entries each')
    if beta is not None:
        col = MOD16.required_parameters.index('beta')
        fill = np.isnan(table[:, col]) & ~np.isnan(table[:, 0])
        table[fill, col] = beta
    return np.ascontiguousarray(table)
