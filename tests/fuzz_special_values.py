#!/usr/bin/env python3
"""Special values in the drivers (zeros, NaN, infinities, fill values, huge and tiny numbers): the
EXACT and FAST kernels against the numpy oracle, one special value in one driver per pixel, and
then pairs. Prints, per arithmetic, the pixels whose NaN / zero / inf masks differ from the
oracle's, tallied by (driver, value). Uses oracle/ as the checker, so it lives under tests/ (run it as a script:
`python tests/fuzz_special_values.py`); the two assertions that came out of it are
tests/test_gpu_parity.py::test_special_values_*."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import mod16_amd as m16  # noqa: E402
from mod16_amd.utils import restore_bplut, bplut_table  # noqa: E402
from mod16_amd.models import COLLECTION61_BPLUT  # noqa: E402
from oracle import mod16_oracle as oracle  # noqa: E402

NAMES = ['lw_net_day', 'lw_net_night', 'sw_rad_day', 'sw_rad_night', 'sw_albedo', 'temp_day',
         'temp_night', 'temp_annual', 'tmin', 'vpd_day', 'vpd_night', 'pressure', 'fpar', 'lai']
SPECIAL = [0.0, -0.0, np.nan, -9999.0, 65535.0, 1.0, -1.0, 1e-7, 273.15, 35.85, 34.15, 3.4e38,
           -3.4e38, 1e300, -1e300, 1e-300, np.inf, -np.inf]


def main():
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    rng = np.random.default_rng(123)
    per = 600                                   # pixels per (driver, value)
    n = per * 14 * len(SPECIAL)
    t_d = rng.uniform(255, 305, n)
    t_n = t_d - rng.uniform(0, 12, n)
    es = lambda t: 610.8 * np.exp(17.27 * (t - 273.15) / (t - 273.15 + 237.3))
    drv = [rng.uniform(-100, 0, n), rng.uniform(-50, 0, n), rng.uniform(0, 360, n), np.zeros(n),
           rng.uniform(0.1, 0.22, n), t_d, t_n, rng.uniform(265, 300, n), t_n - rng.uniform(0, 3, n),
           es(t_d) * (1 - rng.uniform(0.05, 1, n)), es(t_n) * (1 - rng.uniform(0.05, 1, n)),
           rng.uniform(7e4, 101340, n), rng.uniform(0.02, 0.89, n), rng.uniform(0.13, 5.34, n)]
    which = np.repeat(np.arange(14 * len(SPECIAL)), per)
    for j in range(14):
        for s, v in enumerate(SPECIAL):
            drv[j][which == j * len(SPECIAL) + s] = v
    cls = rng.choice(np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12], np.uint8), n)
    with np.errstate(all='ignore'):
        want = oracle.evapotranspiration_raster(bplut, cls, *drv)
    for mode, flag in (('exact', m16._lib.MATH_EXACT), ('fast', m16._lib.MATH_FAST)):
        got = m16.evapotranspiration_raster(table, cls, *drv, math=flag)
        bad = np.zeros(n, bool)
        off = np.zeros(n, bool)
        worst = 0.0
        for g, w in zip(got, want):
            bad |= (np.isnan(g) != np.isnan(w)) | ((g == 0) != (w == 0)) | (np.isinf(g) != np.isinf(w))
            ok = np.isfinite(w) & (w != 0) & np.isfinite(g)
            rel = np.zeros(n)
            rel[ok] = np.abs(g[ok] - w[ok]) / np.abs(w[ok])
            off |= rel > 1e-9
            worst = max(worst, float(rel[~bad].max()))
        off &= ~bad
        print('%s: %d of %d pixels with a mask that differs from the oracle; %d more off by > 1e-9 (worst %.2e)'
              % (mode, int(bad.sum()), n, int(off.sum()), worst))
        for what, sel in (('masks', bad), ('> 1e-9', off)):
            tally = np.bincount(which[sel], minlength=14 * len(SPECIAL)).reshape(14, len(SPECIAL))
            for j in range(14):
                hits = ['%g: %d' % (SPECIAL[s], tally[j, s]) for s in range(len(SPECIAL)) if tally[j, s]]
                if hits:
                    print('   %-7s %-13s %s' % (what, NAMES[j], ', '.join(hits)))


if __name__ == '__main__':
    main()
