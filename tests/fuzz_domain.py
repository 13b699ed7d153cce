#!/usr/bin/env python3
"""Maps the domain of the strength-reduced (FAST) and mixed-precision arithmetic: one probe
value in one driver per pixel, a ladder of magnitudes per driver, against the numpy oracle.
Prints, per arithmetic and driver, the probe values at which a NaN / zero / inf mask differs
from the oracle's or a value is off by more than 1e-8 (float64) / 1e-4 (mixed). With the
domain guard in place (mod16_physics.hpp: fast_out_of_domain) every list must be empty; with
MOD16_NO_GUARD=1 in the environment of the BUILD (-DMOD16_NO_GUARD) the lists are the map the
guard was drawn from. Uses oracle/ as the checker, so it lives under tests/
(`python tests/fuzz_domain.py`)."""
import collections
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
LADDER = [0.0, -0.0, np.nan, np.inf, -np.inf, 1e-300, -1e-300, 1e-7, -1e-7, 1.0, -1.0, 34.15, 35.85,
          50.0, 100.0, 120.0, 150.0, 180.0, 200.0, 273.15, 350.0, 400.0, 500.0, 700.0, 1000.0, 1300.0,
          1332.0, 1400.0, 2000.0, -100.0, -273.15, -9999.0, 9999.0, 65535.0, 1e6, -1e6, 1e8, 1e10,
          -1e10, 1e15, -1e15, 1e20, 1e25, 1e30, -1e30, 1e35, 3.4e38, -3.4e38, 1e45, 1e60, 1e100,
          -1e100, 1e150, 1e200, 1e300, -1e300]


def rasters(values, per, seed=123):
    rng = np.random.default_rng(seed)
    n = per * 14 * len(values)
    t_d = rng.uniform(255, 305, n)
    t_n = t_d - rng.uniform(0, 12, n)
    es = lambda t: 610.8 * np.exp(17.27 * (t - 273.15) / (t - 273.15 + 237.3))
    drv = [rng.uniform(-100, 0, n), rng.uniform(-50, 0, n), rng.uniform(0, 360, n), np.zeros(n),
           rng.uniform(0.1, 0.22, n), t_d, t_n, rng.uniform(265, 300, n), t_n - rng.uniform(0, 3, n),
           es(t_d) * (1 - rng.uniform(0.05, 1, n)), es(t_n) * (1 - rng.uniform(0.05, 1, n)),
           rng.uniform(7e4, 101340, n), rng.uniform(0.02, 0.89, n), rng.uniform(0.13, 5.34, n)]
    which = np.repeat(np.arange(14 * len(values)), per)
    for j in range(14):
        for s, v in enumerate(values):
            drv[j][which == j * len(values) + s] = v
    cls = rng.choice(np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12], np.uint8), n)
    return cls, drv, which


def report(name, got, want, which, values, tol, mixed=False):
    n = which.size
    bad = np.zeros(n, bool)
    off = np.zeros(n, bool)
    for g, w in zip(got, want):
        g = g.astype(np.float64)
        if mixed:   # parity.assert_mixed_parity: float32 subnormals count as zero, absolute bound
            tiny = float(np.finfo(np.float32).tiny)
            g = np.where(np.abs(g) < tiny, 0, g)
            w = np.where(np.abs(w) < tiny, 0, w)
        bad |= (np.isnan(g) != np.isnan(w)) | ((g == 0) != (w == 0)) | (np.isinf(g) != np.isinf(w))
        ok = np.isfinite(w) & (w != 0) & np.isfinite(g)
        rel = np.zeros(n)
        rel[ok] = np.abs(g[ok] - w[ok]) / np.abs(w[ok])
        if mixed and ok.any():
            rel[ok] = np.where(np.abs(g[ok] - w[ok]) <= 1e-6 * np.abs(w[ok]).max(), 0, rel[ok])
        off |= rel > tol
    off &= ~bad
    print('%s: %d of %d pixels with a mask that differs from the oracle, %d more off by > %g'
          % (name, int(bad.sum()), n, int(off.sum()), tol))
    for what, sel in (('masks', bad), ('values', off)):
        tally = np.bincount(which[sel], minlength=14 * len(values)).reshape(14, len(values))
        for j in range(14):
            hits = ['%g' % values[s] for s in range(len(values)) if tally[j, s]]
            if hits:
                print('   %-6s %-13s %s' % (what, NAMES[j], ' '.join(hits)))
    return int(bad.sum()), int(off.sum())


def main():
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    cls, drv, which = rasters(LADDER, per=128)
    total = 0
    with np.errstate(all='ignore'):
        want = oracle.evapotranspiration_raster(bplut, cls, *drv)
    got = m16.evapotranspiration_raster(table, cls, *drv, math=m16._lib.MATH_FAST)
    total += sum(report('fast float64', got, want, which, LADDER, 1e-8))
    with np.errstate(all='ignore'):
        want6 = oracle.evapotranspiration_raster(bplut, cls, *drv, separate=True)
    got6 = m16.evapotranspiration_raster(table, cls, *drv, separate=True, math=m16._lib.MATH_FAST)
    for k, part in enumerate(('canopy', 'soil', 'transpiration')):
        total += sum(report('fast float64, ' + part, [got6[0][k], got6[1][k]], [want6[0][k], want6[1][k]],
                            which, LADDER, 1e-8))
    # float32 rasters: the checker is the float64 oracle on the widened inputs, rounded once
    lad32 = [v for v in LADDER if not np.isfinite(v) or v == 0 or 1e-37 < abs(v) < 3.41e38]
    cls, drv, which = rasters(lad32, per=128)
    d32 = [d.astype(np.float32) for d in drv]
    with np.errstate(all='ignore'):
        want = [w.astype(np.float32).astype(np.float64) for w in
                oracle.evapotranspiration_raster(bplut, cls, *[d.astype(np.float64) for d in d32])]
    got = m16.evapotranspiration_raster(table, cls, *d32, math=m16._lib.MATH_FAST)
    total += sum(report('fast float32', got, want, which, lad32, 1e-6))
    got = m16.evapotranspiration_raster(table, cls, *d32, math=m16._lib.MATH_MIXED)
    total += sum(report('mixed float32', got, want, which, lad32, 1e-3, mixed=True))
    total += raw_forms(table, bplut)
    total += pairs(table, bplut)
    total += raw_pairs(table, bplut)
    total += storm(table, bplut)
    total += calibration_path(table)
    return 0 if total == 0 else 1


def calibration_path(table, ndraw=6):
    """The FAST arithmetic of the calibration path (MOD16._et_batch(math=FAST) and the bound problem,
    round 4) over the same ladder: one probe value in one driver per pixel, a few parameter vectors
    (one with csl = 0: the whole-array switch off), against the oracle's restatement of
    MOD16._evapotranspiration (mod16/__init__.py:195-382) and against the reference-order kernels."""
    rng = np.random.default_rng(11)
    _, drv, which = rasters(LADDER, per=128)
    lo = np.array([-10, 5, 400, 2000, 0.01, 0.01, 1e-6, 0.001, 20, 60, 50.0])
    hi = np.array([-6, 15, 1000, 5000, 0.12, 0.12, 1e-4, 0.01, 70, 120, 800.0])
    params = rng.uniform(lo, hi, (ndraw, 11))
    params[1, 7] = 0.0
    fast = m16.MOD16._et_batch(params, *drv, math=m16._lib.MATH_FAST)
    prob = m16.MOD16._et_bind(*drv, max_draws=ndraw)
    bound = prob.rows(params)
    bad = 0
    if not np.array_equal(fast, bound, equal_nan=True):
        print('calibration: bound rows differ from the unbound call')
        bad += 1
    with np.errstate(all='ignore'):
        for d in range(ndraw):
            want = oracle.et_static(list(params[d]), *drv)
            bad += sum(report('calibration fast float64, draw %d' % d, [fast[d]], [want], which, LADDER, 1e-8))
    print('calibration: %d of %d pixels outside the fast domain (reference order)' % (prob.n_outside_domain, fast.shape[1]))
    return bad


PAIR_VALUES = [0.0, -0.0, np.nan, np.inf, -np.inf, -9999.0, 65535.0, 1e15, 3.4e38, -3.4e38, 1e300, -1e300,
               1e-300, -1e-300, 1e-7, 1.0, -1.0, 35.85, 34.15, 1400.0, 1e40, -1e40, 1e49, 1e60, 1e100, 1e150, 1e200]


def pairs(table, bplut, n=1200000, seed=5):
    """TWO special values in two different drivers of every pixel (random pairs: an infinity next
    to a NaN, to a zero, to the albedo 1 that turns inf * (1 - albedo) into NaN ...): the guard
    has to hold for combinations, not only for the single values it was drawn from."""
    rng = np.random.default_rng(seed)
    cls, drv, _ = rasters([1.0], per=1)             # shapes only
    t_d = rng.uniform(255, 305, n)
    t_n = t_d - rng.uniform(0, 12, n)
    es = lambda t: 610.8 * np.exp(17.27 * (t - 273.15) / (t - 273.15 + 237.3))
    drv = [rng.uniform(-100, 0, n), rng.uniform(-50, 0, n), rng.uniform(0, 360, n), np.zeros(n),
           rng.uniform(0.1, 0.22, n), t_d, t_n, rng.uniform(265, 300, n), t_n - rng.uniform(0, 3, n),
           es(t_d) * (1 - rng.uniform(0.05, 1, n)), es(t_n) * (1 - rng.uniform(0.05, 1, n)),
           rng.uniform(7e4, 101340, n), rng.uniform(0.02, 0.89, n), rng.uniform(0.13, 5.34, n)]
    a = rng.integers(0, 14, n)
    b = (a + rng.integers(1, 14, n)) % 14
    va = np.array(PAIR_VALUES)[rng.integers(0, len(PAIR_VALUES), n)]
    vb = np.array(PAIR_VALUES)[rng.integers(0, len(PAIR_VALUES), n)]
    for k in range(14):
        drv[k][a == k] = va[a == k]
        drv[k][b == k] = vb[b == k]
    cls = rng.choice(np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12], np.uint8), n)
    with np.errstate(all='ignore'):
        want = oracle.evapotranspiration_raster(bplut, cls, *drv, separate=True)
    got = m16.evapotranspiration_raster(table, cls, *drv, separate=True, math=m16._lib.MATH_FAST)
    bad = np.zeros(n, bool)
    off = np.zeros(n, bool)
    for g3, w3 in zip(got, want):
        for g, w in zip(g3, w3):
            bad |= (np.isnan(g) != np.isnan(w)) | ((g == 0) != (w == 0)) | (np.isinf(g) != np.isinf(w))
            ok = np.isfinite(w) & (w != 0) & np.isfinite(g)
            rel = np.zeros(n)
            rel[ok] = np.abs(g[ok] - w[ok]) / np.abs(w[ok])
            off |= rel > 1e-8
    off &= ~bad
    print('pairs, fast float64 (six components): %d of %d pixels with a mask that differs from the oracle, '
          '%d more off by > 1e-08' % (int(bad.sum()), n, int(off.sum())))
    tally = collections.Counter()
    for i in np.nonzero(bad | off)[0][:100000]:
        tally[(NAMES[a[i]], '%g' % va[i], NAMES[b[i]], '%g' % vb[i])] += 1
    for key, cnt in tally.most_common(40):
        print('   %-13s = %-9s with %-13s = %-9s : %d' % (key + (cnt,)))
    total = int(bad.sum()) + int(off.sum())
    # float32 rasters, FAST and MIXED (totals): the values a float32 holds
    keep = np.array([not np.isfinite(v) or v == 0 or 1e-37 < abs(v) < 3.41e38 for v in PAIR_VALUES])
    ok_pix = np.isin(va, np.array(PAIR_VALUES)[keep]) | np.isnan(va)
    ok_pix &= np.isin(vb, np.array(PAIR_VALUES)[keep]) | np.isnan(vb)
    d32 = [d[ok_pix].astype(np.float32) for d in drv]
    c32 = cls[ok_pix]
    with np.errstate(all='ignore'):
        want = [w.astype(np.float32).astype(np.float64) for w in
                oracle.evapotranspiration_raster(bplut, c32, *[d.astype(np.float64) for d in d32])]
    which = np.zeros(c32.size, np.int64)
    for name, math, tol, mixed in (('pairs, fast float32', m16._lib.MATH_FAST, 1e-6, False),
                                   ('pairs, mixed float32', m16._lib.MATH_MIXED, 1e-3, True)):
        got = m16.evapotranspiration_raster(table, c32, *d32, math=math)
        total += sum(report(name, got, want, which, [0.0], tol, mixed=mixed))
        tiny = float(np.finfo(np.float32).tiny)
        badp = np.zeros(c32.size, bool)
        for g, w in zip(got, want):
            g = g.astype(np.float64)
            if mixed:
                g, w = np.where(np.abs(g) < tiny, 0, g), np.where(np.abs(w) < tiny, 0, w)
            badp |= (np.isnan(g) != np.isnan(w)) | ((g == 0) != (w == 0)) | (np.isinf(g) != np.isinf(w))
        tally = collections.Counter()
        ia, ib, xa, xb = a[ok_pix], b[ok_pix], va[ok_pix], vb[ok_pix]
        for i in np.nonzero(badp)[0][:2000]:
            tally[(NAMES[ia[i]], '%g' % xa[i], NAMES[ib[i]], '%g' % xb[i])] += 1
        for key, cnt in tally.most_common(25):
            print('   %-13s = %-9s with %-13s = %-9s : %d' % (key + (cnt,)))
    return total


RAW_NAMES = NAMES[:9] + ['qv10m_day', 'qv10m_night', 'ps_day', 'ps_night', 'elevation']
RAW_LADDER = LADDER + [-1.7, -1.65, -0.5, 0.5, 0.99, 1.5, 8848.0, 2e4, 4e4, 44330.0, 44331.0, 5e4, -5e4]


def raw_forms(table, bplut):
    """The raw-driver forms (N1): special values in the raw fields."""
    global NAMES
    total = 0
    for name, dtype, math, tol in (('raw float64', np.float64, m16._lib.MATH_FAST, 1e-8),
                                   ('raw fast float32', np.float32, m16._lib.MATH_FAST, 1e-6),
                                   ('raw mixed float32', np.float32, m16._lib.MATH_MIXED, 1e-3)):
        values = RAW_LADDER if dtype == np.float64 else \
            [v for v in RAW_LADDER if not np.isfinite(v) or v == 0 or 1e-37 < abs(v) < 3.41e38]
        rng = np.random.default_rng(7)
        per = 96
        n = per * 14 * len(values)
        t_d = rng.uniform(255, 305, n)
        t_n = t_d - rng.uniform(0, 12, n)
        raw = [rng.uniform(-100, 0, n), rng.uniform(-50, 0, n), rng.uniform(0, 360, n), np.zeros(n),
               rng.uniform(0.1, 0.22, n), t_d, t_n, rng.uniform(265, 300, n), t_n - rng.uniform(0, 3, n),
               rng.uniform(5e-4, 2e-2, n), rng.uniform(5e-4, 2e-2, n),
               rng.uniform(70000, 101340, n), rng.uniform(70000, 101340, n), rng.uniform(-50, 4500, n)]
        which = np.repeat(np.arange(14 * len(values)), per)
        for j in range(14):
            for s, v in enumerate(values):
                raw[j][which == j * len(values) + s] = v
        raw = [a.astype(dtype) for a in raw]
        fpar = rng.integers(0, 101, n).astype(np.uint8)
        lai = rng.integers(0, 70, n).astype(np.uint8)
        cls = rng.choice(np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12], np.uint8), n)
        with np.errstate(all='ignore'):
            want = oracle.evapotranspiration_raw(bplut, cls, [a.astype(np.float64) for a in raw], fpar, lai)
            want = [w.astype(dtype).astype(np.float64) for w in want]
        got = m16.evapotranspiration_raw(table, cls, *raw, fpar, lai, math=math)
        keep, NAMES = NAMES, RAW_NAMES
        total += sum(report(name, got, want, which, values, tol, mixed=math == m16._lib.MATH_MIXED))
        NAMES = keep
    return total



def raw_pairs(table, bplut, n=600000, seed=6):
    """Pairs of special values in the RAW fields (specific humidity, surface pressure, elevation
    included), float64 and float32 (FAST, MIXED)."""
    rng = np.random.default_rng(seed)
    values = np.array(PAIR_VALUES + [0.5, 0.99, 1.5, -1.65, 8848.0, 4e4, 44330.0, -5e4])
    t_d = rng.uniform(255, 305, n)
    t_n = t_d - rng.uniform(0, 12, n)
    raw = [rng.uniform(-100, 0, n), rng.uniform(-50, 0, n), rng.uniform(0, 360, n), np.zeros(n),
           rng.uniform(0.1, 0.22, n), t_d, t_n, rng.uniform(265, 300, n), t_n - rng.uniform(0, 3, n),
           rng.uniform(5e-4, 2e-2, n), rng.uniform(5e-4, 2e-2, n),
           rng.uniform(70000, 101340, n), rng.uniform(70000, 101340, n), rng.uniform(-50, 4500, n)]
    a = rng.integers(0, 14, n)
    b = (a + rng.integers(1, 14, n)) % 14
    va, vb = values[rng.integers(0, len(values), n)], values[rng.integers(0, len(values), n)]
    for k in range(14):
        raw[k][a == k] = va[a == k]
        raw[k][b == k] = vb[b == k]
    fpar = rng.integers(0, 101, n).astype(np.uint8)
    lai = rng.integers(0, 70, n).astype(np.uint8)
    cls = rng.choice(np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12], np.uint8), n)
    total = 0
    f32ok = np.array([not np.isfinite(v) or v == 0 or 1e-37 < abs(v) < 3.41e38 for v in values])
    pix32 = (np.isin(va, values[f32ok]) | np.isnan(va)) & (np.isin(vb, values[f32ok]) | np.isnan(vb))
    for name, dtype, math, tol, mixed in (('raw pairs, fast float64', np.float64, m16._lib.MATH_FAST, 1e-8, False),
                                          ('raw pairs, fast float32', np.float32, m16._lib.MATH_FAST, 1e-6, False),
                                          ('raw pairs, mixed float32', np.float32, m16._lib.MATH_MIXED, 1e-3, True)):
        sel = np.ones(n, bool) if dtype == np.float64 else pix32
        r = [x[sel].astype(dtype) for x in raw]
        with np.errstate(all='ignore'):
            want = [w.astype(dtype).astype(np.float64) for w in oracle.evapotranspiration_raw(
                bplut, cls[sel], [x.astype(np.float64) for x in r], fpar[sel], lai[sel])]
        got = m16.evapotranspiration_raw(table, cls[sel], *r, fpar[sel], lai[sel], math=math)
        global NAMES
        keep, NAMES = NAMES, RAW_NAMES
        total += sum(report(name, got, want, np.zeros(int(sel.sum()), np.int64), [0.0], tol, mixed=mixed))
        tiny = float(np.finfo(np.float32).tiny)
        badp = np.zeros(int(sel.sum()), bool)
        for g, w in zip(got, want):
            g = g.astype(np.float64)
            if mixed:
                g, w = np.where(np.abs(g) < tiny, 0, g), np.where(np.abs(w) < tiny, 0, w)
            badp |= (np.isnan(g) != np.isnan(w)) | ((g == 0) != (w == 0)) | (np.isinf(g) != np.isinf(w))
            ok = np.isfinite(w) & (w != 0) & np.isfinite(g)
            rel = np.zeros(badp.size)
            rel[ok] = np.abs(g[ok] - w[ok]) / np.abs(w[ok])
            if mixed and ok.any():
                rel[ok] = np.where(np.abs(g[ok] - w[ok]) <= 1e-6 * np.abs(w[ok]).max(), 0, rel[ok])
            badp |= rel > tol
        tally = collections.Counter()
        ia, ib, xa, xb = a[sel], b[sel], va[sel], vb[sel]
        for i in np.nonzero(badp)[0][:3000]:
            tally[(RAW_NAMES[ia[i]], '%g' % xa[i], RAW_NAMES[ib[i]], '%g' % xb[i])] += 1
        for key, cnt in tally.most_common(25):
            print('   %-13s = %-9s with %-13s = %-9s : %d' % (key + (cnt,)))
        NAMES = keep
    return total


def storm(table, bplut, n=2000000, seed=9, p=0.12):
    """Every driver of every pixel independently a special value with probability p (so three,
    four, five at a time are common), float64 totals + components, float32 FAST and MIXED."""
    rng = np.random.default_rng(seed)
    t_d = rng.uniform(255, 305, n)
    t_n = t_d - rng.uniform(0, 12, n)
    es = lambda t: 610.8 * np.exp(17.27 * (t - 273.15) / (t - 273.15 + 237.3))
    drv = [rng.uniform(-100, 0, n), rng.uniform(-50, 0, n), rng.uniform(0, 360, n), np.zeros(n),
           rng.uniform(0.1, 0.22, n), t_d, t_n, rng.uniform(265, 300, n), t_n - rng.uniform(0, 3, n),
           es(t_d) * (1 - rng.uniform(0.05, 1, n)), es(t_n) * (1 - rng.uniform(0.05, 1, n)),
           rng.uniform(7e4, 101340, n), rng.uniform(0.02, 0.89, n), rng.uniform(0.13, 5.34, n)]
    values = np.array(PAIR_VALUES)
    f32ok = np.array([not np.isfinite(v) or v == 0 or 1e-37 < abs(v) < 3.41e38 for v in values])
    all32 = np.ones(n, bool)
    for k in range(14):
        hit = rng.random(n) < p
        v = values[rng.integers(0, len(values), n)]
        drv[k][hit] = v[hit]
        all32 &= ~hit | np.isin(v, values[f32ok]) | np.isnan(v)
    cls = rng.choice(np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12], np.uint8), n)
    total = 0
    with np.errstate(all='ignore'):
        want6 = oracle.evapotranspiration_raster(bplut, cls, *drv, separate=True)
        want = oracle.evapotranspiration_raster(bplut, cls, *drv)
    got6 = m16.evapotranspiration_raster(table, cls, *drv, separate=True)
    got = m16.evapotranspiration_raster(table, cls, *drv)
    z = np.zeros(n, np.int64)
    # (values to 1e-6 here: a pixel with several absurd drivers is computed in the reference's
    # operation order with ocml's pow / exp where the oracle has glibc's -- on garbage like 1e-80
    # their last-bit differences are amplified to 1e-8; the masks are what this run is about)
    total += sum(report('storm, fast float64 totals', got, want, z, [0.0], 1e-6))
    for k, part in enumerate(('canopy', 'soil', 'transpiration')):
        total += sum(report('storm, fast float64 ' + part, [got6[0][k], got6[1][k]], [want6[0][k], want6[1][k]], z, [0.0], 1e-6))
    for g, w, what in ((got6[0][2], want6[0][2], 'day transpiration'), (got6[1][2], want6[1][2], 'night transpiration'),
                       (got[0], want[0], 'day'), (got[1], want[1], 'night')):
        ok = np.isfinite(w) & (w != 0) & np.isfinite(g)
        rel = np.zeros(n)
        rel[ok] = np.abs(g[ok] - w[ok]) / np.abs(w[ok])
        for i in np.nonzero(rel > 1e-6)[0][:5]:
            print('   %s: got %.17g want %.17g rel %.2e cls %d drivers %s'
                  % (what, g[i], w[i], rel[i], cls[i], ' '.join('%s=%.6g' % (NAMES[k][:6], drv[k][i]) for k in range(14))))
    d32 = [d[all32].astype(np.float32) for d in drv]
    c32 = cls[all32]
    with np.errstate(all='ignore'):
        w32 = [w.astype(np.float32).astype(np.float64) for w in
               oracle.evapotranspiration_raster(bplut, c32, *[d.astype(np.float64) for d in d32])]
    z = np.zeros(c32.size, np.int64)
    total += sum(report('storm, fast float32', m16.evapotranspiration_raster(table, c32, *d32), w32, z, [0.0], 1e-6))
    total += sum(report('storm, mixed float32', m16.evapotranspiration_raster(table, c32, *d32, math=m16._lib.MATH_MIXED),
                        w32, z, [0.0], 1e-3, mixed=True))
    return total


if __name__ == '__main__':
    sys.exit(main())
