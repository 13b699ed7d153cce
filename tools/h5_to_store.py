#!/usr/bin/env python3
"""Cal-Val HDF5 container -> the RasterStore layout of mod16_amd.io (for hosts that have h5py;
this image does not: tests/test_h5_to_store.py runs `convert()` on a dict-backed stand-in for an
open h5py.File -- tests/fake_h5py.py -- and the store it writes through `io.run_store`).

The reference's calibration driver reads one HDF5 file (mod16/calibration.py:50-112 documents its
layout, :304-423 `_load_data` reads it). `mod16_amd.io.run_store` streams the same datasets from
one `.npy` file per dataset under directories named like the HDF5 groups. Field map
(HDF5 dataset -> store file, shape in the store):

    MERRA2/LWGNT_daytime, LWGNT_nighttime        -> same name .npy      (T, N') float32
    MERRA2/SWGDN_daytime                          -> same                (T, N')   (night: zero, :383)
    MERRA2/T10M_daytime, T10M_nighttime, Tmin     -> same                (T, N')
    MERRA2/QV10M_daytime, QV10M_nighttime         -> same                (T, N')   (VPD is computed in the kernel, :395-401)
    MERRA2/PS_daytime, PS_nighttime               -> same                (T, N')
    MERRA2/T10M (24-h mean), mean over T (:390)   -> MERRA2/T10M_annual  (N',)
    MODIS/MCD43GF_black_sky_sw_albedo (T, N, P)   -> same                (T, N')
    MODIS/MOD15A2HGF_fPAR, MOD15A2HGF_LAI (T,N,P) -> same                (T, N') uint8 MODIS codes: 0 .. 248 as
                                                                          stored; NaN, negative or >= 249 -> 255 (fill -> NaN in the kernel)
    state/PFT (N, P)                              -> state/PFT           (N',) uint8
    state/elevation_m (N,)                        -> state/elevation_m   (N',)

The tower sub-grid axis P (the MODIS pixels around a tower) becomes PIXELS: N' = N x P with
--subgrid flatten (default; every MODIS pixel is run, the tower-level fields repeat over P), or
N' = N with --subgrid center (the middle pixel of each sub-grid). The reference instead averages
fPAR / LAI / albedo over P before the forward run (:412-419, np.nanmean, float results) and runs
the forward model ONCE per tower on the means; the store keeps the MODIS integer codes, so that
mode is not offered here: `flatten` runs every sub-pixel and the mean over P can be taken of the
OUTPUTS afterwards -- ET of the mean inputs and the mean of the ETs differ wherever the forward run
is not linear in fPAR / LAI / albedo. Dataset names follow the reference's defaults; --name
KEY=PATH overrides one (the starred names of calibration.py:50-112 are configurable there too).

    python tools/h5_to_store.py CALVAL.h5 STORE_DIR [--t0 K] [--subgrid flatten|center] [--name albedo=MODIS/...]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mod16_amd import io as store_io  # noqa: E402

NAMES = {'albedo': 'MODIS/MCD43GF_black_sky_sw_albedo', 'fPAR': 'MODIS/MOD15A2HGF_fPAR',
         'LAI': 'MODIS/MOD15A2HGF_LAI', 'PFT': 'state/PFT', 'elevation': 'state/elevation_m',
         'MAT': 'MERRA2/T10M'}


def modis_code(a):
    """fPAR [%] / LAI [x 10] as the reference reads them (floats, NaN = missing, calibration.py:410-419)
    or as MOD15A2H stores them (uint8, codes >= 249 = fill) -> the uint8 codes of the store: values
    0 .. 248 rounded to the nearest code, everything else (NaN, inf, negative, >= 248.5) the fill
    code 255, which the raw-driver kernels decode to NaN."""
    a = np.asarray(a)
    if a.dtype == np.uint8:
        return a
    with np.errstate(invalid='ignore'):
        f = a.astype(np.float64)
        ok = np.isfinite(f) & (f > -0.5) & (f < 248.5)
        return np.where(ok, np.rint(np.where(ok, f, 0.0)), 255).astype(np.uint8)


def convert(hdf, store_dir, t0=0, subgrid='flatten', names=None):
    """`hdf`: an open h5py.File (or anything that maps dataset paths to objects with `.shape` and
    numpy-style `[...]`); writes the store under `store_dir` and returns it."""
    names = dict(NAMES, **(names or {}))
    if subgrid not in ('flatten', 'center'):
        raise ValueError("subgrid must be 'flatten' or 'center'")
    pft = np.asarray(hdf[names['PFT']][:])
    P = pft.shape[1] if pft.ndim == 2 else 1
    N = pft.shape[0]
    T = hdf['MERRA2/Tmin'].shape[0] - t0
    if T <= 0:
        raise ValueError('t0 = %d leaves no time step' % t0)
    mid = P // 2

    def pixels(a):
        """(..., N) tower-level or (..., N, P) sub-grid data -> (..., N')"""
        sub = a.ndim >= 2 and a.shape[-1] == P and a.shape[-2] == N and P > 1
        if subgrid == 'center':
            return a[..., mid] if sub else a
        return a.reshape(a.shape[:-2] + (N * P,)) if sub else np.repeat(a, P, axis=-1)

    n_pix = N * P if subgrid == 'flatten' else N
    store = store_io.RasterStore.create(store_dir, T, n_pix, np.float32)
    for _, name in store_io.DYNAMIC_FIELDS:
        src = names['albedo'] if name == NAMES['albedo'] else name
        out = store.array(name, 'r+')
        for t in range(T):          # step by step: the container may not fit host memory
            out[t] = pixels(np.asarray(hdf[src][t0 + t], np.float32))
        out.flush()
    for name, key in ((store_io.FPAR, 'fPAR'), (store_io.LAI, 'LAI')):
        out = store.array(name, 'r+')
        for t in range(T):
            out[t] = modis_code(pixels(np.asarray(hdf[names[key]][t0 + t])))
        out.flush()
    mat = np.zeros(N, np.float64)
    for t in range(T):
        mat += np.asarray(hdf[names['MAT']][t0 + t], np.float64)
    out = store.array('MERRA2/T10M_annual', 'r+')
    out[:] = pixels((mat / T).astype(np.float32))
    out.flush()
    out = store.array('state/elevation_m', 'r+')
    out[:] = pixels(np.asarray(hdf[names['elevation']][:], np.float32))
    out.flush()
    out = store.array(store_io.PFT, 'r+')
    out[:] = pixels(pft).astype(np.uint8)
    out.flush()
    store.subgrid = (subgrid, P)
    return store


def main():
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('h5')
    ap.add_argument('store')
    ap.add_argument('--t0', type=int, default=0, help='first time step to convert (the reference skips a spin-up: t0)')
    ap.add_argument('--subgrid', default='flatten', choices=['flatten', 'center'])
    ap.add_argument('--name', action='append', default=[], help='KEY=HDF5 path, e.g. albedo=MODIS/MCD43GF_white_sky_sw_albedo')
    args = ap.parse_args()
    try:
        import h5py
    except ImportError:
        sys.exit('h5py is not installed on this host: the converter needs it (the store side needs only numpy)')
    names = {}
    for item in args.name:
        key, _, path = item.partition('=')
        if key not in NAMES or not path:
            sys.exit('--name takes one of %s as KEY=PATH' % ', '.join(sorted(NAMES)))
        names[key] = path
    with h5py.File(args.h5, 'r') as hdf:
        store = convert(hdf, args.store, args.t0, args.subgrid, names)
    print('wrote %s: %d steps x %d pixels (%s sub-grid of %d)'
          % (args.store, store.n_steps, store.n_pixels, store.subgrid[0], store.subgrid[1]))


if __name__ == '__main__':
    main()
