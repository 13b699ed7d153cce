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
N' = N with --subgrid center (the middle pixel of each sub-grid).

--subgrid mean is the reference's own tower protocol (:412-419): fPAR, LAI and albedo averaged over
the sub-grid (np.nanmean, float results) BEFORE ONE forward run per tower-day, on the dominant PFT
of the sub-grid (utils.pft_dominant, :336-340). Means are no MODIS codes, so this mode writes the
PROCESSED drivers of MOD16.evapotranspiration instead -- STORE_DIR/processed/<driver>.npy, (T, N)
float32, with the reference's pre-processing applied (:380-423: MOD16.vpd and the night-time clamp,
MOD16.air_pressure, fPAR / 100, LAI / 10, the annual mean temperature; VPD and air pressure through
the library's own methods, i.e. on the GPU) and class.npy -- which `run_processed` /
mod16_amd.io.evapotranspiration_npy stream through the forward run.

Which dataset feeds which field is keyed by the reference's OWN look-up keys (`lookup[...]` in
_load_data, the `data: datasets:` mapping of its configuration file): SOURCES below; --name KEY=PATH
(KEY=DAY,NIGHT for the day / night pairs) overrides a path, as the reference's configuration does.
tests/golden/calval_layout.json holds the layout and the keys as the reference states them
(tests/golden/make_calval_layout.py reads them out of its text); tests/test_h5_to_store.py checks
this map against it.

    python tools/h5_to_store.py CALVAL.h5 STORE_DIR [--t0 K] [--subgrid flatten|center|mean] [--name albedo=MODIS/...]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mod16_amd import io as store_io  # noqa: E402

#: the reference's look-up key -> dataset path(s) of the documented layout (calibration.py:50-112); a
#: pair is [daytime, nighttime] as in its configuration. `class_map` is the configuration key of the
#: land-cover dataset (:336); VPD is not part of the documented layout (it is computed from QV10M and
#: PS, :395-401) but may be named, as in the shipped configuration (--subgrid mean reads it then).
SOURCES = {
    'LWGNT': ['MERRA2/LWGNT_daytime', 'MERRA2/LWGNT_nighttime'],
    'SWGDN': ['MERRA2/SWGDN_daytime', None],                     # no night-time short-wave (:383)
    'T10M': ['MERRA2/T10M_daytime', 'MERRA2/T10M_nighttime'],
    'Tmin': 'MERRA2/Tmin',
    'QV10M': ['MERRA2/QV10M_daytime', 'MERRA2/QV10M_nighttime'],
    'PS': ['MERRA2/PS_daytime', 'MERRA2/PS_nighttime'],
    'MAT': 'MERRA2/T10M',
    'albedo': 'MODIS/MCD43GF_black_sky_sw_albedo',
    'fPAR': 'MODIS/MOD15A2HGF_fPAR',
    'LAI': 'MODIS/MOD15A2HGF_LAI',
    'elevation': 'state/elevation_m',
    'class_map': 'state/PFT',
}
#: store file of mod16_amd.io (named like the documented dataset) <- (look-up key, index of the pair)
STORE_FROM = {
    'MERRA2/LWGNT_daytime': ('LWGNT', 0), 'MERRA2/LWGNT_nighttime': ('LWGNT', 1),
    'MERRA2/SWGDN_daytime': ('SWGDN', 0), 'MODIS/MCD43GF_black_sky_sw_albedo': ('albedo', None),
    'MERRA2/T10M_daytime': ('T10M', 0), 'MERRA2/T10M_nighttime': ('T10M', 1), 'MERRA2/Tmin': ('Tmin', None),
    'MERRA2/QV10M_daytime': ('QV10M', 0), 'MERRA2/QV10M_nighttime': ('QV10M', 1),
    'MERRA2/PS_daytime': ('PS', 0), 'MERRA2/PS_nighttime': ('PS', 1),
}
NAMES = SOURCES            # (the name round 5 gave the overridable part of this table)


def source(names, key, index=None):
    """dataset path behind a look-up key (``index``: 0 = daytime, 1 = night-time of a pair)"""
    v = names[key]
    return v[index] if index is not None else v


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
    names = dict(SOURCES, **(names or {}))
    if subgrid == 'mean':
        return convert_processed(hdf, store_dir, t0, names)
    if subgrid not in ('flatten', 'center'):
        raise ValueError("subgrid must be 'flatten', 'center' or 'mean'")
    pft = np.asarray(hdf[names['class_map']][:])
    P = pft.shape[1] if pft.ndim == 2 else 1
    N = pft.shape[0]
    T = hdf[names['Tmin']].shape[0] - t0
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
        src = source(names, *STORE_FROM[name])
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


PROCESSED = 'processed'


def convert_processed(hdf, store_dir, t0=0, names=None):
    """--subgrid mean: the reference's tower protocol (calibration.py:336-340, :380-423) -> the
    processed drivers of MOD16.evapotranspiration, one (T, N) float32 .npy each under
    store_dir/processed/ (mod16_amd.io.DRIVER_NAMES) + class.npy. Returns the directory."""
    import mod16_amd
    from mod16_amd.utils import pft_dominant
    names = dict(SOURCES, **(names or {}))
    T = hdf[names['Tmin']].shape[0] - t0
    if T <= 0:
        raise ValueError('t0 = %d leaves no time step' % t0)
    f32 = lambda key, index=None: np.asarray(hdf[source(names, key, index)][t0:], np.float32)
    pft = np.asarray(hdf[names['class_map']][:])
    sites = None
    if 'FLUXNET/site_id' in hdf:
        sites = [x.decode('utf-8') if hasattr(x, 'decode') else x for x in np.asarray(hdf['FLUXNET/site_id'][:]).tolist()]
    dominant = pft_dominant(pft if pft.ndim == 2 else pft[:, None], site_list=sites).astype(np.uint8)      # :336-340
    N = dominant.shape[0]
    t_day, t_night = f32('T10M', 0), f32('T10M', 1)
    if 'VPD' in names:                                   # (:391-393: precomputed VPD, if the configuration names it)
        vpd_day, vpd_night = f32('VPD', 0), f32('VPD', 1)
    else:                                                # :395-400, MOD16.vpd on the GPU
        vpd_day = mod16_amd.MOD16.vpd(f32('QV10M', 0), f32('PS', 0), t_day)
        vpd_night = mod16_amd.MOD16.vpd(f32('QV10M', 1), f32('PS', 1), t_night)
    vpd_night = np.where(vpd_night < 0, 0, vpd_night).astype(np.float32)          # :401
    elevation = np.asarray(hdf[names['elevation']][:], np.float32)
    if elevation.ndim == 2:
        elevation = elevation.mean(axis=-1)              # :408
    pressure = np.asarray(mod16_amd.MOD16.air_pressure(elevation), np.float32)     # :408
    sub = {}
    with np.errstate(invalid='ignore'):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', RuntimeWarning)      # (a sub-grid of NaN has a NaN mean, as in the reference)
            for key in ('albedo', 'fPAR', 'LAI'):
                a = np.asarray(hdf[names[key]][t0:], np.float64)
                sub[key] = (np.nanmean(a, axis=-1) if a.ndim == 3 else a)      # :412-419
    fields = {
        'lw_net_day': f32('LWGNT', 0), 'lw_net_night': f32('LWGNT', 1),
        'sw_rad_day': f32('SWGDN', 0), 'sw_rad_night': np.zeros((T, N), np.float32),      # :383
        'sw_albedo': sub['albedo'].astype(np.float32),
        'temp_day': t_day, 'temp_night': t_night,
        'temp_annual': np.broadcast_to(f32('MAT').mean(axis=0, dtype=np.float64).astype(np.float32), (T, N)),      # :388-390
        'tmin': f32('Tmin'), 'vpd_day': np.asarray(vpd_day, np.float32), 'vpd_night': vpd_night,
        'pressure': np.broadcast_to(pressure, (T, N)),
        'fpar': (sub['fPAR'] / 100).astype(np.float32), 'lai': (sub['LAI'] / 10).astype(np.float32),      # :422-423
    }
    out_dir = os.path.join(store_dir, PROCESSED)
    os.makedirs(out_dir, exist_ok=True)
    for name in store_io.DRIVER_NAMES:
        a = np.ascontiguousarray(fields[name], np.float32)
        if a.shape != (T, N):
            raise ValueError('%s: shape %s, expected %s' % (name, a.shape, (T, N)))
        np.save(os.path.join(out_dir, name + '.npy'), a)
    np.save(os.path.join(out_dir, 'class.npy'), np.ascontiguousarray(np.broadcast_to(dominant, (T, N))))
    return out_dir


def run_processed(bplut, out_dir, beta=None, **kwargs):
    """The forward run over a directory `convert_processed` wrote: ET_daytime.npy / ET_nighttime.npy
    beside the drivers (mod16_amd.io.evapotranspiration_npy)."""
    paths = {k: os.path.join(out_dir, k + '.npy') for k in store_io.DRIVER_NAMES}
    return store_io.evapotranspiration_npy(bplut, os.path.join(out_dir, 'class.npy'), paths,
                                           os.path.join(out_dir, 'ET_daytime.npy'),
                                           os.path.join(out_dir, 'ET_nighttime.npy'), beta=beta, **kwargs)


def main():
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('h5')
    ap.add_argument('store')
    ap.add_argument('--t0', type=int, default=0, help='first time step to convert (the reference skips a spin-up: t0)')
    ap.add_argument('--subgrid', default='flatten', choices=['flatten', 'center', 'mean'])
    ap.add_argument('--name', action='append', default=[],
                    help='KEY=HDF5 path (KEY=DAY,NIGHT for a pair), e.g. albedo=MODIS/MCD43GF_white_sky_sw_albedo; '
                         'keys: %s, VPD' % ', '.join(sorted(SOURCES)))
    args = ap.parse_args()
    try:
        import h5py
    except ImportError:
        sys.exit('h5py is not installed on this host: the converter needs it (the store side needs only numpy)')
    names = {}
    for item in args.name:
        key, _, path = item.partition('=')
        if (key not in SOURCES and key != 'VPD') or not path:
            sys.exit('--name takes one of %s, VPD as KEY=PATH' % ', '.join(sorted(SOURCES)))
        pair = isinstance(SOURCES.get(key, [None, None]), list)
        names[key] = [x or None for x in path.split(',')] if pair else path
        if pair and len(names[key]) != 2:
            sys.exit('--name %s takes DAY,NIGHT' % key)
    with h5py.File(args.h5, 'r') as hdf:
        store = convert(hdf, args.store, args.t0, args.subgrid, names)
    if args.subgrid == 'mean':
        print('wrote the processed drivers of the tower protocol to %s' % store)
    else:
        print('wrote %s: %d steps x %d pixels (%s sub-grid of %d)'
              % (args.store, store.n_steps, store.n_pixels, store.subgrid[0], store.subgrid[1]))


if __name__ == '__main__':
    main()
