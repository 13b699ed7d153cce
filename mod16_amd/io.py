'''
Rasters on disk (SURVEY.md section 8f, N4 -- the part of it this image can
serve: HDF5 / GeoTIFF libraries are not installed, numpy's ``.npy`` is).

Drivers, class raster and outputs are ``.npy`` files opened as memory maps; the
forward run streams them through the library's HOST mode (2 Mi-pixel tiles,
several staging threads and device slabs), so neither inputs nor outputs are
ever resident in host memory as a whole: the page cache and the disks set the
pace (the GPU's share is below 1 % of the wall time).
'''
import numpy as np

from . import _lib, evapotranspiration_raster

DRIVER_NAMES = (
    'lw_net_day', 'lw_net_night', 'sw_rad_day', 'sw_rad_night', 'sw_albedo',
    'temp_day', 'temp_night', 'temp_annual', 'tmin', 'vpd_day', 'vpd_night',
    'pressure', 'fpar', 'lai')


def evapotranspiration_npy(bplut, cls_path, driver_paths, out_day_path, out_night_path,
                           beta=None, math=_lib.MATH_FAST, device=0):
    '''
    ``evapotranspiration_raster`` on ``.npy`` files.

    Parameters
    ----------
    bplut : dict or numpy.ndarray
        As for ``evapotranspiration_raster``
    cls_path : str
        ``.npy`` file of the land-cover class raster (uint8)
    driver_paths : dict or sequence
        The 14 driver files, by name (``DRIVER_NAMES``) or in that order; all
        of the class raster's shape, all float64 or all float32
    out_day_path, out_night_path : str
        ``.npy`` files to create (same shape and dtype as the drivers)

    Returns
    -------
    tuple
        The two output memory maps (flushed)
    '''
    if isinstance(driver_paths, dict):
        missing = [k for k in DRIVER_NAMES if k not in driver_paths]
        if missing:
            raise KeyError('missing driver files: %s' % ', '.join(missing))
        driver_paths = [driver_paths[k] for k in DRIVER_NAMES]
    if len(driver_paths) != len(DRIVER_NAMES):
        raise ValueError('expected %d driver files' % len(DRIVER_NAMES))
    cls = np.load(cls_path, mmap_mode='r')
    drivers = [np.load(p, mmap_mode='r') for p in driver_paths]
    dtype = drivers[0].dtype
    if dtype not in (np.float64, np.float32):
        raise TypeError('drivers must be float64 or float32')
    for name, d in zip(DRIVER_NAMES, drivers):
        if d.shape != cls.shape or d.dtype != dtype or not d.flags.c_contiguous:
            raise ValueError('%s: expected a C-ordered %s array of shape %s' % (name, dtype, cls.shape))
    outs = [np.lib.format.open_memmap(p, mode='w+', dtype=dtype, shape=cls.shape)
            for p in (out_day_path, out_night_path)]
    evapotranspiration_raster(bplut, cls, *drivers, beta=beta, math=math, device=device, out=outs)
    for o in outs:
        o.flush()
    return tuple(outs)
