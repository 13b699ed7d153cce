#!/usr/bin/env python3
"""PCIe-inclusive rate of the numpy-in / numpy-out path (MOD16_HOST mode):
evapotranspiration_raster on host arrays, wall time per call."""
import json
import os
import sys
import time


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import mod16_amd  # noqa: E402
from mod16_amd.models import COLLECTION61_BPLUT  # noqa: E402
from mod16_amd.utils import bplut_table, restore_bplut  # noqa: E402
from _drivers import drivers as synth_drivers  # noqa: E402


def main():
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    for shape in ((400, 400), (600, 600), (800, 800), (1200, 1200), (1440, 1440), (2400, 2400), (4800, 4800), (9600, 9600)):
        cls, drv = synth_drivers(shape, seed=16)
        mod16_amd.evapotranspiration_raster(table, cls, *drv)      # warm-up (allocations)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            mod16_amd.evapotranspiration_raster(table, cls, *drv)
            ts.append(time.perf_counter() - t0)
        n = shape[0] * shape[1]
        best = min(ts)
        print(json.dumps({'shape': shape, 'seconds': best, 'pixels_per_s': n / best,
                          'GBps_129B': 129 * n / best / 1e9}))


def raw():
    """The numpy call on RAW drivers in the light input form (float32 fields + uint8 fPAR / LAI: 58 bytes
    per pixel up, 8 down), HOST mode of mod16_et_raw_*: staged by MOD16_HOST_THREADS threads since round 5."""
    import numpy as np
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    rng = np.random.default_rng(3)
    for shape in ((4800, 4800), (9600, 9600)):
        n = shape[0] * shape[1]
        cls, drv = synth_drivers(shape, seed=16)
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        raw = [f32(d) for d in drv[:9]] + [f32(rng.uniform(0.001, 0.02, shape)), f32(rng.uniform(0.001, 0.02, shape)),
                                           f32(rng.uniform(7e4, 1.0134e5, shape)), f32(rng.uniform(7e4, 1.0134e5, shape)),
                                           f32(rng.uniform(0, 3500, shape))]
        fpar = rng.integers(0, 101, shape).astype(np.uint8)
        lai = rng.integers(0, 71, shape).astype(np.uint8)
        for math, name in ((mod16_amd._lib.MATH_FAST, 'fast'), (mod16_amd._lib.MATH_MIXED, 'mixed')):
            mod16_amd.evapotranspiration_raw(table, cls, *raw, fpar, lai, math=math)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                mod16_amd.evapotranspiration_raw(table, cls, *raw, fpar, lai, math=math)
                ts.append(time.perf_counter() - t0)
            best = min(ts)
            print(json.dumps({'form': 'raw float32', 'math': name, 'shape': shape, 'seconds': best, 'pixels_per_s': n / best,
                              'pcie_GBps_both_directions': 67 * n / best / 1e9}), flush=True)


if __name__ == '__main__':
    if 'raw' in sys.argv[1:]:
        raw()
    else:
        main()
