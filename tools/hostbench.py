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
    for shape in ((1200, 1200), (4800, 4800), (9600, 9600)):
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


if __name__ == '__main__':
    main()
