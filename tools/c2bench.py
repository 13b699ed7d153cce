#!/usr/bin/env python3
"""BASELINE.json configs[1]: one 1200 x 1200 float64 tile resident on the
device (launch-latency dominated) and the same work as one batched launch
over 64 tiles; HIP events on the launch stream (mod16_time_et)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mod16_amd.raster import RasterEngine  # noqa: E402
from mod16_amd.utils import restore_bplut, bplut_table  # noqa: E402
from mod16_amd.models import COLLECTION61_BPLUT  # noqa: E402


def main():
    eng = RasterEngine(bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250))
    for tiles in (1, 64):
        n = 1200 * 1200 * tiles
        cls, drv, day, night = eng.alloc_raster(n)
        eng.synth(n, seed=16, out=(cls, drv))
        eng.time_kernel(cls, drv, day, night, launches=20)
        ms = min(eng.time_kernel(cls, drv, day, night, launches=200) for _ in range(3))
        print(json.dumps({'tiles_per_launch': tiles, 'pixels': n, 'us_per_launch': round(ms * 1e3, 2),
                          'us_per_tile': round(ms * 1e3 / tiles, 2),
                          'gpix_s': round(n / ms / 1e6, 2), 'GBps_129B': round(129 * n / ms / 1e6, 1)}))


if __name__ == '__main__':
    main()
