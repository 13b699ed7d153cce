#!/usr/bin/env python3
"""BASELINE.json configs[1]: one 1200 x 1200 float64 tile resident on the device per
launch (latency-bound) and 64 tiles per launch; HIP events around replays of the
captured step (mod16_time_graph). MOD16_STATIC_BELOW selects where the static schedule of small rasters ends."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mod16_amd.raster import RasterEngine  # noqa: E402
from mod16_amd.utils import restore_bplut, bplut_table  # noqa: E402
from mod16_amd.models import COLLECTION61_BPLUT  # noqa: E402


def main():
    # MOD16_STATIC_BELOW etc. are read by the experiments build of the library only
    eng = RasterEngine(bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250),
                       experiments=any(k in os.environ for k in ('MOD16_STATIC_BELOW', 'MOD16_RUN_SHIFT', 'MOD16_STREAM_BLOCKS')))
    diag = torch.zeros(8, dtype=torch.float64, device='cuda')
    for tiles in (1, 4, 64):
        n = 1200 * 1200 * tiles
        r = eng.synth_tiled(eng.alloc_tiled(n), seed=16)
        step = eng.bind_tiled(r, diag)
        for _ in range(20):
            step()
        us = min(step.time(200) for _ in range(3)) * 1e3
        direct = min(eng.time_tiled(r, 200, diag) for _ in range(3)) * 1e3
        plain = min(eng.time_tiled(r, 200) for _ in range(3)) * 1e3
        print(json.dumps({'tiles_per_launch': tiles, 'pixels': n, 'us_per_step': round(us, 2),
                          'us_direct_launches_with_diag': round(direct, 2), 'us_direct_launches_no_diag': round(plain, 2),
                          'us_per_tile': round(us / tiles, 2), 'GBps_129B': round(129 * n / us / 1e3, 1),
                          'static_below': os.environ.get('MOD16_STATIC_BELOW', 'default')}), flush=True)


if __name__ == '__main__':
    main()
