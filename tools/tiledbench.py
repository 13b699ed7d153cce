#!/usr/bin/env python3
"""Kernel time of the production pipeline on plain arrays (one slab, arrays 0.5 GiB
apart) against tiled rasters of several tile sizes, same process, same field.

  python tools/tiledbench.py [--rows 21600] [--dtype float64] [--math fast] [--tiles 4096,8192,16384] [--experiments]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mod16_amd import _lib  # noqa: E402
from mod16_amd.raster import RasterEngine  # noqa: E402
from mod16_amd.utils import restore_bplut, bplut_table  # noqa: E402
from mod16_amd.models import COLLECTION61_BPLUT  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=21600)
    ap.add_argument('--dtype', default='float64')
    ap.add_argument('--math', default='fast')
    ap.add_argument('--tiles', default='')
    ap.add_argument('--launches', type=int, default=10)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--no-plain', action='store_true')
    ap.add_argument('--experiments', action='store_true',
                    help='the -DMOD16_EXPERIMENTS build: MOD16_RUN_SHIFT, MOD16_STATIC_BELOW, ... from the environment apply')
    args = ap.parse_args()
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    math = {'mixed': _lib.MATH_MIXED}.get(args.math, _lib.MATH_FAST)
    eng = RasterEngine(table, dtype=args.dtype, math=math, experiments=args.experiments)
    n = args.rows * 43200
    bpp = eng.bytes_per_pixel
    diag = torch.zeros(8, dtype=torch.float64, device='cuda')

    def report(name, step):
        step()
        torch.cuda.synchronize()
        ms = [round(step.time(args.launches), 4) for _ in range(args.rounds)]
        best = min(ms)
        print(json.dumps({'layout': name, 'ms': ms, 'best_ms': best, 'GBps': round(bpp * n / best / 1e6, 1),
                          'frac_8TBs': round(bpp * n / best / 8e9, 4)}), flush=True)

    if not args.no_plain:
        cls, drv, day, night = eng.alloc_raster(n, 512 << 20)
        eng.synth(n, seed=16, out=(cls, drv))
        report('plain arrays, +0.5 GiB apart', eng.bind(cls, drv, day, night, diag))
        del cls, drv, day, night
        torch.cuda.empty_cache()
    esz = 8 if args.dtype == 'float64' else 4
    tiles = [int(t) for t in args.tiles.split(',')] if args.tiles else [eng.TILE_BYTES // esz]
    for tile in tiles:
        r = eng.synth_tiled(eng.alloc_tiled(n, tile=tile), seed=16)
        report('tiled, %d px = %d KiB per field' % (tile, tile * esz // 1024), eng.bind_tiled(r, diag))
        del r
        torch.cuda.empty_cache()
    eng.check()


if __name__ == '__main__':
    main()
