#!/usr/bin/env python3
"""What the domain guard costs on the headline step: the float64 (and float32 mixed) totals kernel
on the tiled global grid with and without MOD16_DOMAIN_TRUSTED, same process, same raster, graph
replays timed with HIP events, interleaved.  python tools/guardcost.py [rows=21600]"""
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
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 21600
    n = rows * 43200
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    for dtype, math, bpp in (('float64', _lib.MATH_FAST, 129), ('float32', _lib.MATH_MIXED, 65)):
        guarded = RasterEngine(table, dtype=dtype, math=math)
        trusted = RasterEngine(table, dtype=dtype, math=math, trusted=True)
        ras = guarded.synth_tiled(guarded.alloc_tiled(n), seed=16)
        diag = torch.zeros(8, dtype=torch.float64, device='cuda')
        steps = {'guarded': guarded.bind_tiled(ras, diag), 'trusted': trusted.bind_tiled(ras, diag)}
        for s in steps.values():
            s()
        torch.cuda.synchronize()
        res = {k: [] for k in steps}
        for _ in range(4):
            for k, s in steps.items():
                res[k].append(s.time(10))
        out = {'dtype': dtype, 'math': 'mixed' if math == _lib.MATH_MIXED else 'fast', 'pixels': n}
        for k, v in res.items():
            ms = min(v)
            out[k + '_ms'] = round(ms, 4)
            out[k + '_frac'] = round(bpp * n / ms / 1e6 / 8000.0, 4)
        out['guard_cost_percent'] = round(100 * (out['guarded_ms'] / out['trusted_ms'] - 1), 2)
        print(json.dumps(out), flush=True)
        del steps, ras
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
