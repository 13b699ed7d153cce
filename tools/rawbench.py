#!/usr/bin/env python3
"""Kernel timing of the raw-driver forward run (mod16_et_raw_*, N1) on
device-resident synthetic fields; torch events on the launch stream."""
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
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10800
    n = rows * 43200
    eng = RasterEngine(bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250))
    cls, drv = eng.synth(n, seed=16)
    g = torch.Generator(device='cuda').manual_seed(1)
    u = lambda lo, hi: torch.empty(n, dtype=torch.float64, device='cuda').uniform_(lo, hi, generator=g)
    raw = drv[:9] + [u(0.001, 0.02), u(0.001, 0.02), u(7e4, 1.0134e5), u(7e4, 1.0134e5), u(0, 3500)]
    fpar = torch.randint(0, 101, (n,), dtype=torch.uint8, device='cuda', generator=g)
    lai = torch.randint(0, 71, (n,), dtype=torch.uint8, device='cuda', generator=g)
    hours = u(8, 16)
    for label, kw, bpp in (('day+night', {}, 14 * 8 + 3 + 16), ('total8', {'day_hours': hours}, 15 * 8 + 3 + 8)):
        outs = {}
        if 'day_hours' in kw:
            outs = {'out_total8': eng.empty(n, 1)[0]}
        else:
            d, g2 = eng.empty(n, 2)
            outs = {'out_day': d, 'out_night': g2}
        eng.run_raw(cls, raw, fpar, lai, **kw, **outs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            eng.run_raw(cls, raw, fpar, lai, **kw, **outs)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(json.dumps({'variant': label, 'pixels': n, 'ms': ms, 'gpix_s': n / ms / 1e6,
                          'bytes_per_pixel': bpp, 'GBps': bpp * n / ms / 1e6}))


if __name__ == '__main__':
    main()
