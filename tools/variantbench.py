#!/usr/bin/env python3
"""Kernel timing of the other forms of the forward run on device-resident
synthetic rasters: potential ET (N3), separate components, raw drivers (N1),
each on the production pipeline (et_stream_kernel) over plain arrays and over the
engine's tiled layout, and on the plain kernels of the same library (a MOD16_NO_DMA=1
context). torch events on the launch
stream; bytes per pixel are the algorithmic ones.

  python tools/variantbench.py [rows=10800] [dtype=float64] [mixed]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mod16_amd import _lib  # noqa: E402
from mod16_amd.raster import RasterEngine  # noqa: E402
from mod16_amd.utils import restore_bplut, bplut_table  # noqa: E402
from mod16_amd.models import COLLECTION61_BPLUT  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10800
    dtype = sys.argv[2] if len(sys.argv) > 2 else 'float64'
    math = {'mixed': _lib.MATH_MIXED}.get(sys.argv[3] if len(sys.argv) > 3 else '', _lib.MATH_FAST)
    n = rows * 43200
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    eng = RasterEngine(table, dtype=dtype, math=math)
    os.environ['MOD16_NO_DMA'] = '1'
    plain = RasterEngine(table, dtype=dtype)
    plain.ctx = _lib.Context(0, experiments=True)
    plain.ctx.set_bplut(np.ascontiguousarray(table, np.float64))
    del os.environ['MOD16_NO_DMA']
    esz = eng.np_dtype.itemsize
    cls, drv, day, night = eng.alloc_raster(n)
    eng.synth(n, seed=16, out=(cls, drv))
    g = torch.Generator(device='cuda').manual_seed(1)
    u = lambda lo, hi: torch.empty(n, dtype=eng.dtype, device='cuda').uniform_(lo, hi, generator=g)
    raw = drv[:9] + [u(0.001, 0.02), u(0.001, 0.02), u(7e4, 1.0134e5), u(7e4, 1.0134e5), u(0, 3500)]
    fpar = torch.randint(0, 101, (n,), dtype=torch.uint8, device='cuda', generator=g)
    lai = torch.randint(0, 71, (n,), dtype=torch.uint8, device='cuda', generator=g)
    hours = u(8, 16)
    extra = eng.empty(n, 6)
    cases = [
        ('totals (production kernel)', 14 * esz + 1 + 2 * esz,
         lambda e: e.run(cls, drv, day, night), _lib.FORM_TOTALS),
        ('pet: day, night, pet day, pet night', 14 * esz + 1 + 4 * esz,
         lambda e: e.run_pet(cls, drv, out=(day, night, extra[0], extra[1])), _lib.FORM_PET),
        ('separate: six components', 14 * esz + 1 + 6 * esz,
         lambda e: e.run(cls, drv, out_sep=extra), _lib.FORM_COMPONENTS),
        ('separate: totals + six components', 14 * esz + 1 + 8 * esz,
         lambda e: e.run(cls, drv, day, night, out_sep=extra), _lib.FORM_TOTALS_COMPONENTS),
        ('raw drivers: day, night', 14 * esz + 3 + 2 * esz,
         lambda e: e.run_raw(cls, raw, fpar, lai, out_day=day, out_night=night), _lib.FORM_RAW),
        ('raw drivers: day, night, 8-day total', 15 * esz + 3 + 3 * esz,
         lambda e: e.run_raw(cls, raw, fpar, lai, day_hours=hours, out_day=day, out_night=night,
                             out_total8=extra[0]), _lib.FORM_RAW_TOTAL8_HOURS),
    ]
    for label, bpp, fn, form in cases:
        ms = timed(lambda: fn(eng))
        ms0 = timed(lambda: fn(plain))
        eng.check()
        plain.check()
        # the same form on the engine's tiled layout (mod16_et_form_tiled_*)
        r = eng.alloc_tiled(n, form=form)
        wide = (raw + [hours])[:len(r.wide)] if form >= _lib.FORM_RAW else drv
        for dst, src in zip(r.wide, wide):
            r.put(dst, src)
        for dst, src in zip(r.bytes, [cls, fpar, lai]):
            r.put(dst, src)
        ms_t = timed(lambda: eng.run_form_tiled(r))
        eng.check()
        del r
        torch.cuda.empty_cache()
        print(json.dumps({'form': label, 'dtype': dtype, 'math': 'mixed' if math == _lib.MATH_MIXED else 'fast', 'pixels': n, 'bytes_per_pixel': bpp,
                          'pipeline_ms': round(ms, 3), 'pipeline_GBps': round(bpp * n / ms / 1e6, 1),
                          'pipeline_frac_8TBs': round(bpp * n / ms / 1e6 / 8000, 4),
                          'tiled_ms': round(ms_t, 3), 'tiled_GBps': round(bpp * n / ms_t / 1e6, 1),
                          'tiled_frac_8TBs': round(bpp * n / ms_t / 1e6 / 8000, 4),
                          'plain_ms': round(ms0, 3), 'plain_GBps': round(bpp * n / ms0 / 1e6, 1)}),
              flush=True)


if __name__ == '__main__':
    main()
