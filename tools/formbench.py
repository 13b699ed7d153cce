#!/usr/bin/env python3
"""One form of the forward run on a tiled raster, launched a few times -- the target of the
rocprofv3 passes of tools/run_form_profiles.sh (kernel stats, VALUBusy, FETCH_SIZE, WRITE_SIZE).

  python tools/formbench.py FORM [dtype=float64] [math=fast|mixed] [rows=10800] [launches=6]
  FORM: totals pet components totals_components raw raw_total8 raw_total8_hours
"""
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

FORMS = {'totals': _lib.FORM_TOTALS, 'pet': _lib.FORM_PET, 'components': _lib.FORM_COMPONENTS,
         'totals_components': _lib.FORM_TOTALS_COMPONENTS, 'raw': _lib.FORM_RAW,
         'raw_total8': _lib.FORM_RAW_TOTAL8, 'raw_total8_hours': _lib.FORM_RAW_TOTAL8_HOURS}


def main():
    form = FORMS[sys.argv[1]]
    dtype = sys.argv[2] if len(sys.argv) > 2 else 'float64'
    math = {'mixed': _lib.MATH_MIXED}.get(sys.argv[3] if len(sys.argv) > 3 else 'fast', _lib.MATH_FAST)
    rows = int(sys.argv[4]) if len(sys.argv) > 4 else 10800
    launches = int(sys.argv[5]) if len(sys.argv) > 5 else 6
    n = rows * 43200
    eng = RasterEngine(bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250), dtype=dtype, math=math)
    esz = eng.np_dtype.itemsize
    cls, drv = eng.synth(n, seed=16)
    g = torch.Generator(device='cuda').manual_seed(1)
    u = lambda lo, hi: torch.empty(n, dtype=eng.dtype, device='cuda').uniform_(lo, hi, generator=g)
    r = eng.alloc_tiled(n, form=form)
    if form >= _lib.FORM_RAW:
        wide = drv[:9] + [u(0.001, 0.02), u(0.001, 0.02), u(7e4, 1.0134e5), u(7e4, 1.0134e5), u(0, 3500), u(8, 16)]
        extra = [torch.randint(0, 101, (n,), dtype=torch.uint8, device='cuda', generator=g),
                 torch.randint(0, 71, (n,), dtype=torch.uint8, device='cuda', generator=g)]
    else:
        wide, extra = drv, []
    for dst, src in zip(r.wide, wide):
        r.put(dst, src)
    for dst, src in zip(r.bytes, [cls] + extra):
        r.put(dst, src)
    del wide, drv, extra
    torch.cuda.empty_cache()
    hours = 11.5 if form == _lib.FORM_RAW_TOTAL8 else None
    eng.run_form_tiled(r, day_hours=hours)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        eng.run_form_tiled(r, day_hours=hours)
    e1.record()
    torch.cuda.synchronize()
    eng.check()
    nw, nb, no = _lib.FORM_SHAPE[form]
    bpp = nw * esz + nb + no * esz
    ms = e0.elapsed_time(e1) / launches
    line = {'form': sys.argv[1], 'dtype': dtype, 'math': 'mixed' if math == _lib.MATH_MIXED else 'fast',
            'pixels': n, 'bytes_per_pixel': bpp, 'ms': round(ms, 3),
            'GBps': round(bpp * n / ms / 1e6, 1), 'frac_8TBs': round(bpp * n / ms / 1e6 / 8000, 4)}
    if os.environ.get('FORMBENCH_SENSORS') == '1':      # shader clock and package power under this form (bench.py's reader)
        import bench
        u = bench.device_under_load(torch, lambda: eng.run_form_tiled(r, day_hours=hours), max(20, int(2000 / ms)), True)
        if u:
            line.update(sclk_mhz=round(u['sclk_mhz']), power_w=round(u['power_w'], 1), power_cap_w=u['power_cap_w'])
    print(json.dumps(line))


if __name__ == '__main__':
    main()
