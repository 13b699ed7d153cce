#!/usr/bin/env python3
"""The worst values of the float32 mixed-precision totals against the float64 arithmetic on the tiled
global grid: where they are, their drivers, and the six components of both arithmetics -- which
component and which period a remaining error belongs to.   python tools/mixed_worst.py [rows] [count]"""
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

NAMES = ('lw_d', 'lw_n', 'sw_d', 'sw_n', 'alb', 't_d', 't_n', 't_ann', 'tmin', 'vpd_d', 'vpd_n', 'pa', 'fpar', 'lai')


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 21600
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    n = rows * 43200
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    mixed = RasterEngine(table, dtype='float32', math=_lib.MATH_MIXED)
    fast = RasterEngine(table, dtype='float32', math=_lib.MATH_FAST)
    r = mixed.synth_tiled(mixed.alloc_tiled(n), seed=16)
    ref = fast.alloc_tiled(n)
    ref.slab.copy_(r.slab)
    mixed.run_tiled(r)
    fast.run_tiled(ref)
    torch.cuda.synchronize()
    mixed.check()
    found = []
    for name, got, want in (('day', r.day, ref.day), ('night', r.night, ref.night)):
        g, w = r.flat(got).double(), ref.flat(want).double()
        err = torch.nan_to_num((g - w).abs() / w.abs(), nan=0.0, posinf=0.0)
        top = torch.topk(err, count)
        for e, i in zip(top.values.tolist(), top.indices.tolist()):
            found.append((e, name, i))
        del g, w, err
    found.sort(reverse=True)
    cls = r.flat(r.cls)
    for e, name, i in found[:count]:
        lo = i - i % 256           # the pixel's piece: the pipeline kernel (mixed arithmetic) takes it
        drv = [r.flat(d)[lo:lo + 256].clone() for d in r.drivers]
        c = cls[lo:lo + 256].clone()
        sep_m, sep_f = mixed.empty(256, 6), fast.empty(256, 6)
        mixed.run(c, drv, out_sep=sep_m)
        fast.run(c, drv, out_sep=sep_f)
        k = i - lo
        print(json.dumps({'period': name, 'pixel': i, 'rel_err': e, 'class': int(c[k]),
                          'mixed_total': float(r.flat(r.day if name == 'day' else r.night)[i]),
                          'fast_total': float(ref.flat(ref.day if name == 'day' else ref.night)[i]),
                          'drivers': {nm: float(d[k]) for nm, d in zip(NAMES, drv)},
                          'components_mixed': [float(s[k]) for s in sep_m],
                          'components_fast': [float(s[k]) for s in sep_f]}), flush=True)


if __name__ == '__main__':
    main()
