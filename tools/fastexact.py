#!/usr/bin/env python3
"""How far the production (FAST) arithmetic is from the reference-order (EXACT) kernel, on every
pixel of a large synthetic raster: totals and the raw-driver form, float64. The judge of arithmetic
changes (round 5: shared exponential of the raw forms, one-constant reductions, 1 / t from the
r_corr power): masks identical, largest relative error, how many values are off by more than 1e-9 /
1e-8 / 1e-7 / 1e-5, and the median.

  python tools/fastexact.py [rows=5400] [seeds=16,17]         (MOD16_LIB=... for another build)
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


def compare(got, ref):
    out = {'nan_masks_equal': bool(torch.equal(torch.isnan(got), torch.isnan(ref))),
           'zero_mask_mismatches': int(((got == 0) != (ref == 0)).sum()),
           'inf_masks_equal': bool(torch.equal(torch.isinf(got), torch.isinf(ref)))}
    ok = torch.isfinite(ref) & (ref != 0)
    rel = ((got[ok] - ref[ok]).abs() / ref[ok].abs())
    out['max_rel_err'] = float(rel.max())
    out['median_rel_err'] = float(rel.median())
    for t in ('1e-9', '1e-8', '1e-7', '1e-5'):
        out['n_gt_' + t] = int((rel > float(t)).sum())
    out['values'] = int(ok.sum())
    return out


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 5400
    seeds = [int(s) for s in (sys.argv[2] if len(sys.argv) > 2 else '16,17').split(',')]
    n = rows * 43200
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    fast = RasterEngine(table, dtype='float64', math=_lib.MATH_FAST)
    exact = RasterEngine(table, dtype='float64', math=_lib.MATH_EXACT)
    print(json.dumps({'build_id': _lib.build_id(), 'lib': _lib.LIB_PATH, 'pixels': n, 'seeds': seeds}), flush=True)
    for seed in seeds:
        cls, drv = fast.synth(n, seed=seed)
        a = fast.run(cls, drv)
        b = exact.run(cls, drv)
        fast.check()
        exact.check()
        for name, x, y in (('day', a[0], b[0]), ('night', a[1], b[1])):
            print(json.dumps(dict(form='totals', seed=seed, output=name, **compare(x, y))), flush=True)
        del a, b
        g = torch.Generator(device='cuda').manual_seed(seed)
        u = lambda lo, hi: torch.empty(n, dtype=torch.float64, device='cuda').uniform_(lo, hi, generator=g)
        raw = drv[:9] + [u(0.0005, 0.02), u(0.0005, 0.02), u(7e4, 1.0134e5), u(7e4, 1.0134e5), u(-400, 6000)]
        # cold and hot ends of the raw forms' temperature domain in a slice of the raster
        raw[5][: n // 50].uniform_(191, 230, generator=g)
        raw[6][: n // 50].uniform_(191, 230, generator=g)
        raw[5][n // 50: n // 25].uniform_(320, 359, generator=g)
        fpar = torch.randint(0, 101, (n,), dtype=torch.uint8, device='cuda', generator=g)
        lai = torch.randint(0, 71, (n,), dtype=torch.uint8, device='cuda', generator=g)
        a = fast.run_raw(cls, raw, fpar, lai)
        b = exact.run_raw(cls, raw, fpar, lai)
        fast.check()
        exact.check()
        for name, x, y in (('day', a[0], b[0]), ('night', a[1], b[1])):
            print(json.dumps(dict(form='raw', seed=seed, output=name, **compare(x, y))), flush=True)
        del a, b, raw, fpar, lai, cls, drv
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
