#!/usr/bin/env python3
"""MOD16(params).evapotranspiration(*device tensors): the reference's own signature on rasters
resident in HBM (scalar parameters of one plant functional type, no class raster) -- the plain
vector kernel (et_kernel), HIP events around 10 calls. 129 B/pixel algorithmic as the production
pipeline (the parameters are scalars)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import mod16_amd  # noqa: E402
from mod16_amd.raster import RasterEngine  # noqa: E402
from mod16_amd.utils import restore_bplut, bplut_table  # noqa: E402
from mod16_amd.models import COLLECTION61_BPLUT  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10800 * 21600
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    from mod16_amd import _lib
    for dtype, math in (('float64', _lib.MATH_FAST), ('float32', _lib.MATH_FAST), ('float32', _lib.MATH_MIXED)):
        eng = RasterEngine(table, dtype=dtype)
        cls, drv = eng.synth(n, seed=16)
        del cls
        model = mod16_amd.MOD16(dict(zip(mod16_amd.MOD16.required_parameters, (float(v) for v in table[7]))))
        model.math = math          # (the class attribute: MATH_MIXED is the float32 rasters' mixed-precision form)
        ref = None
        if math == _lib.MATH_MIXED:
            plain = mod16_amd.MOD16(model.params)
            ref = [t.double() for t in plain.evapotranspiration(*drv)]
        model.evapotranspiration(*drv)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            out = model.evapotranspiration(*drv)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        bpp = 129 if dtype == 'float64' else 65
        line = {'dtype': dtype, 'math': 'mixed' if math == _lib.MATH_MIXED else 'fast', 'pixels': n, 'ms': ms,
                'GBps': bpp * n / ms / 1e6, 'frac_of_8TBps': bpp * n / ms / 1e6 / 8000}
        if ref is not None:       # against the float64 arithmetic on the same float32 tensors
            errs = []
            for got, want in zip(out, ref):
                got = got.double()
                same_nan = bool(torch.equal(torch.isnan(got), torch.isnan(want)))
                rel = torch.nan_to_num((got - want).abs() / want.abs(), nan=0.0, posinf=0.0)
                errs.append({'nan_masks_equal': same_nan, 'max_rel': float(rel.max()), 'n_gt_1e-4': int((rel > 1e-4).sum())})
            line['vs_float64_arithmetic'] = errs
            del ref
        print(json.dumps(line), flush=True)
        del drv, out, eng
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
