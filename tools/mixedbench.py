#!/usr/bin/env python3
"""BASELINE.json configs[4] in one line per build of the library: the float32 mixed-precision totals
kernel on the tiled global grid -- step time (graph replays, HIP events; the whole step: pipeline
kernel + the pass over flagged pieces + the diagnostics sum), share of the HBM peak, and its values
against the float64 arithmetic (the FAST kernel of the SAME build) on every pixel: masks, counts of
relative errors above 1e-6 ... 1e-3, the largest, and how many values went through the pass behind
the loop.

  [MOD16_LIB=build_variants/x.so] python tools/mixedbench.py [--rows 21600]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch  # noqa: E402
from mod16_amd import _lib  # noqa: E402
from mod16_amd.raster import RasterEngine  # noqa: E402
from mod16_amd.utils import restore_bplut, bplut_table  # noqa: E402
from mod16_amd.models import COLLECTION61_BPLUT  # noqa: E402

THRESHOLDS = ('1e-6', '1e-5', '1e-4', '1e-3')


def compare(got, ref):
    res = {'nan_masks_equal': True, 'zero_mask_mismatches': 0, 'max_rel_err': 0.0}
    res.update({'n_gt_' + t: 0 for t in THRESHOLDS})
    step = 1 << 27
    for lo in range(0, ref.numel(), step):
        a, b = got[lo:lo + step], ref[lo:lo + step]
        res['nan_masks_equal'] &= bool(torch.equal(torch.isnan(a), torch.isnan(b)))
        res['zero_mask_mismatches'] += int(((a == 0) != (b == 0)).sum())
        err = (a.double() - b.double()).abs_()
        err = torch.nan_to_num_(err.div_(b.double().abs_()), nan=0.0, posinf=0.0)
        res['max_rel_err'] = max(res['max_rel_err'], float(err.max()))
        for t in THRESHOLDS:
            res['n_gt_' + t] += int((err > float(t)).sum())
        del err
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=21600)
    ap.add_argument('--launches', type=int, default=10)
    args = ap.parse_args()
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    n = args.rows * 43200
    diag = torch.zeros(8, dtype=torch.float64, device='cuda')
    for lib in [None]:
        mixed = RasterEngine(table, dtype='float32', math=_lib.MATH_MIXED)
        trusted = RasterEngine(table, dtype='float32', math=_lib.MATH_MIXED, trusted=True)
        fast = RasterEngine(table, dtype='float32', math=_lib.MATH_FAST)
        r = mixed.synth_tiled(mixed.alloc_tiled(n), seed=16)
        ref = fast.alloc_tiled(n)
        ref.slab.copy_(r.slab)
        out = {'lib': _lib.LIB_PATH, 'build_id': _lib.build_id(), 'pixels': n}
        steps = {'mixed': mixed.bind_tiled(r, diag), 'mixed_trusted': trusted.bind_tiled(r, diag),
                 'fast': fast.bind_tiled(ref, diag)}
        for s in steps.values():
            s()
        torch.cuda.synchronize()
        times = {k: [] for k in steps}
        for _ in range(3):
            for k, s in steps.items():
                times[k].append(s.time(args.launches))
        for k, v in times.items():
            out[k + '_ms'] = round(min(v), 4)
            out[k + '_frac'] = round(65.0 * n / min(v) / 1e6 / 8000.0, 4)
        steps['mixed']()                       # the guarded form's values are the ones compared
        torch.cuda.synchronize()
        mixed.check()
        fast.check()
        full = None
        for got, want in ((r.day, ref.day), (r.night, ref.night)):
            res = compare(r.flat(got), ref.flat(want))
            if full is None:
                full = res
            else:
                full['nan_masks_equal'] &= res['nan_masks_equal']
                full['zero_mask_mismatches'] += res['zero_mask_mismatches']
                full['max_rel_err'] = max(full['max_rel_err'], res['max_rel_err'])
                for t in THRESHOLDS:
                    full['n_gt_' + t] += res['n_gt_' + t]
        out.update(full)
        print(json.dumps(out), flush=True)
        del steps, r, ref, mixed, trusted, fast
        import gc
        gc.collect()
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
