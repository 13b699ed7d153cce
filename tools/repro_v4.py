#!/usr/bin/env python3
"""Reproduction of DESIGN.md 5.2: the plain kernels' float32 FAST instances with 4 pixels per
thread (built only with -DMOD16_REPRO_V4) against the shipped 2-pixel ones, same inputs.

  MOD16_LIB=build_variants/v4.so python tools/repro_v4.py dump /tmp/v4.npz
  python tools/repro_v4.py dump /tmp/ref.npz
  python tools/repro_v4.py compare /tmp/ref.npz /tmp/v4.npz
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def dump(path):
    os.environ['MOD16_NO_DMA'] = '1'            # plain kernels only
    from mod16_amd.raster import RasterEngine
    from mod16_amd.utils import restore_bplut, bplut_table
    from mod16_amd.models import COLLECTION61_BPLUT
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    eng = RasterEngine(table, dtype='float32')
    n = 4 * 300000
    cls, drv = eng.synth(n, seed=21)
    out = {}
    day, night = eng.run(cls, drv)
    out['day'], out['night'] = day.cpu().numpy(), night.cpu().numpy()
    sep = eng.empty(n, 6)
    eng.run(cls, drv, out_sep=sep)
    for k in range(6):
        out['sep%d' % k] = sep[k].cpu().numpy()
    pet = eng.run_pet(cls, drv)
    for k in range(4):
        out['pet%d' % k] = pet[k].cpu().numpy()
    eng.check()
    np.savez(path, **out)


def compare(a, b):
    A, B = np.load(a), np.load(b)
    for k in A.files:
        x, y = A[k], B[k]
        same = (x == y) | (np.isnan(x) & np.isnan(y))
        bad = np.flatnonzero(~same)
        msg = '%-6s %8d of %d differ' % (k, bad.size, x.size)
        if bad.size:
            lanes = np.bincount((bad // 4) % 64, minlength=64)
            msg += '; pixel-in-thread histogram %s; first %s: %r vs %r; lanes with errors %d of 64' % (
                np.bincount(bad % 4, minlength=4).tolist(), bad[:4].tolist(), x[bad[:3]].tolist(),
                y[bad[:3]].tolist(), int((lanes > 0).sum()))
        print(msg)


if __name__ == '__main__':
    if sys.argv[1] == 'dump':
        dump(sys.argv[2])
    else:
        compare(sys.argv[2], sys.argv[3])
