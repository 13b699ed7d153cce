#!/usr/bin/env python3
"""How far the pipeline instances (et_stream_kernel) and the plain kernels (et_kernel, et_raw_kernel;
MOD16_NO_DMA=1 context) of one build are apart on the same pixels: per form and output, the number
of values that differ at all and the largest relative difference. Both are instantiated from one
pixel function with implicit contraction off, so the expectation is ZERO differing values; what
this prints is what tests/test_gpu_stream.py may assert."""
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

N = 64 * 16 * 2 * 37 + 64 * 5 + 3


def plain_engine(table, dtype):
    os.environ['MOD16_NO_DMA'] = '1'
    try:
        eng = RasterEngine(table, dtype=dtype)
        eng.ctx = _lib.Context(0, experiments=True)
        eng.ctx.set_bplut(np.ascontiguousarray(table, np.float64))
        return eng
    finally:
        del os.environ['MOD16_NO_DMA']


def report(what, got, ref):
    for k, (g, w) in enumerate(zip(got, ref)):
        g, w = g.double(), w.double()
        same = (g == w) | (torch.isnan(g) & torch.isnan(w))
        ok = torch.isfinite(w) & (w != 0)
        rel = float(((g[ok] - w[ok]).abs() / w[ok].abs()).max()) if ok.any() else 0.0
        print('%-34s output %d: %7d of %d values differ, largest relative difference %.2e'
              % (what, k, int((~same).sum()), g.numel(), rel))


def main():
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    for dtype in ('float64', 'float32'):
        eng, ref = RasterEngine(table, dtype=dtype), plain_engine(table, dtype)
        cls, drv = eng.synth(N, seed=31)
        report(dtype + ' totals', eng.run(cls, drv), ref.run(cls, drv))
        report(dtype + ' potential ET', eng.run_pet(cls, drv), ref.run_pet(cls, drv))
        a, b = eng.empty(N, 6), ref.empty(N, 6)
        eng.run(cls, drv, None, None, out_sep=a)
        ref.run(cls, drv, None, None, out_sep=b)
        report(dtype + ' components', a, b)
        rng = np.random.default_rng(33)
        t_d = rng.uniform(255, 305, N)
        t_n = t_d - rng.uniform(0, 12, N)
        raw = [rng.uniform(-100, 0, N), rng.uniform(-50, 0, N), rng.uniform(0, 360, N), np.zeros(N),
               rng.uniform(0.1, 0.22, N), t_d, t_n, rng.uniform(265, 300, N), t_n - rng.uniform(0, 3, N),
               rng.uniform(5e-4, 2e-2, N), rng.uniform(5e-4, 2e-2, N),
               rng.uniform(70000, 101340, N), rng.uniform(70000, 101340, N), rng.uniform(-50, 4500, N)]
        d_raw = [torch.from_numpy(np.ascontiguousarray(x, eng.np_dtype)).cuda() for x in raw]
        fpar = torch.from_numpy(rng.integers(0, 101, N).astype(np.uint8)).cuda()
        lai = torch.from_numpy(rng.integers(0, 70, N).astype(np.uint8)).cuda()
        hours = torch.from_numpy(np.ascontiguousarray(rng.uniform(6, 18, N), eng.np_dtype)).cuda()
        report(dtype + ' raw drivers + 8-day total', eng.run_raw(cls, d_raw, fpar, lai, day_hours=hours),
               ref.run_raw(cls, d_raw, fpar, lai, day_hours=hours))


if __name__ == '__main__':
    main()
