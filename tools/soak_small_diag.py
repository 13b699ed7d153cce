#!/usr/bin/env python3
"""Soak test of the small rasters' in-kernel sum of the diagnostics partials (the block that
finishes last adds them up; agent-scope stores and loads, no fences): 60 x 64 launches per
size and data type, three streams taking turns, every result compared bit for bit with the
first and once with the stand-alone reduction."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mod16_amd.raster import RasterEngine
from mod16_amd.utils import restore_bplut, bplut_table
from mod16_amd.models import COLLECTION61_BPLUT
table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
bad = 0
for dtype in ('float64', 'float32'):
    eng = RasterEngine(table, dtype=dtype)
    for n in (1200 * 1200, 64 * 2 * 7, 2048 * 128 * 3 + 128 * 5):
        r = eng.synth_tiled(eng.alloc_tiled(n), seed=9)
        ref = torch.zeros(8, dtype=torch.float64, device='cuda')
        eng.run_tiled(r, diag=ref)
        torch.cuda.synchronize()
        alone = eng.diagnostics(r.flat(r.day), r.flat(r.night))
        torch.cuda.synchronize()
        assert np.allclose(ref.cpu().numpy()[:2], alone.cpu().numpy()[:2], rtol=1e-12) and np.array_equal(ref.cpu().numpy()[2:], alone.cpu().numpy()[2:])
        ds = [torch.zeros(8, dtype=torch.float64, device='cuda') for _ in range(64)]
        streams = [torch.cuda.Stream() for _ in range(3)]
        for rep in range(60):
            for i, d in enumerate(ds):
                d.fill_(-1.0)
                with torch.cuda.stream(streams[i % 3]):
                    eng.run_tiled(r, diag=d)
            torch.cuda.synchronize()
            for d in ds:
                if not torch.equal(d, ref):
                    bad += 1
        print(dtype, n, 'launches', 60 * 64, 'mismatches so far', bad, flush=True)
eng.check()
print('BAD' if bad else 'OK')
