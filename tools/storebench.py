#!/usr/bin/env python3
"""SURVEY.md 8f N4 with a measurement: a raster time series on disk (Cal-Val field set,
raw drivers, one .npy per dataset) run through mod16_amd.io.run_store -- file -> pinned ->
H2D -> fused raw-driver kernel -> D2H -> file, several workers deep. Prints one JSON report
per worker count: every stage's rate while busy, wall time, sustained pixels/s.

  python tools/storebench.py [--dir /tmp] [--pixels 93312000] [--steps 4] [--dtype float32]
                             [--workers 1,4,8] [--tile 4194304]

The synthetic fields are produced on the GPU and written once (set-up, not measured); the
files then sit in the page cache, so `read` is a copy out of the page cache, not disk speed.
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mod16_amd import io  # noqa: E402
from mod16_amd.utils import restore_bplut, bplut_table  # noqa: E402
from mod16_amd.models import COLLECTION61_BPLUT  # noqa: E402


def fill(store, chunk=1 << 24):
    """Random raw fields in physical ranges, generated on the GPU chunk by chunk."""
    g = torch.Generator(device='cuda').manual_seed(16)
    tdt = torch.float32 if store.dtype == np.float32 else torch.float64
    T, N = store.n_steps, store.n_pixels
    ranges = {0: (-100, 0), 1: (-50, 0), 2: (0, 360), 4: (0.1, 0.22), 5: (255, 305), 6: (250, 300),
              8: (245, 298), 9: (0.001, 0.02), 10: (0.001, 0.02), 11: (70000, 101340), 12: (70000, 101340)}
    maps = {name: store.array(name, 'r+') for _, name in io.DYNAMIC_FIELDS}
    u8 = {name: store.array(name, 'r+') for name in (io.FPAR, io.LAI)}

    def rnd(n, lo, hi):
        return (torch.rand(n, generator=g, device='cuda', dtype=tdt) * (hi - lo) + lo).cpu().numpy()

    for p0 in range(0, N, chunk):
        m = min(chunk, N - p0)
        for t in range(T):
            for idx, name in io.DYNAMIC_FIELDS:
                maps[name][t, p0:p0 + m] = rnd(m, *ranges[idx])
            u8[io.FPAR][t, p0:p0 + m] = torch.randint(0, 101, (m,), generator=g, device='cuda', dtype=torch.uint8).cpu().numpy()
            u8[io.LAI][t, p0:p0 + m] = torch.randint(0, 71, (m,), generator=g, device='cuda', dtype=torch.uint8).cpu().numpy()
        store.array('MERRA2/T10M_annual', 'r+')[p0:p0 + m] = rnd(m, 265, 300)
        store.array('state/elevation_m', 'r+')[p0:p0 + m] = rnd(m, -50, 3500)
        cls = torch.randint(1, 11, (m,), generator=g, device='cuda', dtype=torch.uint8)
        store.array(io.PFT, 'r+')[p0:p0 + m] = cls.cpu().numpy()
        print('filled %d / %d pixels' % (p0 + m, N), file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dir', default=tempfile.gettempdir())
    ap.add_argument('--pixels', type=int, default=43200 * 2160)
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--dtype', default='float32')
    ap.add_argument('--workers', default='1,4,8')
    ap.add_argument('--tile', type=int, default=1 << 22)
    ap.add_argument('--readers', default='1,3', help='reader threads per worker, comma-separated')
    args = ap.parse_args()
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    root = tempfile.mkdtemp(prefix='mod16_store_', dir=args.dir)
    try:
        t0 = time.perf_counter()
        store = io.RasterStore.create(root, args.steps, args.pixels, np.dtype(args.dtype))
        fill(store)
        esz = np.dtype(args.dtype).itemsize
        size = args.steps * args.pixels * (11 * esz + 2 + 2 * esz) + args.pixels * (2 * esz + 1)
        print(json.dumps({'store': root, 'bytes_in_and_out': size, 'fill_seconds': time.perf_counter() - t0}), flush=True)
        for w in [int(x) for x in args.workers.split(',')]:
            for rd in [int(x) for x in args.readers.split(',')]:
                rep = io.run_store(table, root, tile_pixels=args.tile, workers=w, readers=rd)
                print(json.dumps(rep), flush=True)
        # spot check: the last tile of the last step against a direct HOST-mode call
        import mod16_amd
        p0 = max(0, args.pixels - 100000)
        raw = [None] * 14
        for idx, name in io.DYNAMIC_FIELDS:
            raw[idx] = np.array(store.array(name)[-1, p0:])
        raw[3] = np.zeros_like(raw[0])
        raw[7] = np.array(store.array('MERRA2/T10M_annual')[p0:])
        raw[13] = np.array(store.array('state/elevation_m')[p0:])
        want = mod16_amd.evapotranspiration_raw(
            table, np.array(store.array(io.PFT)[p0:]), *raw,
            np.array(store.array(io.FPAR)[-1, p0:]), np.array(store.array(io.LAI)[-1, p0:]))
        got = np.array(store.array(io.OUT_DAY)[-1, p0:])
        print(json.dumps({'spot_check_equal_to_direct_call': bool(np.array_equal(got, want[0], equal_nan=True))}))
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == '__main__':
    main()
