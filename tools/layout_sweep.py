#!/usr/bin/env python3
"""Kernel time against the layout of the raster slab inside ONE allocation of
one process: extra bytes between successive arrays (stagger) and the offset of
the first array inside the allocation (shift). Separates what the layout can
control from what the physical placement of the allocation decides.

  python tools/layout_sweep.py [rows=21600]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mod16_amd.raster import RasterEngine  # noqa: E402
from mod16_amd.utils import restore_bplut, bplut_table  # noqa: E402
from mod16_amd.models import COLLECTION61_BPLUT  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 21600
    n = rows * 43200
    eng = RasterEngine(bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250))
    KiB = 1024
    MiB = 1024 * KiB
    if len(sys.argv) > 2 and sys.argv[2] == 'coarse':
        staggers = [33 * KiB] + [m * MiB + 33 * KiB for m in (2, 3, 5, 8, 17, 32, 97, 128, 512, 1024)] + [33 * KiB]
        shifts = [0]
    elif len(sys.argv) > 2 and sys.argv[2] == 'fine':
        lo, hi, step = (int(x) for x in sys.argv[3:6])          # MiB
        staggers = [m * MiB + 33 * KiB for m in range(lo, hi + 1, step)]
        shifts = [0]
    elif len(sys.argv) > 2 and sys.argv[2] == 'gib':
        staggers = [m * 128 * MiB + 33 * KiB for m in range(0, 25)]
        shifts = [0]
    else:
        staggers = [33 * KiB, 1 * KiB, 5 * KiB, 9 * KiB, 17 * KiB, 65 * KiB, 129 * KiB, 257 * KiB, 1025 * KiB, 0, 4 * KiB]
        shifts = [0, 1 << 20, 37 << 20]
    if not (len(sys.argv) > 2 and sys.argv[2] == 'outputs'):
        max_per = (n * 8 + 4095) // 4096 * 4096 + max(staggers)
        big = torch.empty(16 * max_per + n + max(shifts) + 4096, dtype=torch.uint8, device='cuda')
    if len(sys.argv) > 2 and sys.argv[2] == 'outputs':
        # drivers at a fixed spacing, the two output arrays moved as a block
        extra = int(sys.argv[3]) * MiB if len(sys.argv) > 3 else 512 * MiB
        per = (n * 8 + 4095) // 4096 * 4096 + 33 * KiB + extra
        big2 = torch.empty(18 * per + n + 4096 + 2048 * MiB, dtype=torch.uint8, device='cuda')
        views = [big2[k * per:k * per + n * 8].view(torch.float64) for k in range(14)]
        cls = big2[14 * per:14 * per + n]
        eng.synth(n, seed=16, out=(cls, views))
        base_out = 14 * per + (n + 4095) // 4096 * 4096 + 33 * KiB
        for m in range(0, 2049, 128):
            o0 = base_out + m * MiB
            day = big2[o0:o0 + n * 8].view(torch.float64)
            night = big2[o0 + per:o0 + per + n * 8].view(torch.float64)
            eng.time_kernel(cls, views, day, night, launches=2)
            ms = eng.time_kernel(cls, views, day, night, launches=8)
            print(json.dumps({'outputs_moved_MiB': m, 'drivers_extra_MiB': extra // MiB, 'ms': round(ms, 3)}), flush=True)
        return
    for shift in shifts:
        for st in staggers:
            per = (n * 8 + 4095) // 4096 * 4096 + st
            base = big[shift:]
            views = [base[k * per:k * per + n * 8].view(torch.float64) for k in range(16)]
            cls = base[16 * per:16 * per + n]
            drv, day, night = views[:14], views[14], views[15]
            eng.synth(n, seed=16, out=(cls, drv))
            eng.time_kernel(cls, drv, day, night, launches=2)
            ms = eng.time_kernel(cls, drv, day, night, launches=8)
            print(json.dumps({'shift': shift, 'stagger': st, 'pitch_GiB': round(per / 2**30, 4), 'ms': round(ms, 3)}), flush=True)


if __name__ == '__main__':
    main()
