#!/usr/bin/env python3
"""Does the physical placement of the raster slab change the kernel time?
One process; the slab is allocated repeatedly behind paddings of different
sizes (which shifts where in HBM it lands) and the production kernel is timed
on each placement.

  python tools/placement.py [rows=21600] [pad_gb ...]
"""
import gc
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
    pads = [float(x) for x in sys.argv[2:]] or [0, 0, 40, 80, 120, 0]
    n = rows * 43200
    eng = RasterEngine(bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250))
    free0, total = torch.cuda.mem_get_info()
    print(json.dumps({'free_GB': free0 / 1e9, 'total_GB': total / 1e9}), flush=True)
    for pad_gb in pads:
        pad = torch.empty(int(pad_gb * 1e9), dtype=torch.uint8, device='cuda') if pad_gb else None
        cls, drv, day, night = eng.alloc_raster(n)
        eng.synth(n, seed=16, out=(cls, drv))
        eng.time_kernel(cls, drv, day, night, launches=2)
        ms = [round(eng.time_kernel(cls, drv, day, night, launches=10), 3) for _ in range(2)]
        print(json.dumps({'pad_GB': pad_gb, 'slab_va': hex(drv[0].data_ptr()), 'ms': ms}), flush=True)
        del cls, drv, day, night, pad
        gc.collect()
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
