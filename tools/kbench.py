#!/usr/bin/env python3
"""Kernel-only timing of the fused ET kernel for one or more builds of the
library (interleaved rounds in one process per build are not possible with
ctypes-loaded same-named symbols, so each build runs in a child process).

  python tools/kbench.py [--rows 4320] [--launches 10] [--rounds 3] lib1.so lib2.so ...
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(args):
    sys.path.insert(0, ROOT)
    import torch
    from mod16_amd import _lib
    from mod16_amd.raster import RasterEngine
    from mod16_amd.utils import restore_bplut, bplut_table
    from mod16_amd.models import COLLECTION61_BPLUT
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    math = {'exact': _lib.MATH_EXACT, 'mixed': _lib.MATH_MIXED}.get(args.math, _lib.MATH_FAST)
    eng = RasterEngine(table, dtype=args.dtype, math=math)
    n = args.rows * 43200
    if args.slab >= 0:
        # experiment: every array carved out of ONE allocation, successive
        # arrays offset by an extra `slab` bytes (physical placement control)
        esz = 8 if args.dtype == 'float64' else 4
        pad = (args.slab + 255) // 256 * 256
        per = (n * esz + 4095) // 4096 * 4096 + pad
        slab = torch.empty(16 * per + n + 4096, dtype=torch.uint8, device='cuda')
        views = [slab[k * per:k * per + n * esz].view(eng.dtype) for k in range(16)]
        drv, (day, night) = views[:14], views[14:]
        cls = slab[16 * per:16 * per + n]
        eng.synth(n, seed=16, out=(cls, drv))
    elif args.stagger:
        # experiment: offset the arrays' base addresses against each other
        pad = args.stagger * 16 // 8
        bufs = [torch.empty(n + pad, dtype=eng.dtype, device='cuda') for _ in range(16)]
        views = [b[(k * args.stagger // 8) % (pad + 1):][:n] for k, b in enumerate(bufs)]
        drv, (day, night) = views[:14], views[14:]
        cls = torch.empty(n, dtype=torch.uint8, device='cuda')
        eng.synth(n, seed=16, out=(cls, drv))
    elif args.separate:
        cls, drv = eng.synth(n, seed=16)
        day, night = eng.empty(n, 2)
    else:
        cls, drv, day, night = eng.alloc_raster(n, args.extra_mib << 20)
        eng.synth(n, seed=16, out=(cls, drv))
    if args.ab:
        # same process, same buffers: contexts created under NAME=0 and NAME=1
        # (process-to-process timing differs by several % with the physical
        # placement of the slab, which confounds a process-per-variant A/B)
        import numpy as np
        engs = {}
        vals = args.ab_values.split(',')
        for val in vals:
            os.environ[args.ab] = val
            e = RasterEngine(table, dtype=args.dtype, math=math)
            e.ctx = _lib.Context(0, experiments=True)
            e.ctx.set_bplut(np.ascontiguousarray(table, np.float64))
            engs[val] = e
        del os.environ[args.ab]
        res = {v: [] for v in vals}
        for val in vals:
            engs[val].time_kernel(cls, drv, day, night, launches=2)
        for _ in range(args.rounds):
            for val in vals:
                res[val].append(round(engs[val].time_kernel(cls, drv, day, night, launches=args.launches), 3))
        print(json.dumps({'ab': args.ab, 'ms': res, 'best': {v: min(res[v]) for v in vals}}))
        return
    eng.time_kernel(cls, drv, day, night, launches=2)
    ms = [eng.time_kernel(cls, drv, day, night, launches=args.launches) for _ in range(args.rounds)]
    best = min(ms)
    print(json.dumps({'lib': os.path.basename(_lib.LIB_PATH), 'ms': ms, 'best_ms': best,
                      'gpix_s': n / best / 1e6, 'GBps': eng.bytes_per_pixel * n / best / 1e6,
                      'frac_8TBs': eng.bytes_per_pixel * n / best / 1e6 / 8000}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=4320)
    ap.add_argument('--launches', type=int, default=10)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--dtype', default='float64')
    ap.add_argument('--math', default='fast')
    ap.add_argument('--child', action='store_true')
    ap.add_argument('--stagger', type=int, default=0, help='bytes between successive array bases (mod allocation)')
    ap.add_argument('--slab', type=int, default=-1, help='carve all arrays from one allocation with this extra offset (bytes)')
    ap.add_argument('--separate', action='store_true', help='one allocation per array instead of alloc_raster')
    ap.add_argument('--ab', default='', help='environment switch (e.g. MOD16_PITCH) to A/B inside one process')
    ap.add_argument('--ab-values', default='0,1', help='values of the --ab switch to compare')
    ap.add_argument('--extra-mib', type=int, default=0, help='extra distance between the arrays of the slab [MiB]')
    ap.add_argument('libs', nargs='*')
    args = ap.parse_args()
    if args.child:
        return child(args)
    libs = args.libs or [os.path.join(ROOT, 'mod16_amd', 'libmod16hip.so')]
    for lib in libs:
        env = dict(os.environ, MOD16_LIB=os.path.abspath(lib))
        subprocess.run([sys.executable, __file__, '--child', '--rows', str(args.rows),
                        '--launches', str(args.launches), '--rounds', str(args.rounds),
                        '--dtype', args.dtype, '--math', args.math, '--stagger', str(args.stagger), '--slab', str(args.slab)] + (['--separate'] if args.separate else []) + (['--ab', args.ab, '--ab-values', args.ab_values] if args.ab else []) + ['--extra-mib', str(args.extra_mib)], env=env, check=False)


if __name__ == '__main__':
    main()
