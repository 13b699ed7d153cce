#!/usr/bin/env python3
"""The series step with and without RasterEngine._gate() (a zero-work dispatch on the compute stream
between the cross-stream event wait and the pipeline kernel): ms per step of run_series_tiled on the
global float64 grid, the two kernels alone, and -- run under `rocprofv3 --kernel-trace` -- the trace
from which tools/gate_trace.py reads how the generator and the pipeline kernel overlap in either case.

  python tools/gate_probe.py [steps=12] [rows=21600]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mod16_amd.raster import RasterEngine  # noqa: E402
from mod16_amd.utils import restore_bplut, bplut_table  # noqa: E402
from mod16_amd.models import COLLECTION61_BPLUT  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 21600
    n = rows * 43200
    eng = RasterEngine(bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250))
    ring = [eng.alloc_tiled(n), eng.alloc_tiled(n)]
    eng.run_series_tiled(n, 2, ring=ring)
    torch.cuda.synchronize()
    diag = torch.zeros(8, dtype=torch.float64, device='cuda')
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    for k in range(3):
        eng.synth_tiled(ring[0], step=k)
    ev[1].record()
    for k in range(3):
        eng.run_tiled(ring[0], diag=diag)
    ev[2].record()
    torch.cuda.synchronize()
    out = {'pixels': n, 'steps': steps, 'generator_kernel_ms': ev[0].elapsed_time(ev[1]) / 3, 'et_kernel_ms': ev[1].elapsed_time(ev[2]) / 3}
    gate = RasterEngine._gate
    for name, fn in (('with_gate', gate), ('without_gate', lambda self: None), ('with_gate_again', gate)):
        RasterEngine._gate = fn
        # a marker the trace can be cut at: one tiny dispatch of a distinctive size
        torch.zeros(977 + len(name), device='cuda')
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run_series_tiled(n, steps, ring=ring)
        torch.cuda.synchronize()
        out[name + '_ms_per_step'] = 1e3 * (time.perf_counter() - t0) / steps
    RasterEngine._gate = gate
    out['sum_of_kernels_ms'] = out['generator_kernel_ms'] + out['et_kernel_ms']
    print(json.dumps(out))


if __name__ == '__main__':
    main()
