#!/usr/bin/env python3
"""Shader clock and package power while the batched calibration path runs:
2048 draws x 100 k pixels per call, reference-order and FAST arithmetic, calls back to back for
~2 s each (bench.py's sensor reader)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import mod16_amd  # noqa: E402
from mod16_amd import _lib  # noqa: E402
from _drivers import drivers  # noqa: E402


def main():
    n, ndraw = 100000, 2048
    rng = np.random.default_rng(0)
    _, drv = drivers((n,), seed=1)
    lo = np.array([-10, 5, 400, 2000, 0.01, 0.01, 1e-6, 0.001, 20, 60, 50.0])
    hi = np.array([-6, 15, 1000, 5000, 0.12, 0.12, 1e-4, 0.01, 70, 120, 800.0])
    params = rng.uniform(lo, hi, (ndraw, 11))
    obs = rng.normal(30, 5, n)
    M = mod16_amd.MOD16
    torch.cuda.init()
    for name, math, per_draw in (('reference_order', _lib.MATH_EXACT, bench.N2_INSTR['reference_order']),
                                 ('fast', _lib.MATH_FAST, bench.N2_INSTR['fast'])):
        M._et_batch(params[:4], *drv, observed=obs, math=math)
        t0 = time.perf_counter()
        M._et_batch(params, *drv, observed=obs, math=math)
        one = time.perf_counter() - t0
        calls = max(5, int(2.0 / one))
        count = [0]

        def step():
            M._et_batch(params, *drv, observed=obs, math=math)
            count[0] += 1
        t0 = time.perf_counter()
        u = bench.device_under_load(torch, step, calls, True)
        dt = (time.perf_counter() - t0) / calls
        rate = n * ndraw / dt
        line = {'arithmetic': name, 'seconds_per_call': round(dt, 5), 'pixel_draws_per_s': rate,
                'frac_of_issue_peak_at_2.4GHz': rate * per_draw / (256 * 4 * 16 * 2.4e9)}
        if u:
            line.update(sclk_mhz=round(u['sclk_mhz']), power_w=round(u['power_w'], 1),
                        frac_of_issue_peak_at_this_clock=rate * per_draw / (256 * 4 * 16 * u['sclk_mhz'] * 1e6))
        print(json.dumps(line), flush=True)


if __name__ == '__main__':
    main()
