#!/usr/bin/env python3
"""Throughput of the calibration path: MOD16._et for D
parameter vectors over N tower-days -- one batched launch against D calls of
the single-vector interface (the numpy oracle of tests/ takes 15.6 ms per
draw of 100 k pixels on one core of the same host, measured once: profiles/).

  python tools/calbench.py [N=100000] [D=2048]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import mod16_amd  # noqa: E402
from _drivers import drivers as synth_drivers  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    ndraw = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    rng = np.random.default_rng(0)
    _, drv = synth_drivers((n,), seed=1)
    lo = np.array([-10, 5, 400, 2000, 0.01, 0.01, 1e-6, 0.001, 20, 60, 50.0])
    hi = np.array([-6, 15, 1000, 5000, 0.12, 0.12, 1e-4, 0.01, 70, 120, 800.0])
    params = rng.uniform(lo, hi, (ndraw, 11))
    M = mod16_amd.MOD16
    obs = M._et(list(params[0]), *drv) + rng.normal(0, 5, n)
    M._et_batch(params[:4], *drv, observed=obs)            # warm-up
    t = time.perf_counter()
    sse, cnt = M._et_batch(params, *drv, observed=obs)
    t_batch = time.perf_counter() - t
    M._et_batch(params[:4], *drv, observed=obs, math=mod16_amd._lib.MATH_FAST)
    t = time.perf_counter()
    sse_f, cnt_f = M._et_batch(params, *drv, observed=obs, math=mod16_amd._lib.MATH_FAST)
    t_fast = time.perf_counter() - t
    k = min(ndraw, 64)
    t = time.perf_counter()
    for d in range(k):
        e = M._et(list(params[d]), *drv)
        r = e - obs
        np.nansum(r * r)
    t_loop = (time.perf_counter() - t) / k
    print(json.dumps({
        'pixels': n, 'draws': ndraw,
        'batched_s': round(t_batch, 4), 'batched_pixel_draws_per_s': round(n * ndraw / t_batch),
        'batched_fast_s': round(t_fast, 4), 'batched_fast_pixel_draws_per_s': round(n * ndraw / t_fast),
        'fast_vs_exact_objective_max_rel_diff': float(np.max(np.abs(sse_f - sse) / sse)),
        'single_call_s_per_draw': round(t_loop, 5),
        'single_call_pixel_draws_per_s': round(n / t_loop),
        'batched_vs_single_call': round(t_loop * ndraw / t_batch, 1),
        'best_draw': int(np.argmin(sse / cnt))}))


if __name__ == '__main__':
    main()
