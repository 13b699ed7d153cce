#!/usr/bin/env python3
"""The resident calibration problem (MOD16._et_bind) evaluated back to back: the target of a
rocprofv3 --kernel-trace --stats pass (per-kernel times of one objective evaluation) and a
quick rate check.  python tools/calres.py [pixels=100000] [draws=2048] [evaluations=20]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import mod16_amd  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _drivers import drivers  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    ndraw = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    rng = np.random.default_rng(0)
    _, drv = drivers((n,), seed=1)
    lo = np.array([-10, 5, 400, 2000, 0.01, 0.01, 1e-6, 0.001, 20, 60, 50.0])
    hi = np.array([-6, 15, 1000, 5000, 0.12, 0.12, 1e-4, 0.01, 70, 120, 800.0])
    params = rng.uniform(lo, hi, (ndraw, 11))
    obs = rng.normal(30, 10, n)
    prob = mod16_amd.MOD16._et_bind(*drv, observed=obs, max_draws=ndraw)
    prob.objective(params)
    t0 = time.perf_counter()
    for _ in range(reps):
        sse, cnt = prob.objective(params)
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({'pixels': n, 'draws': ndraw, 'seconds_per_evaluation': dt, 'pixel_draws_per_s': n * ndraw / dt,
                      'gpu_ms': prob.gpu_milliseconds(10), 'rmsd_first_draw': float(np.sqrt(sse[0] / cnt[0]))}))


if __name__ == '__main__':
    main()
