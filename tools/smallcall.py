#!/usr/bin/env python3
"""BASELINE.json configs[0]: how long ONE call of the drop-in takes on small inputs -- the
flux-tower scalars of the reference's tests/tests.py, a year of one site (365 values), small
windows. Wall clock of the Python call, best and median of 200 (bench.py's `c1_single_site` leg
holds the same calls against the numpy oracle, value and time); `split`: where the scalar call's
time goes. MOD16_SMALL_PIXELS=0 in the environment: every call through the staged path."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import mod16_amd  # noqa: E402

# the reference's flux-tower scalars (tests/tests.py:21-54), as the golden fixture F1 holds them
_F1 = np.load(os.path.join(ROOT, 'tests', 'golden', 'f1_tests_scalars.npz'))
PARAMS = dict(zip(mod16_amd.MOD16.required_parameters, (float(v) for v in _F1['params'])))
SITE = [float(v) for v in _F1['drivers']]


def timed(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[0], ts[len(ts) // 2]


def main():
    reps = 200
    rng = np.random.default_rng(16)
    model = mod16_amd.MOD16(PARAMS)
    for name, shape in (('scalars', ()), ('one_site_year', (365,)), ('window_32x32', (32, 32)), ('window_64x64', (64, 64)),
                        ('window_100x100', (100, 100)), ('window_128x128', (128, 128)), ('window_200x200', (200, 200)),
                        ('window_256x256', (256, 256)), ('window_300x300', (300, 300)), ('window_400x400', (400, 400)),
                        ('window_512x512', (512, 512))):
        drv = [np.asarray(v, np.float64) * (1 + 0.01 * rng.uniform(-1, 1, shape)) if shape else float(v)
               for v in SITE]
        model.evapotranspiration(*drv)
        best, med = timed(lambda: model.evapotranspiration(*drv), reps)
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        print(json.dumps({'case': name, 'pixels': n, 'gpu_call_us_best': round(best * 1e6, 1),
                          'gpu_call_us_median': round(med * 1e6, 1),
                          'small_pixels': os.environ.get('MOD16_SMALL_PIXELS', 'default')}), flush=True)


def split():
    """Where the scalar call's time goes: the C entry point alone (addresses marshalled once) against the
    whole Python call, and a cProfile listing of the latter."""
    import cProfile
    import pstats
    from mod16_amd import _lib
    model = mod16_amd.MOD16(PARAMS)
    drv = [float(v) for v in SITE]
    model.evapotranspiration(*drv)
    ctx = _lib.context(0)
    dt = np.dtype(np.float64)
    keep_d = [np.full(1, v) for v in drv]
    keep_p = [np.full(1, PARAMS[k]) for k in mod16_amd.MOD16.required_parameters]
    out = [_lib.pinned.empty((), dt) for _ in range(2)]
    args = (dt, None, [a.ctypes.data for a in keep_d], [0] * 14, [a.ctypes.data for a in keep_p], [0] * 11, 1,
            out[0].ctypes.data, out[1].ctypes.data, None)
    ctx.et(*args)
    best, med = timed(lambda: ctx.et(*args), 500)
    print(json.dumps({'what': 'Context.et alone (ctypes arrays built per call)', 'us_best': round(best * 1e6, 1),
                      'us_median': round(med * 1e6, 1)}))
    fn = ctx.lib.mod16_et_f64
    cargs = (ctx.handle, None, _lib.ptr_array(args[2]), _lib.i64_array(args[3]), _lib.ptr_array(args[4]),
             _lib.i64_array(args[5]), 1, args[7], args[8], None, int(_lib.MATH_FAST), int(_lib.HOST), None)
    best, med = timed(lambda: fn(*cargs), 500)
    print(json.dumps({'what': 'mod16_et_f64 alone (HOST mode, one pixel)', 'us_best': round(best * 1e6, 1),
                      'us_median': round(med * 1e6, 1)}))
    best, med = timed(lambda: model.evapotranspiration(*drv), 500)
    print(json.dumps({'what': 'MOD16.evapotranspiration(scalars)', 'us_best': round(best * 1e6, 1),
                      'us_median': round(med * 1e6, 1)}))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2000):
        model.evapotranspiration(*drv)
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(14)


def static():
    """MOD16._et (the calibration interface a sampler calls once per draw): a year of 1, 10, 30 and 120 sites."""
    rng = np.random.default_rng(16)
    par = [PARAMS[k] for k in mod16_amd.MOD16.required_parameters]
    for sites in (1, 10, 30, 120, 400):
        shape = (365, sites)
        drv = [np.asarray(v, np.float64) * (1 + 0.01 * rng.uniform(-1, 1, shape)) for v in SITE]
        mod16_amd.MOD16._et(par, *drv)
        best, med = timed(lambda: mod16_amd.MOD16._et(par, *drv), 100)
        print(json.dumps({'case': '_et, 365 days x %d sites' % sites, 'pixels': 365 * sites, 'gpu_call_us_best': round(best * 1e6, 1),
                          'gpu_call_us_median': round(med * 1e6, 1),
                          'small_pixels': os.environ.get('MOD16_SMALL_PIXELS', 'default')}), flush=True)


if __name__ == '__main__':
    {'split': split, 'static': static}.get(sys.argv[1] if len(sys.argv) > 1 else '', main)()
