"""Seeded numpy driver fields for the timing tools (ranges of the reference's
sensitivity.py:31-46; VPD from a random relative humidity through the
reference's svp formula, mod16/__init__.py:1340-1367). Timing input only -- the
parity tests use their own generator next to the oracle."""
import numpy as np


def drivers(shape, seed=0, dtype=np.float64):
    rng = np.random.default_rng(seed)
    u = lambda lo, hi: rng.uniform(lo, hi, shape)
    svp = lambda t: 610.8 * np.exp(17.27 * (t - 273.15) / (t - 273.15 + 237.3))
    t_d = u(255, 305)
    t_n = t_d - u(0, 12)
    tmin = t_n - u(0, 3)
    drv = [u(-100, 0), u(-50, 0), u(0, 360), np.zeros(shape), u(0.1, 0.22), t_d, t_n, u(265, 300), tmin,
           svp(t_d) * (1 - u(0.05, 1.0)), svp(t_n) * (1 - u(0.05, 1.0)), u(70000, 101340),
           u(0.02, 0.89), u(0.13, 5.34)]
    cls = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12], np.uint8)[rng.integers(0, 11, shape)]
    return cls, [np.ascontiguousarray(a, dtype) for a in drv]
