"""
TEST INFRASTRUCTURE ONLY -- seeded synthetic MOD16 driver fields (numpy).

Physically consistent random drivers for parity tests and golden fixtures.
Ranges follow the 2nd-98th percentile driver bounds of the reference's
``mod16/sensitivity.py:31-46``; VPD is derived from a random relative humidity
so that both the wet (rh >= 0.7) and dry branches are populated (SURVEY.md
section 8d). The GPU bench uses its own on-device generator
(``mod16_synth_*`` in the C-ABI); this one only feeds the small CPU cases.
"""
import numpy as np

from . import mod16_oracle as oracle


def drivers(shape, seed=0, dtype=np.float64, special=True):
    """Return (cls_u8, [14 driver arrays]) of ``shape`` in the argument order
    of ``MOD16.evapotranspiration`` (reference mod16/__init__.py:675-682)."""
    rng = np.random.default_rng(seed)
    u = lambda lo, hi: rng.uniform(lo, hi, shape)
    temp_day = u(255, 305)
    temp_night = temp_day - u(0, 12)
    tmin = temp_night - u(0, 3)
    temp_annual = u(265, 300)
    rh_day = u(0.05, 1.0)
    rh_night = u(0.05, 1.0)
    vpd_day = oracle.svp(temp_day) * (1 - rh_day)
    vpd_night = oracle.svp(temp_night) * (1 - rh_night)
    sw_rad_day = u(0, 360)
    sw_rad_night = np.zeros(shape)
    lw_net_day = u(-100, 0)
    lw_net_night = u(-50, 0)
    sw_albedo = u(0.1, 0.22)
    pressure = u(70000, 101340)
    fpar = u(0.02, 0.89)
    lai = u(0.13, 5.34)
    valid = np.array(oracle.PFT_VALID, np.uint8)
    cls = valid[rng.integers(0, len(valid), shape)]
    if special:
        r = rng.uniform(0, 1, shape)
        fpar = np.where(r < 0.01, 0.0, np.where(r < 0.015, 1.0, fpar))
        r = rng.uniform(0, 1, shape)
        lai = np.where(r < 0.01, 0.0, lai)
        r = rng.uniform(0, 1, shape)
        fpar = np.where(r < 0.005, np.nan, fpar)
        r = rng.uniform(0, 1, shape)
        lai = np.where(r < 0.005, np.nan, lai)
        r = rng.uniform(0, 1, shape)
        cls = np.where(r < 0.01, 0, np.where(r < 0.02, 11, cls)).astype(np.uint8)
    drv = [lw_net_day, lw_net_night, sw_rad_day, sw_rad_night, sw_albedo,
           temp_day, temp_night, temp_annual, tmin, vpd_day, vpd_night,
           pressure, fpar, lai]
    return cls, [np.ascontiguousarray(a, dtype=dtype) for a in drv]
