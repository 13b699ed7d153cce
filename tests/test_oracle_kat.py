"""Known-answer tests of the reference (its tests/tests.py:64-232) restated
against the oracle: same inputs (tests.py:19-62), same rounding, same expected
numbers. These pin the oracle independently of the golden .npz files. The
static calibration path (`_et`, tests.py:66-87 first two asserts) is out of
scope; its instance-path assertion is kept."""
import numpy as np

from oracle import mod16_oracle as oracle

P = dict(gl_sh=0.01, gl_wv=0.01, g_cuticular=1e-5, tmin_close=-8, tmin_open=8,
         vpd_open=650, vpd_close=3000, rbl_min=60, rbl_max=90, csl=2.4e-3,
         beta=250)
PRESSURE = 100e3
TEMP_K = 273.15 + 30
TMIN = 285                      # tests.py:36 is overwritten by :52
VPD, LAI, FPAR = 1000, 1.5, 0.5
RAD_CANOPY = RAD_SOIL = 5000
R_CORR = (101300 / PRESSURE) * (TEMP_K / 293.15)**1.75
_PRESSURE = np.arange(98e3, 103e3, 1e3)
_TEMP_K = 273.15 + np.array([0, 10, 20, 30, 40])
_VPD = np.arange(0, 5000, 1000)
_LAI = np.arange(0.5, 3, 0.5)
_FPAR = np.array([0.1, 0.3, 0.5, 0.7, 0.9])
_RAD_CANOPY = np.arange(3e3, 8e3, 1e3)


def test_et_instance_interface():          # tests.py:64-90
    day, night = oracle.evapotranspiration(
        P, -50, -30, 150, 0, 0.3, 293, 290, 285, 285, 1000, 500, PRESSURE,
        FPAR, LAI)
    lhv_day = oracle.latent_heat_vaporization(293)
    lhv_night = oracle.latent_heat_vaporization(290)
    assert round(float(day * lhv_day + night * lhv_night), 1) == 41.0


def test_evaporation_soil():               # tests.py:92-98
    evap = 3600 * oracle.evaporation_soil(
        P, PRESSURE, TEMP_K, VPD, FPAR, RAD_SOIL, R_CORR)
    assert evap.round(3) == 3.102


def test_evaporation_soil_by_fpar():       # tests.py:100-111
    evap = 3600 * oracle.evaporation_soil(
        P, PRESSURE, TEMP_K, VPD, _FPAR, RAD_SOIL, R_CORR)
    assert np.array_equal(evap.round(3), [3.128, 3.115, 3.102, 3.089, 3.076])


def test_transpiration_daytime():          # tests.py:113-121
    t = 3600 * oracle.transpiration(
        P, PRESSURE, TEMP_K, VPD, LAI, FPAR, RAD_CANOPY, TMIN, R_CORR,
        daytime=True)
    assert t.round(3) == 1.248


def test_transpiration_nighttime():        # tests.py:123-132
    t = 3600 * oracle.transpiration(
        P, PRESSURE, TEMP_K, VPD, LAI, FPAR, RAD_CANOPY, TMIN, R_CORR,
        daytime=False)
    assert t.round(3) == 0.011


def wet(pressure=PRESSURE, temp_k=TEMP_K, vpd=VPD, lai=LAI, fpar=FPAR,
        rad=RAD_CANOPY):
    return oracle.evaporation_wet_canopy(P, pressure, temp_k, vpd, lai, fpar, rad)


def test_wet_canopy_evaporation():         # tests.py:134-141
    assert wet().round(6) == 4.49e-4


def test_wet_canopy_sweeps():              # tests.py:143-201
    eq = lambda a, b: np.array_equal((3600 * a).round(3), np.array(b))
    assert eq(wet(pressure=_PRESSURE), [1.623, 1.62, 1.618, 1.615, 1.612])
    assert eq(wet(temp_k=_TEMP_K), [0., 0., 0., 1.618, 3.222])
    assert eq(wet(vpd=_VPD), [5.382, 1.618, 0, 0, 0])
    assert eq(wet(lai=_LAI), [1.174, 1.478, 1.618, 1.699, 1.752])
    assert eq(wet(fpar=_FPAR), [1.611, 1.615, 1.618, 1.621, 1.624])
    assert eq(wet(rad=_RAD_CANOPY), [0.974, 1.296, 1.618, 1.94, 2.262])


def test_psychrometric_constant():         # tests.py:203-215
    pressure = np.array((100e3, 80e3, 100e3, 80e3))
    temp_k = 273.15 + np.array((10, 10, 25, 25))
    got = oracle.psychrometric_constant(pressure, temp_k).round(2)
    assert list(got) == [65.74, 52.59, 66.69, 53.35]
    assert np.round(oracle.psychrometric_constant(81.8e3, 25 + 273.15), 2) == 54.55


def test_svp_slope():                      # tests.py:228-232
    assert oracle.svp_slope(273.15 + 10).round(1) == 82.3
    assert oracle.svp_slope(273.15 + 20).round(1) == 144.8
    assert oracle.svp_slope(273.15 + 25).round(1) == 188.8


def test_notebook_known_answer():
    """Forward-run notebook cell 12: MOD16Collection61(12).evaporation_soil(
    101e3, 293.15, 1000, 0.5, 100) = 5.083295223395212e-06 (independent of the
    mod17 ramps)."""
    p = dict(tmin_close=-8, tmin_open=12.02, vpd_open=650, vpd_close=4500,
             gl_sh=0.02, gl_wv=0.02, g_cuticular=1e-5, csl=0.0055, rbl_min=60,
             rbl_max=95, beta=250)
    assert float(oracle.evaporation_soil(p, 101e3, 293.15, 1000, 0.5, 100)) \
        == 5.083295223395212e-06


def test_linear_constraint_ramps():
    """Pins for the restated mod17 ramps: the prescribed reductions in the
    reference's tests/verification/transpiration.c:154-155 for the
    verify.py:54-56 inputs with PFT-7 parameters."""
    up = oracle.linear_constraint(-8, 8.8)
    down = oracle.linear_constraint(650, 4400, 'reversed')
    tmin = np.array((278.92, 284.43, 289.88)) - 273.15
    vpd = np.array((710.9, 1249.4, 1979.))
    np.testing.assert_allclose(up(tmin), [0.8196, 1, 1], atol=5e-5)
    np.testing.assert_allclose(down(vpd), [0.98376, 0.84016, 0.6456], atol=5e-6)
    assert up(-8.0) == 0 and up(8.8) == 1 and up(-9) == 0
    assert down(650.) == 1 and down(4400.) == 0 and down(5000.) == 0
    assert np.isnan(up(np.nan)) and np.isnan(down(np.nan))
