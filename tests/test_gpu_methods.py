"""The reference's own unit tests (tests/tests.py:64-232), restated against
mod16_amd on the GPU -- same inputs, same rounding, same expected numbers --
followed by full-precision parity of every sub-method with golden vectors made
by the reference (tests/golden/f6_submethods.npz, f1_tests_scalars.npz)."""
import numpy as np
import pytest

from oracle import mod16_oracle as oracle
from oracle import synth
from parity import assert_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def m16():
    import mod16_amd
    return mod16_amd


class Setup:
    """tests/tests.py:19-62 (setUp)"""
    params = dict(gl_sh=0.01, gl_wv=0.01, g_cuticular=1e-5, tmin_close=-8, tmin_open=8,
                  vpd_open=650, vpd_close=3000, rbl_min=60, rbl_max=90, csl=2.4e-3, beta=250)
    pressure = 100e3
    temp_k = 273.15 + 30
    tmin = 285
    vpd = 1000
    lai = 1.5
    fpar = 0.5
    rad_canopy = 5000
    rad_soil = 5000
    r_corr = (101300 / pressure) * (temp_k / 293.15)**1.75
    _pressure = np.arange(98e3, 103e3, 1e3)
    _temp_k = 273.15 + np.array([0, 10, 20, 30, 40])
    _vpd = np.arange(0, 5000, 1000)
    _lai = np.arange(0.5, 3, 0.5)
    _fpar = np.array([0.1, 0.3, 0.5, 0.7, 0.9])
    _rad_canopy = np.arange(3e3, 8e3, 1e3)


S = Setup


def test_et_instance_interface(m16):                      # tests.py:64-90
    model = m16.MOD16(S.params)
    day, night = model.evapotranspiration(
        -50, -30, 150, 0, 0.3, 293, 290, 285, 285, 1000, 500, S.pressure, S.fpar, S.lai)
    lhv_day = m16.latent_heat_vaporization(293)
    lhv_night = m16.latent_heat_vaporization(290)
    assert round(float((day * lhv_day) + (night * lhv_night)), 1) == 41.0


def test_evaporation_soil(m16):                           # tests.py:92-98
    model = m16.MOD16(S.params)
    evap = 60 * 60 * model.evaporation_soil(
        S.pressure, S.temp_k, S.vpd, S.fpar, S.rad_soil, S.r_corr)
    assert evap.round(3) == 3.102


def test_evaporation_soil_by_fpar(m16):                   # tests.py:100-111
    model = m16.MOD16(S.params)
    evap = 60 * 60 * model.evaporation_soil(
        S.pressure, S.temp_k, S.vpd, S._fpar, S.rad_soil, S.r_corr)
    assert np.equal(evap.round(3), np.array([3.128, 3.115, 3.102, 3.089, 3.076])).all()


def test_transpiration_daytime(m16):                      # tests.py:113-121
    model = m16.MOD16(S.params)
    trans = 60 * 60 * model.transpiration(
        S.pressure, S.temp_k, S.vpd, S.lai, S.fpar, S.rad_canopy, S.tmin, S.r_corr,
        daytime=True)
    assert trans.round(3) == 1.248


def test_transpiration_nighttime(m16):                    # tests.py:123-132
    model = m16.MOD16(S.params)
    trans = 60 * 60 * model.transpiration(
        S.pressure, S.temp_k, S.vpd, S.lai, S.fpar, S.rad_canopy, S.tmin, S.r_corr,
        daytime=False)
    assert trans.round(3) == 0.011


def test_wet_canopy_evaporation(m16):                     # tests.py:134-141
    model = m16.MOD16(S.params)
    evap = model.evaporation_wet_canopy(
        S.pressure, S.temp_k, S.vpd, S.lai, S.fpar, S.rad_canopy).round(6)
    assert evap == 4.49e-4


@pytest.mark.parametrize('arg,values,expected', [        # tests.py:143-201
    ('pressure', S._pressure, [1.623, 1.62, 1.618, 1.615, 1.612]),
    ('temp_k', S._temp_k, [0., 0., 0., 1.618, 3.222]),
    ('vpd', S._vpd, [5.382, 1.618, 0, 0, 0]),
    ('lai', S._lai, [1.174, 1.478, 1.618, 1.699, 1.752]),
    ('fpar', S._fpar, [1.611, 1.615, 1.618, 1.621, 1.624]),
    ('rad_canopy', S._rad_canopy, [0.974, 1.296, 1.618, 1.94, 2.262]),
])
def test_wet_canopy_evaporation_sweeps(m16, arg, values, expected):
    model = m16.MOD16(S.params)
    kw = dict(pressure=S.pressure, temp_k=S.temp_k, vpd=S.vpd, lai=S.lai, fpar=S.fpar,
              rad_canopy=S.rad_canopy)
    kw[arg] = values
    evap = 60 * 60 * model.evaporation_wet_canopy(**kw)
    assert np.equal(evap.round(3), np.array(expected)).all()


def test_psychrometric_constant(m16):                     # tests.py:203-215
    pressure = np.array((100e3, 80e3, 100e3, 80e3))
    temp_k = 273.15 + np.array((10, 10, 25, 25))
    answer = [65.74, 52.59, 66.69, 53.35]
    for i in range(0, 4):
        assert answer[i] == m16.psychrometric_constant(pressure[i], temp_k[i]).round(2)
    assert 54.55 == np.round(m16.psychrometric_constant(81.8e3, 25 + 273.15), 2)


def test_radiation_net(m16):                              # tests.py:217-226
    swrad = np.array((500, 5000, 500, 5000, 500, 5000, 500, 5000))
    albedo = np.array((0.4, 0.4, 0.8, 0.8, 0.4, 0.4, 0.8, 0.8))
    temp_k = 273.15 + np.array((10, 10, 10, 10, 25, 25, 25, 25))
    answer = [223.3, 2923.3, 23.3, 923.3, 241.8, 2941.8, 41.8, 941.8]
    for i in range(0, 8):
        assert answer[i] == m16.radiation_net(swrad[i], albedo[i], temp_k[i]).round(1)


def test_svp_slope(m16):                                  # tests.py:228-232
    assert 82.3 == m16.svp_slope(273.15 + 10).round(1)
    assert 144.8 == m16.svp_slope(273.15 + 20).round(1)
    assert 188.8 == m16.svp_slope(273.15 + 25).round(1)


def test_notebook_cell_12(m16):
    """MOD16Collection61(12).evaporation_soil(101e3, 293.15, 1000, 0.5, 100)
    = 5.083295223395212e-06 in the reference's forward-run notebook."""
    from mod16_amd.models import MOD16Collection61
    got = MOD16Collection61(12).evaporation_soil(101e3, 293.15, 1000, 0.5, 100)
    assert abs(float(got) / 5.083295223395212e-06 - 1) < 1e-13


# ---------------------------------------------------------- full precision
RTOL = 2e-13   # exp / pow differ from glibc by an ulp; everything else is IEEE-exact


def test_submethods_against_reference_vectors(m16, golden):
    f = golden('f6_submethods')
    m = m16.MOD16(dict(zip(oracle.PARAM_NAMES, f['params'])))
    (lw_d, lw_n, sw_d, sw_n, alb, t_d, t_n, t_a, tmin, vpd_d, vpd_n, pa, fpar, lai) = \
        list(f['drivers'])
    chk = lambda got, name: assert_parity(np.asarray(got), f[name], RTOL, name)
    chk(m16.svp(t_d), 'svp')
    chk(m16.svp_slope(t_d), 'svp_slope')
    chk(m16.svp_slope(t_d, f['svp']), 'svp_slope')
    chk(m16.latent_heat_vaporization(t_d), 'lhv')
    chk(m16.psychrometric_constant(pa, t_d), 'psychrometric_constant')
    chk(m16.MOD16.rhumidity(t_d, vpd_d), 'rhumidity')
    chk(m16.MOD16.air_density(t_d, pa, f['rhumidity']), 'air_density')
    g = m.soil_heat_flux(sw_d * (1 - alb) + lw_d, lw_n, t_d, t_n, t_a)
    assert isinstance(g, list)
    chk(g[0], 'soil_heat_flux_day')
    chk(g[1], 'soil_heat_flux_night')
    rs = m.radiation_soil(lw_d, lw_n, sw_d, sw_n, alb, t_d, t_n, t_a, fpar)
    assert isinstance(rs, tuple)
    chk(rs[0], 'radiation_soil_day')
    chk(rs[1], 'radiation_soil_night')
    chk(m.surface_conductance(tmin, vpd_d), 'surface_conductance')
    rad_c = f['rad_canopy']
    chk(m.evaporation_wet_canopy(pa, t_d, vpd_d, lai, fpar, rad_c), 'evaporation_wet_canopy')
    chk(m.evaporation_soil(pa, t_d, vpd_d, fpar, f['radiation_soil_day']), 'evaporation_soil')
    sat, unsat = m16.MOD16.potential_soil_evaporation(
        pa, t_d, vpd_d, fpar, f['radiation_soil_day'], vpd_open=m.vpd_open,
        vpd_close=m.vpd_close, rbl_min=m.rbl_min, rbl_max=m.rbl_max)
    chk(sat, 'potential_soil_sat')
    chk(unsat, 'potential_soil_unsat')
    chk(m.transpiration(pa, t_d, vpd_d, lai, fpar, rad_c, tmin), 'transpiration_day')
    chk(m.transpiration(pa, t_n, vpd_n, lai, fpar, fpar * lw_n, tmin, daytime=False),
        'transpiration_night')
    # optional arguments given explicitly change nothing
    lhv = f['lhv']
    rh = f['rhumidity']
    fw = np.where(rh < 0.7, 0, rh**4)
    chk(m.evaporation_soil(pa, t_d, vpd_d, fpar, f['radiation_soil_day'], None, lhv, rh, fw),
        'evaporation_soil')
    chk(m.transpiration(pa, t_d, vpd_d, lai, fpar, rad_c, tmin, None, lhv, rh, fw),
        'transpiration_day')


def test_component_known_answers(m16, golden):
    f = golden('f1_tests_scalars')
    m = m16.MOD16(dict(zip(oracle.PARAM_NAMES, f['params'])))
    got = m.evaporation_soil(S.pressure, S.temp_k, S.vpd, S.fpar, S.rad_soil, S.r_corr)
    assert np.ndim(got) == 0
    assert_parity(np.asarray(got), f['kat_evaporation_soil'], RTOL, 'soil')
    assert_parity(np.asarray(m.transpiration(
        S.pressure, S.temp_k, S.vpd, S.lai, S.fpar, S.rad_canopy, S.tmin, S.r_corr)),
        f['kat_transpiration_day'], RTOL, 'trans day')
    assert_parity(np.asarray(m.transpiration(
        S.pressure, S.temp_k, S.vpd, S.lai, S.fpar, S.rad_canopy, S.tmin, S.r_corr,
        daytime=False)), f['kat_transpiration_night'], RTOL, 'trans night')
    assert_parity(np.asarray(m.evaporation_wet_canopy(
        S.pressure, S.temp_k, S.vpd, S.lai, S.fpar, S.rad_canopy)),
        f['kat_wet_canopy'], RTOL, 'canopy')


def test_static_helpers(m16):
    """air_pressure, vpd, potential_transpiration: closed forms of the
    reference (mod16/__init__.py:414-447, :604-644, :546-602)."""
    elev = np.array([0.0, 500.0, 1500.0, 4000.0])
    want = 101325.0 * np.power(1 - (0.0065 * elev) / 288.15, m16.AIR_PRESSURE_RATE)
    assert_parity(m16.MOD16.air_pressure(elev), want, RTOL, 'air_pressure')
    qv, pa, tm = np.array([0.004, 0.012]), np.array([95e3, 101e3]), np.array([283.15, 300.0])
    want = 610.7 * np.exp((17.38 * (tm - 273.15)) / (239 + (tm - 273.15))) - \
        (qv * pa) / (0.622 + (0.379 * qv))
    assert_parity(m16.MOD16.vpd(qv, pa, tm), want, RTOL, 'vpd')
    lw, sw, alb, fpar, vpd = -60.0, 300.0, 0.15, 0.6, np.array([400.0, 1500.0])
    t = np.array([290.0, 300.0])
    rh = oracle.rhumidity(t, vpd)
    fw = oracle.wet_fraction(rh)
    s = oracle.svp_slope(t)
    want = (1.26 * (s * (fpar * (sw * (1 - alb) + lw))) * (1 - fw)) / \
        (s + oracle.psychrometric_constant(pa, t))
    assert_parity(m16.MOD16.potential_transpiration(lw, sw, alb, pa, t, vpd, fpar), want,
                  RTOL, 'pet')


def test_f9_potential_transpiration_and_radiation_net(m16, golden):
    """Against the reference's own outputs (tests/golden/f9_round2.npz)."""
    f = golden('f9_round2')
    (lw_d, lw_n, sw_d, sw_n, alb, t_d, t_n, t_a, tmin, vpd_d, vpd_n, pa, fpar, lai) = \
        list(f['drivers'])
    M = m16.MOD16
    assert_parity(M.potential_transpiration(lw_d, sw_d, alb, pa, t_d, vpd_d, fpar),
                  f['potential_transpiration'], RTOL, 'pot. transpiration')
    assert_parity(M.potential_transpiration(lw_d, sw_d, alb, pa, t_d, vpd_d, fpar, alpha=1.0),
                  f['potential_transpiration_alpha1'], RTOL, 'alpha = 1')
    assert_parity(M.potential_transpiration(lw_d, sw_d, alb, pa, t_d, vpd_d, fpar,
                                            rhumidity=f['rhumidity'], f_wet=f['f_wet']),
                  f['potential_transpiration_given'], RTOL, 'rh, f_wet given')
    assert_parity(m16.radiation_net(sw_d, alb, t_d), f['radiation_net'], RTOL, 'radiation_net')


@pytest.mark.parametrize('tag,tiny', [('1e-3', 1e-3), ('1e-12', 1e-12)])
def test_f9_tiny_is_an_argument(m16, golden, tag, tiny):
    """`tiny` as the reference takes it (mod16/__init__.py:869, :1157): a kernel
    argument, checked on inputs where it changes the result."""
    f = golden('f9_round2')
    model = m16.MOD16(dict(zip(m16.MOD16.required_parameters, f['params'])))
    (lw_d, lw_n, sw_d, sw_n, alb, t_d, t_n, t_a, tmin, vpd_d, vpd_n, pa, fpar, lai) = \
        list(f['drivers'])
    rad_c = f['rad_canopy']
    assert_parity(model.evaporation_wet_canopy(pa, t_d, vpd_d, lai, fpar, rad_c, tiny=tiny),
                  f['wet_canopy_tiny_' + tag], RTOL, 'wet canopy')
    assert_parity(model.transpiration(pa, t_d, vpd_d, lai, fpar, rad_c, tmin, tiny=tiny),
                  f['transpiration_day_tiny_' + tag], RTOL, 'transpiration day')
    assert_parity(model.transpiration(pa, t_n, vpd_n, lai, fpar, fpar * lw_n, tmin,
                                      daytime=False, tiny=tiny),
                  f['transpiration_night_tiny_' + tag], RTOL, 'transpiration night')


def test_f9_static_path_tiny(m16, golden):
    f = golden('f9_round2')
    params = [f['static_params'][k:k + 1] for k in range(11)]
    drv = list(f['static_drivers'])
    day, night = m16.MOD16._evapotranspiration(params, *drv, tiny=1e-2)
    assert_parity(day, f['static_day_tiny_1e-2'], 1e-11, 'day')
    assert_parity(night, f['static_night_tiny_1e-2'], 1e-11, 'night')
    # MOD16._et accepts `tiny` and, as the reference (:190-192), does not pass it on
    assert_parity(m16.MOD16._et(params, *drv, tiny=1e-2), f['static_et_ignores_tiny'], 1e-11, '_et')


def test_et_vectorized(m16, golden):                      # tests.py:64-90, all three interfaces
    params_vector = [S.params[p] for p in m16.MOD16.required_parameters]
    drivers = (-50, -30, 150, 0, 0.3, 293, 290, 285, 285, 1000, 500, S.pressure, S.fpar, S.lai)
    _et = m16.MOD16._et(params_vector, *drivers)
    _day, _night = m16.MOD16._evapotranspiration(params_vector, *drivers)
    assert round(float(_et), 2) == 40.94
    assert round(float(_day + _night), 2) == 40.94
    f = golden('f1_tests_scalars')
    assert_parity(np.asarray(_et), f['et_static'], RTOL, '_et')
    assert_parity(np.array([_day, _night]), f['et_static_daynight'], RTOL, '_evapotranspiration')


def test_static_calibration_path(m16, golden):
    """MOD16._evapotranspiration with (T x N) drivers and (1 x N) parameters
    against the reference's own run (f7), incl. r_corr_list and the whole-array
    any(g_surf > 0) switch."""
    f = golden('f7_static_path')
    params = [f['params'][k:k + 1] for k in range(11)]
    drv = list(f['drivers'])
    day, night = m16.MOD16._evapotranspiration(params, *drv)
    assert day.shape == (12, 40)
    assert_parity(day, f['day'], 1e-11, 'day')
    assert_parity(night, f['night'], 1e-11, 'night')
    assert_parity(m16.MOD16._et(params, *drv), f['et'], 1e-11, 'et')
    day, night = m16.MOD16._evapotranspiration(
        params, *drv, r_corr_list=[f['r_corr_day'], f['r_corr_night']])
    assert_parity(day, f['day_rcorr'], 1e-11, 'day r_corr_list')
    assert_parity(night, f['night_rcorr'], 1e-11, 'night r_corr_list')
    cold = list(drv)
    cold[8] = f['tmin_cold']
    day, night = m16.MOD16._evapotranspiration(params, *cold)
    assert_parity(day, f['day_cold'], 1e-11, 'day, no g_surf anywhere')
    assert_parity(night, f['night_cold'], 1e-11, 'night, no g_surf anywhere')


def _calibration_inputs(n, seed, dtype=np.float64):
    """Tower-day-like drivers and a spread of parameter vectors around the
    reference's Collection 6.1 values (ranges: sensitivity.py:31-46)."""
    rng = np.random.default_rng(seed)
    _, drv = synth.drivers((n,), seed=seed, dtype=dtype)
    lo = np.array([-10, 5, 400, 2000, 0.01, 0.01, 1e-6, 0.001, 20, 60, 50.0])
    hi = np.array([-6, 15, 1000, 5000, 0.12, 0.12, 1e-4, 0.01, 70, 120, 800.0])
    return drv, lo, hi, rng


def test_batched_calibration_path(m16):
    """MOD16._et for many parameter vectors in one launch;
    row d equals the single-vector interface bit for bit and the oracle's
    restatement of the reference (mod16/__init__.py:162-382) within 1e-11."""
    drv, lo, hi, rng = _calibration_inputs(5000, 41)
    params = rng.uniform(lo, hi, (37, 11))
    params[5, 7] = 0.0            # csl = 0: g_surf = 0 everywhere -> the whole-array switch is off
    params[6, 1] = 400.0          # tmin_open far above every tmin: ramp 0 everywhere, same switch
    et = m16.MOD16._et_batch(params, *drv)
    assert et.shape == (37, 5000) and et.dtype == np.float64
    day, night = m16.MOD16._et_batch(params, *drv, separate=True)
    for d in (0, 5, 6, 17, 36):
        one = m16.MOD16._et(list(params[d]), *drv)
        assert np.array_equal(et[d], one, equal_nan=True)
        d1, n1 = m16.MOD16._evapotranspiration(list(params[d]), *drv)
        assert np.array_equal(day[d], d1, equal_nan=True)
        assert np.array_equal(night[d], n1, equal_nan=True)
        assert_parity(et[d], oracle.et_static(list(params[d]), *drv), 1e-11, 'draw %d' % d)
    assert not np.array_equal(et[5], et[0])


def test_batched_calibration_objective(m16):
    """The fused residual reduction: sse / count per draw against numpy."""
    drv, lo, hi, rng = _calibration_inputs(3001, 42)
    params = rng.uniform(lo, hi, (9, 11))
    et = m16.MOD16._et_batch(params, *drv)
    obs = et[3] + rng.normal(0, 5, 3001)
    obs[::97] = np.nan                      # gaps in the tower record
    w = rng.uniform(0.5, 2.0, 3001)
    sse, cnt = m16.MOD16._et_batch(params, *drv, observed=obs, weights=w)
    r = (et - obs) * w
    ok = np.isfinite(r)
    assert np.array_equal(cnt, ok.sum(1).astype(float))
    np.testing.assert_allclose(sse, np.where(ok, r * r, 0).sum(1), rtol=1e-12)
    assert np.argmin(sse / cnt) == 3
    sse1, cnt1 = m16.MOD16._et_batch(params, *drv, observed=obs)
    r1 = et - obs
    np.testing.assert_allclose(sse1, np.where(np.isfinite(r1), r1 * r1, 0).sum(1), rtol=1e-12)
    # deterministic: same bits on a second call
    sse2, _ = m16.MOD16._et_batch(params, *drv, observed=obs, weights=w)
    assert np.array_equal(sse, sse2)


def test_batched_calibration_float32_and_shapes(m16):
    drv, lo, hi, rng = _calibration_inputs(12 * 40, 43, np.float32)
    drv = [d.reshape(12, 40) for d in drv]
    params = rng.uniform(lo, hi, (4, 11)).astype(np.float32)
    et = m16.MOD16._et_batch(params, *drv)
    assert et.shape == (4, 12, 40) and et.dtype == np.float32
    for d in range(4):
        assert np.array_equal(et[d], m16.MOD16._et(list(params[d]), *drv), equal_nan=True)
    with pytest.raises(IndexError):
        m16.MOD16._et_batch(np.zeros((3, 10)), *drv)
    assert m16.MOD16._et_batch(np.zeros((0, 11)), *drv).shape == (0, 12, 40)


def test_batched_calibration_fast_arithmetic(m16, golden):
    """MOD16_MATH_FAST on the batched calibration path: the forward run's
    strength-reduced float64 arithmetic; within 1e-9 of the reference-order rows
    (and of the oracle) with identical NaN and exact-zero masks, on random tower
    days with edge cases mixed in and on the reference's own vectors (F7)."""
    drv, lo, hi, rng = _calibration_inputs(20000, 44)
    drv[13][::50] = 0.0                       # lai = 0
    drv[12][::61] = 0.0                       # fpar = 0
    drv[12][7::61] = 1.0                      # fpar = 1
    drv[9][::70] = 0.0                        # vpd = 0 (rh = 1)
    drv[9][3::70] = -50.0                     # vpd < 0 (rh > 1: no upper clamp on this path)
    drv[10][5::70] = 9000.0                   # beyond saturation (rh clamped at 0)
    drv[5][::400] = np.nan
    drv[11][11::500] = 0.0                    # pressure = 0
    params = rng.uniform(lo, hi, (24, 11))
    params[2, 7] = 0.0                        # csl = 0: the whole-array switch off
    exact = m16.MOD16._et_batch(params, *drv, separate=True)
    fast = m16.MOD16._et_batch(params, *drv, separate=True, math=m16._lib.MATH_FAST)
    for e, f, what in zip(exact, fast, ('day', 'night')):
        assert_parity(f, e, 1e-9, 'fast vs exact, ' + what)
    tot = m16.MOD16._et_batch(params, *drv, math=m16._lib.MATH_FAST)
    for d in (0, 2, 23):
        assert_parity(tot[d], oracle.et_static(list(params[d]), *drv), 1e-9, 'oracle, draw %d' % d)
    # the reference's own vectors (F7: 12 days x 40 sites, one parameter vector per site):
    # draw j = site j's parameters over all pixels, of which site j's column is the reference's
    f7 = golden('f7_static_path')
    got = m16.MOD16._et_batch(f7['params'].T, *list(f7['drivers']), math=m16._lib.MATH_FAST)
    assert got.shape == (40, 12, 40)
    mine = np.stack([got[j, :, j] for j in range(40)], axis=1)
    assert_parity(mine, f7['et'], 1e-9, 'F7')
    # the fused objective takes the same switch
    obs = tot[5] + rng.normal(0, 3, 20000)
    s_e, c_e = m16.MOD16._et_batch(params, *drv, observed=obs)
    s_f, c_f = m16.MOD16._et_batch(params, *drv, observed=obs, math=m16._lib.MATH_FAST)
    assert np.array_equal(c_e, c_f)
    np.testing.assert_allclose(s_f, s_e, rtol=1e-8)


def test_batched_calibration_fast_float32(m16):
    """float32 tower-day arrays on the FAST batched path: float64 arithmetic on the
    widened inputs, rounded once."""
    drv, lo, hi, rng = _calibration_inputs(4000, 45, np.float32)
    params = rng.uniform(lo, hi, (7, 11)).astype(np.float32)
    got = m16.MOD16._et_batch(params, *drv, separate=True, math=m16._lib.MATH_FAST)
    want = m16.MOD16._et_batch(params.astype(np.float64), *[d.astype(np.float64) for d in drv],
                               separate=True, math=m16._lib.MATH_FAST)
    for a, b in zip(got, want):
        assert a.dtype == np.float32 and b.dtype == np.float64
        assert np.array_equal(a, b.astype(np.float32), equal_nan=True)


def test_bound_calibration_problem(m16, golden):
    """MOD16._et_bind (mod16_static_batch_bind_*): the calibration problem resident on the GPU --
    drivers up once, per evaluation parameters up and (sse, count) down around one graph launch.
    Rows are the unbound call's bit for bit (FAST and reference order); the fused objective equals
    the unbound one (counts exactly, sums to 1e-12: another, also fixed, order of addition) and the
    oracle's; the reference's whole-array switch any(g_surf > 0) is resolved inside the launch
    (csl = 0, a tmin_open above every tmin); repeated evaluations give the same bits; the number
    of draws may change between evaluations."""
    L = m16._lib
    n = 20000
    drv, lo, hi, rng = _calibration_inputs(n, 46)
    drv[13][::50] = 0.0
    drv[12][::61] = 0.0
    drv[9][3::70] = -50.0
    drv[10][5::70] = 9000.0
    drv[5][::400] = np.nan
    params = rng.uniform(lo, hi, (70, 11))
    params[2, 7] = 0.0                        # csl = 0: no g_surf anywhere -> no transpiration for this draw
    params[33, 1] = 400.0                     # tmin_open far above every tmin: the same
    params[69, 7] = 0.0                       # ... also in the ragged last chunk of 32
    unbound = m16.MOD16._et_batch(params, *drv, math=L.MATH_FAST)
    obs = unbound[5] + rng.normal(0, 3, n)
    obs[::97] = np.nan
    w = rng.uniform(0.5, 2.0, n)
    for math in (L.MATH_FAST, L.MATH_EXACT):
        prob = m16.MOD16._et_bind(*drv, observed=obs, weights=w, max_draws=128, math=math)
        assert prob.n == n and prob.n_outside_domain == 0
        rows = prob.rows(params)
        want = unbound if math == L.MATH_FAST else m16.MOD16._et_batch(params, *drv)
        assert np.array_equal(rows, want, equal_nan=True)
        day, night = prob.rows(params[:5], separate=True)
        d0, n0 = m16.MOD16._et_batch(params[:5], *drv, separate=True, math=math)
        assert np.array_equal(day, d0, equal_nan=True) and np.array_equal(night, n0, equal_nan=True)
        sse, cnt = prob.objective(params)
        s0, c0 = m16.MOD16._et_batch(params, *drv, observed=obs, weights=w, math=math)
        assert np.array_equal(cnt, c0)
        if math == L.MATH_EXACT:
            assert np.array_equal(sse, s0)                # the same kernels
        else:
            np.testing.assert_allclose(sse, s0, rtol=1e-12)
        r = (rows - obs) * w
        ok = np.isfinite(r)
        assert np.array_equal(cnt, ok.sum(1).astype(float))
        np.testing.assert_allclose(sse, np.where(ok, r * r, 0).sum(1), rtol=1e-12)
        assert np.argmin(sse / cnt) == 5
        # the switched-off draws really have no transpiration in them
        for d in (2, 33, 69):
            with np.errstate(all='ignore'):
                o = (oracle.et_static(list(params[d]), *drv) - obs) * w
            np.testing.assert_allclose(sse[d], np.where(np.isfinite(o), o * o, 0).sum(), rtol=1e-8)
        # the same bits again, and with another number of draws in between
        s_few, c_few = prob.objective(params[:33])
        assert np.array_equal(s_few, sse[:33]) and np.array_equal(c_few, cnt[:33])
        s2, c2 = prob.objective(params)
        assert np.array_equal(s2, sse) and np.array_equal(c2, cnt)
        with pytest.raises(ValueError):
            prob.objective(np.zeros((129, 11)))
        with pytest.raises(IndexError):
            prob.objective(np.zeros((3, 10)))
        if math == L.MATH_FAST:
            assert 0 < prob.gpu_milliseconds(3) < 100
        prob.close()
    # without observations: rows only
    prob = m16.MOD16._et_bind(*drv, max_draws=8)
    assert np.array_equal(prob.rows(params[:8]), unbound[:8], equal_nan=True)
    with pytest.raises(ValueError):
        prob.objective(params[:8])
    # the reference's own vectors (F7) through the bound problem
    f7 = golden('f7_static_path')
    prob = m16.MOD16._et_bind(*list(f7['drivers']), max_draws=64)
    got = prob.rows(f7['params'].T)
    assert got.shape == (40, 12, 40)
    assert_parity(np.stack([got[j, :, j] for j in range(40)], axis=1), f7['et'], 1e-9, 'F7')
    # float32 tower-day arrays
    d32 = [d.astype(np.float32) for d in drv]
    p32 = params[:9].astype(np.float32)
    prob = m16.MOD16._et_bind(*d32, observed=obs.astype(np.float32), max_draws=16)
    assert prob.dtype == np.float32
    assert np.array_equal(prob.rows(p32), m16.MOD16._et_batch(p32, *d32, math=L.MATH_FAST), equal_nan=True)
    s32, c32 = prob.objective(p32)
    s0, c0 = m16.MOD16._et_batch(p32, *d32, observed=obs.astype(np.float32), math=L.MATH_FAST)
    assert np.array_equal(c32, c0)
    np.testing.assert_allclose(s32, s0, rtol=1e-6)       # (the unbound call reduces rows rounded to float32)


def test_bound_calibration_problem_at_size(m16):
    """A million pixels (3907 pixel blocks behind every draw's sums) and a ragged number of draws:
    the fused objective against the rows it never materialises, reduced with numpy."""
    n = 1_000_003
    drv, lo, hi, rng = _calibration_inputs(n, 48)
    params = rng.uniform(lo, hi, (45, 11))
    params[44, 7] = 0.0
    obs, w = rng.normal(30, 10, n), rng.uniform(0.5, 2, n)
    prob = m16.MOD16._et_bind(*drv, observed=obs, weights=w, max_draws=64)
    sse, cnt = prob.objective(params)
    rows = prob.rows(params)
    assert rows.shape == (45, n)
    assert np.all(cnt > 0.9 * n) and np.all(np.isfinite(sse))
    # counts = finite rows (the observations and weights are finite)
    r = (rows - obs) * w
    ok = np.isfinite(r)
    assert np.array_equal(cnt, ok.sum(1).astype(float))
    np.testing.assert_allclose(sse, np.where(ok, r * r, 0).sum(1), rtol=1e-11)
    with pytest.raises(m16._lib.Mod16Error):
        m16.MOD16._et_bind(*drv, max_draws=65535 * 32 + 1)


SPECIAL = [np.inf, -np.inf, 1e300, -1e300, 3.4e38, -9999.0, 65535.0, 1e15, 35.85, 1400.0, 0.0, -1.0, 1e-300]


def test_calibration_fast_arithmetic_returns_the_reference_on_fill_values(m16):
    """The FAST arithmetic of the calibration path (unbound rows, bound rows, bound objective) on
    drivers with fill values, infinities, a temperature on the pole of the saturation formula, a zero
    or negative pressure ...: the pixels outside the arithmetic's domain are computed in the
    reference's operation order (mod16_physics.hpp "domain guard"; marked once at bind time), so
    FAST returns what the reference-order kernels and the oracle return -- NaN / zero / inf masks and
    values; the objective counts and sums them as the reference-order objective does."""
    L = m16._lib
    n = 6000
    drv, lo, hi, rng = _calibration_inputs(n, 47)
    drv = [d.copy() for d in drv]
    idx = np.arange(5, n, 7)
    which = rng.integers(0, 14, idx.size)
    val = np.array(SPECIAL)[rng.integers(0, len(SPECIAL), idx.size)]
    for k in range(14):
        drv[k][idx[which == k]] = val[which == k]
    params = rng.uniform(lo, hi, (40, 11))
    params[7, 7] = 0.0
    with np.errstate(all='ignore'):
        exact = m16.MOD16._et_batch(params, *drv)
        fast = m16.MOD16._et_batch(params, *drv, math=L.MATH_FAST)
    # (1e-8 here: in-domain fill values such as a VPD of 65535 Pa or 1e15 sit far outside the tower-day
    # ranges on which the 1e-9 of the FAST arithmetic is stated; worst seen 1.5e-9)
    assert_parity(fast, exact, 1e-8, 'unbound FAST vs reference order')
    with np.errstate(all='ignore'):
        for d in (0, 7, 39):
            assert_parity(fast[d], oracle.et_static(list(params[d]), *drv), 1e-8, 'oracle, draw %d' % d)
    obs = np.nan_to_num(exact[3], nan=10.0, posinf=10.0, neginf=10.0) + rng.normal(0, 3, n)
    prob = m16.MOD16._et_bind(*drv, observed=obs, max_draws=64)
    assert prob.n_outside_domain > 100
    assert np.array_equal(prob.rows(params), fast, equal_nan=True)
    sse, cnt = prob.objective(params)
    with np.errstate(all='ignore'):
        s_e, c_e = m16.MOD16._et_batch(params, *drv, observed=obs)
    assert np.array_equal(cnt, c_e)
    ok = np.isfinite(s_e)
    np.testing.assert_allclose(sse[ok], s_e[ok], rtol=1e-8)
    assert np.array_equal(np.isinf(sse), np.isinf(s_e)) and np.array_equal(np.isnan(sse), np.isnan(s_e))
    # a problem whose ONLY pixels with g_surf > 0 lie outside the domain: the whole-array switch sees them
    cold = [d.copy() for d in drv]
    cold[8][:] = 200.0                        # tmin far below every tmin_close: the ramp is 0 ...
    cold[8][idx[which == 0]] = 290.0          # ... except where lw_net_day holds a special value
    flagged = idx[which == 0][np.isinf(drv[0][idx[which == 0]])]
    assert flagged.size
    with np.errstate(all='ignore'):
        e2 = m16.MOD16._et_batch(params[:8], *cold)
        f2 = m16.MOD16._et_batch(params[:8], *cold, math=L.MATH_FAST)
    assert_parity(f2, e2, 1e-8, 'switch carried by flagged pixels')
    p2 = m16.MOD16._et_bind(*cold, observed=obs, max_draws=8)
    s2, c2 = p2.objective(params[:8])
    with np.errstate(all='ignore'):
        s_e2, c_e2 = m16.MOD16._et_batch(params[:8], *cold, observed=obs)
    assert np.array_equal(c2, c_e2)
    ok = np.isfinite(s_e2)
    np.testing.assert_allclose(s2[ok], s_e2[ok], rtol=1e-8)
