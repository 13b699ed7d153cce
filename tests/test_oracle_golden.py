"""The oracle (oracle/mod16_oracle.py) against golden vectors produced by the
reference itself (tests/golden/make_golden.py). Bit-exact: the oracle keeps the
reference's operation order, so equality is required, not closeness."""
import numpy as np
import pytest

from oracle import mod16_oracle as oracle

SEP = ('canopy_day', 'soil_day', 'trans_day',
       'canopy_night', 'soil_night', 'trans_night')


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
    assert np.array_equal(a, b, equal_nan=True), np.nanmax(np.abs(a - b))


def params_of(vec):
    return dict(zip(oracle.PARAM_NAMES, vec))


def check_sep(res, f):
    flat = list(res[0]) + list(res[1])
    for name, got in zip(SEP, flat):
        same(got, f[name])


def test_f1_scalar_set(golden):
    f = golden('f1_tests_scalars')
    p = params_of(f['params'])
    drv = list(f['drivers'])
    day, night = oracle.evapotranspiration(p, *drv)
    same(day, f['day'])
    same(night, f['night'])
    check_sep(oracle.evapotranspiration(p, *drv, separate=True), f)
    # full-precision known answers recorded in SURVEY.md section 8c
    assert float(day) == 9.872769855543006e-06
    assert float(night) == 6.802468709788471e-06


def test_f1_component_known_answers(golden):
    f = golden('f1_tests_scalars')
    p = params_of(f['params'])
    temp_k, vpd, lai, fpar = 273.15 + 30, 1000, 1.5, 0.5
    pressure, tmin, rad = 100e3, 285, 5000
    r_corr = (101300 / pressure) * (temp_k / 293.15)**1.75
    same(oracle.evaporation_soil(p, pressure, temp_k, vpd, fpar, rad, r_corr),
         f['kat_evaporation_soil'])
    same(oracle.transpiration(p, pressure, temp_k, vpd, lai, fpar, rad, tmin,
                              r_corr, daytime=True), f['kat_transpiration_day'])
    same(oracle.transpiration(p, pressure, temp_k, vpd, lai, fpar, rad, tmin,
                              r_corr, daytime=False), f['kat_transpiration_night'])
    same(oracle.evaporation_wet_canopy(p, pressure, temp_k, vpd, lai, fpar, rad),
         f['kat_wet_canopy'])


def test_f2_verify_three_pixels(golden):
    f = golden('f2_verify_3pixel')
    p = params_of(f['params'])
    drv = [f['drv_' + k] for k in oracle.DRIVER_NAMES]
    check_sep(oracle.evapotranspiration(p, *drv, separate=True), f)
    # SURVEY.md section 8c, F2 day transpiration
    np.testing.assert_array_equal(
        f['trans_day'],
        [8.5064801746365699e-06, 1.4835055809704346e-05, 1.8276591240061832e-05])


@pytest.mark.parametrize('name,dtype', [('f3_random64_f64', np.float64),
                                        ('f5_random64_f32', np.float32)])
def test_random_multiclass_raster(golden, name, dtype):
    f = golden(name)
    bplut = {k: f['table'][:, j].astype(dtype)
             for j, k in enumerate(oracle.PARAM_NAMES)}
    drv = list(f['drivers'])
    assert drv[0].dtype == dtype
    day, night = oracle.evapotranspiration_raster(bplut, f['cls'], *drv)
    same(day, f['day'])
    same(night, f['night'])
    check_sep(oracle.evapotranspiration_raster(
        bplut, f['cls'], *drv, separate=True), f)
    # the fixture exercises every branch the raster path has
    assert np.isnan(f['day']).any() and (f['canopy_day'] > 0).any()
    assert (f['canopy_day'] == 0).any() and (f['trans_day'] == 0).any()


def test_f4_edge_cases(golden):
    f = golden('f4_edge_cases')
    p = params_of(f['params'])
    drv = list(f['drivers'])
    day, night = oracle.evapotranspiration(p, *drv)
    same(day, f['day'])
    same(night, f['night'])
    check_sep(oracle.evapotranspiration(p, *drv, separate=True), f)
    # invalid-but-in-range classes 0 and 11 -> NaN
    bplut = {k: f['cls_case_table'][:, j]
             for j, k in enumerate(oracle.PARAM_NAMES)}
    day, night = oracle.evapotranspiration_raster(
        bplut, f['cls_case_cls'], *list(f['cls_case_drivers']))
    same(day, f['cls_case_day'])
    same(night, f['cls_case_night'])
    assert np.isnan(day[1]) and np.isnan(day[2]) and np.isfinite(day[0])


def test_class_out_of_range_raises(golden):
    f = golden('f4_edge_cases')
    bplut = {k: f['cls_case_table'][:, j]
             for j, k in enumerate(oracle.PARAM_NAMES)}
    with pytest.raises(IndexError):
        oracle.gather_params(bplut, np.array([1, 13], np.uint8))


def test_f6_submethods(golden):
    f = golden('f6_submethods')
    p = params_of(f['params'])
    (lw_d, lw_n, sw_d, sw_n, alb, t_d, t_n, t_a, tmin, vpd_d, vpd_n, pa,
     fpar, lai) = list(f['drivers'])
    same(oracle.svp(t_d), f['svp'])
    same(oracle.svp_slope(t_d), f['svp_slope'])
    same(oracle.latent_heat_vaporization(t_d), f['lhv'])
    same(oracle.psychrometric_constant(pa, t_d), f['psychrometric_constant'])
    rh = oracle.rhumidity(t_d, vpd_d)
    same(rh, f['rhumidity'])
    same(oracle.air_density(t_d, pa, rh), f['air_density'])
    g = oracle.soil_heat_flux(p, sw_d * (1 - alb) + lw_d, lw_n, t_d, t_n, t_a)
    same(g[0], f['soil_heat_flux_day'])
    same(g[1], f['soil_heat_flux_night'])
    rs = oracle.radiation_soil(p, lw_d, lw_n, sw_d, sw_n, alb, t_d, t_n, t_a, fpar)
    same(rs[0], f['radiation_soil_day'])
    same(rs[1], f['radiation_soil_night'])
    same(oracle.surface_conductance(p, tmin, vpd_d), f['surface_conductance'])
    rad_c = f['rad_canopy']
    same(oracle.evaporation_wet_canopy(p, pa, t_d, vpd_d, lai, fpar, rad_c),
         f['evaporation_wet_canopy'])
    same(oracle.evaporation_soil(p, pa, t_d, vpd_d, fpar, rs[0]),
         f['evaporation_soil'])
    sat, unsat = oracle.potential_soil_evaporation(p, pa, t_d, vpd_d, fpar, rs[0])
    same(sat, f['potential_soil_sat'])
    same(unsat, f['potential_soil_unsat'])
    same(oracle.transpiration(p, pa, t_d, vpd_d, lai, fpar, rad_c, tmin),
         f['transpiration_day'])
    same(oracle.transpiration(p, pa, t_n, vpd_n, lai, fpar, fpar * lw_n, tmin,
                              daytime=False), f['transpiration_night'])


def test_f7_static_calibration_path(golden):
    """MOD16._evapotranspiration / _et (mod16/__init__.py:162-382), row N2."""
    f = golden('f7_static_path')
    params = [f['params'][k:k + 1] for k in range(11)]        # (1 x N) each
    drv = list(f['drivers'])
    day, night = oracle.et_static_daynight(params, *drv)
    same(day, f['day'])
    same(night, f['night'])
    same(oracle.et_static(params, *drv), f['et'])
    day, night = oracle.et_static_daynight(
        params, *drv, r_corr_list=[f['r_corr_day'], f['r_corr_night']])
    same(day, f['day_rcorr'])
    same(night, f['night_rcorr'])
    cold = list(drv)
    cold[8] = f['tmin_cold']
    day, night = oracle.et_static_daynight(params, *cold)
    same(day, f['day_cold'])      # no pixel with g_surf > 0: transpiration off everywhere
    same(night, f['night_cold'])
    assert not np.array_equal(f['day_cold'], f['day'])
    # the reference's own assertions, tests/tests.py:85-87
    g = golden('f1_tests_scalars')
    et = oracle.et_static(list(g['params']), *list(g['drivers']))
    assert round(float(et), 2) == 40.94 and float(et) == float(g['et_static'])


def test_potential_transpiration_known_form(golden):
    """oracle.potential_transpiration against the closed form of the reference
    (mod16/__init__.py:546-602) evaluated with the reference's f6 vectors."""
    f = golden('f6_submethods')
    (lw_d, lw_n, sw_d, sw_n, alb, t_d, t_n, t_a, tmin, vpd_d, vpd_n, pa, fpar, lai) = \
        list(f['drivers'])
    rh = f['rhumidity']
    fw = np.where(rh < 0.7, 0, np.power(rh, 4))
    want = (1.26 * (f['svp_slope'] * (fpar * (sw_d * (1 - alb) + lw_d))) * (1 - fw)) / \
        (f['svp_slope'] + f['psychrometric_constant'])
    same(oracle.potential_transpiration(lw_d, sw_d, alb, pa, t_d, vpd_d, fpar), want)


def test_f8_raw_driver_preprocessing(golden):
    """N1: the reference's pre-processing (MOD16.vpd, MOD16.air_pressure, fPAR /
    100, LAI / 10) + forward run + 8-day unit, bit for bit."""
    f = golden('f8_raw_drivers')
    raw = list(f['raw'])
    same(oracle.vpd_from_humidity(raw[9], raw[11], raw[5]), f['vpd_day'])
    same(oracle.air_pressure(raw[13]), f['pressure'])
    bplut = {k: f['table'][:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    day, night, total = oracle.evapotranspiration_raw(
        bplut, f['cls'], raw, f['fpar_pct'], f['lai_x10'], f['day_hours'])
    same(day, f['day'])
    same(night, f['night'])
    same(total, f['total8'])
    assert np.isnan(f['day'][0, 1]) and np.isnan(f['night'][1, 1])     # fill codes


def test_f9_potential_transpiration_and_radiation_net(golden):
    """Reference outputs of MOD16.potential_transpiration (:546-602) and
    radiation_net (:1293-1337), instead of re-typed closed forms."""
    f = golden('f9_round2')
    (lw_d, lw_n, sw_d, sw_n, alb, t_d, t_n, t_a, tmin, vpd_d, vpd_n, pa, fpar, lai) = \
        list(f['drivers'])
    same(oracle.potential_transpiration(lw_d, sw_d, alb, pa, t_d, vpd_d, fpar),
         f['potential_transpiration'])
    same(oracle.potential_transpiration(lw_d, sw_d, alb, pa, t_d, vpd_d, fpar, alpha=1.0),
         f['potential_transpiration_alpha1'])
    same(oracle.potential_transpiration(lw_d, sw_d, alb, pa, t_d, vpd_d, fpar,
                                        rh=f['rhumidity'], f_wet=f['f_wet']),
         f['potential_transpiration_given'])
    same(oracle.radiation_net(sw_d, alb, t_d), f['radiation_net'])


@pytest.mark.parametrize('tag,tiny', [('1e-3', 1e-3), ('1e-12', 1e-12)])
def test_f9_tiny_argument(golden, tag, tiny):
    """`tiny` other than the default (reference :869, :1157)."""
    f = golden('f9_round2')
    p = params_of(f['params'])
    (lw_d, lw_n, sw_d, sw_n, alb, t_d, t_n, t_a, tmin, vpd_d, vpd_n, pa, fpar, lai) = \
        list(f['drivers'])
    rad_c = f['rad_canopy']
    same(oracle.evaporation_wet_canopy(p, pa, t_d, vpd_d, lai, fpar, rad_c, tiny=tiny),
         f['wet_canopy_tiny_' + tag])
    same(oracle.transpiration(p, pa, t_d, vpd_d, lai, fpar, rad_c, tmin, tiny=tiny),
         f['transpiration_day_tiny_' + tag])
    same(oracle.transpiration(p, pa, t_n, vpd_n, lai, fpar, fpar * lw_n, tmin,
                              daytime=False, tiny=tiny), f['transpiration_night_tiny_' + tag])
    # the argument matters on these inputs
    assert not np.array_equal(f['wet_canopy_tiny_1e-3'], f['wet_canopy_tiny_1e-12'], equal_nan=True)


def test_f9_static_path_tiny(golden):
    f = golden('f9_round2')
    params = [f['static_params'][k:k + 1] for k in range(11)]
    drv = list(f['static_drivers'])
    day, night = oracle.et_static_daynight(params, *drv, tiny=1e-2)
    same(day, f['static_day_tiny_1e-2'])
    same(night, f['static_night_tiny_1e-2'])
    # MOD16._et drops its `tiny` (reference :190-192)
    same(oracle.et_static(params, *drv), f['static_et_ignores_tiny'])
    assert not np.array_equal(day + night, f['static_et_ignores_tiny'], equal_nan=True)


def bcast_inputs(f):
    """The mixed-shape argument list of the F9 broadcast case: (T, N) dense
    drivers, (N,) temp_annual / pressure, a (T, 1) albedo, scalar sw_rad_night."""
    dense = list(f['bcast_dense'])
    lw_d, lw_n, sw_d, t_d, t_n, tmin, vpd_d, vpd_n, fpar, lai = dense
    return [lw_d, lw_n, sw_d, 0, f['bcast_albedo'], t_d, t_n, f['bcast_temp_annual'], tmin,
            vpd_d, vpd_n, f['bcast_pressure'], fpar, lai]


def test_f9_broadcast_rows_and_columns(golden):
    """(N,) against (T, N) (reference mod16/__init__.py:180-181, notebook cell 17)."""
    f = golden('f9_round2')
    p = {k: f['bcast_site_params'][j] for j, k in enumerate(oracle.PARAM_NAMES)}
    drv = bcast_inputs(f)
    day, night = oracle.evapotranspiration(p, *drv)
    same(day, f['bcast_day'])
    same(night, f['bcast_night'])
    res = oracle.evapotranspiration(p, *drv, separate=True)
    for name, got in zip(SEP, list(res[0]) + list(res[1])):
        same(got, f['bcast_' + name])


def test_fused_numpy_baseline_agrees_with_the_oracle(golden):
    """bench.py's second CPU baseline (fused / strength-reduced numpy) is the
    same computation: identical NaN and exact-zero masks, 1e-9, on the reference's F3
    raster and on a synthetic tile with the generator's special values."""
    from oracle import fused_numpy, synth
    f = golden('f3_random64_f64')
    bplut = {k: f['table'][:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    cases = [(f['cls'], list(f['drivers']), (f['day'], f['night']))]
    cls, drv = synth.drivers((300, 400), seed=3)
    cases.append((cls, drv, oracle.evapotranspiration_raster(bplut, cls, *drv)))
    for cls, drv, want in cases:
        got = fused_numpy.evapotranspiration_raster(bplut, cls, *drv)
        for g, w in zip(got, want):
            assert np.array_equal(np.isnan(g), np.isnan(w))
            assert np.array_equal(g == 0, w == 0)
            ok = np.isfinite(w) & (w != 0)
            assert np.max(np.abs(g[ok] - w[ok]) / np.abs(w[ok])) < 1e-9
