"""GPU parity: the HIP path (through the C ABI, via mod16_amd) against the
golden vectors of the reference and against the oracle on seeded inputs.

Tolerances (float64): north_star asks for 1e-5 relative. The tests hold the
worst pixel of the EXACT kernel (reference operation order) to 1e-10 and of the
FAST kernel (production) to 1e-8, both with identical NaN and exact-zero
masks; the worst pixels are cancellation cases (s*A_soil against the
aerodynamic term), where one ulp of exp() is amplified, the typical pixel
agrees to ~1e-15 (medians are asserted in test_conus_tile_1200)."""
import numpy as np
import pytest

from oracle import mod16_oracle as oracle
from oracle import synth
from parity import assert_mixed_parity, assert_parity, in_a_fresh_thread, same_bits

pytestmark = pytest.mark.gpu

RTOL = {'fast': 1e-8, 'exact': 1e-10}
MEDIAN = {'fast': 1e-13, 'exact': 1e-14}
SEP = ('canopy_day', 'soil_day', 'trans_day',
       'canopy_night', 'soil_night', 'trans_night')


@pytest.fixture(scope='module')
def m16():
    import mod16_amd
    return mod16_amd


def math_flag(m16, mode):
    return {'fast': m16._lib.MATH_FAST, 'exact': m16._lib.MATH_EXACT}[mode]


def model(m16, params, mode):
    m = m16.MOD16(dict(zip(oracle.PARAM_NAMES, params)))
    m.math = math_flag(m16, mode)
    return m


def check_sep(res, f, rtol, what):
    flat = list(res[0]) + list(res[1])
    return max(assert_parity(np.asarray(got), f[name], rtol, what + ':' + name)
               for name, got in zip(SEP, flat))


@pytest.mark.parametrize('mode', ['fast', 'exact'])
def test_f1_scalar_set(m16, golden, mode):
    f = golden('f1_tests_scalars')
    m = model(m16, f['params'], mode)
    day, night = m.evapotranspiration(*[float(v) for v in f['drivers']])
    assert np.ndim(day) == 0 and np.ndim(night) == 0
    assert_parity(np.asarray(day), f['day'], RTOL[mode], 'day')
    assert_parity(np.asarray(night), f['night'], RTOL[mode], 'night')
    check_sep(m.evapotranspiration(*f['drivers'], separate=True), f, RTOL[mode], 'f1')
    # the reference's own assertion, tests/tests.py:88-90
    lhv = lambda t: (2.501 - 0.002361 * (t - 273.15)) * 1e6
    assert round(float(day * lhv(293) + night * lhv(290)), 1) == 41.0


@pytest.mark.parametrize('mode', ['fast', 'exact'])
def test_f2_verify_three_pixels(m16, golden, mode):
    f = golden('f2_verify_3pixel')
    m = model(m16, f['params'], mode)
    drv = [f['drv_' + k] for k in oracle.DRIVER_NAMES]
    res = m.evapotranspiration(*drv, f_wet=np.array((0, 0.4, 0.8)), separate=True)
    check_sep(res, f, RTOL[mode], 'f2')
    assert res[0][0].shape == (3,)


@pytest.mark.parametrize('mode', ['fast', 'exact'])
def test_f4_edge_cases(m16, golden, mode):
    f = golden('f4_edge_cases')
    m = model(m16, f['params'], mode)
    drv = list(f['drivers'])
    res = m.evapotranspiration(*drv, separate=True)
    flat = list(res[0]) + list(res[1])
    for i, case in enumerate(f['names']):
        for name, got in zip(SEP, flat):
            assert_parity(got[i:i + 1], f[name][i:i + 1], RTOL[mode],
                          '%s:%s' % (case, name))
    day, night = m.evapotranspiration(*drv)
    assert_parity(day, f['day'], RTOL[mode], 'day')
    assert_parity(night, f['night'], RTOL[mode], 'night')


@pytest.mark.parametrize('mode', ['fast', 'exact'])
def test_f4_invalid_classes_give_nan(m16, golden, mode):
    f = golden('f4_edge_cases')
    day, night = m16.evapotranspiration_raster(
        f['cls_case_table'], f['cls_case_cls'], *list(f['cls_case_drivers']),
        math=math_flag(m16, mode))
    assert_parity(day, f['cls_case_day'], RTOL[mode], 'day')
    assert_parity(night, f['cls_case_night'], RTOL[mode], 'night')


@pytest.mark.parametrize('mode', ['fast', 'exact'])
def test_f3_multiclass_raster(m16, golden, mode):
    f = golden('f3_random64_f64')
    drv = list(f['drivers'])
    day, night = m16.evapotranspiration_raster(
        f['table'], f['cls'], *drv, math=math_flag(m16, mode))
    assert day.shape == (64, 64) and day.dtype == np.float64
    e1 = assert_parity(day, f['day'], RTOL[mode], 'day')
    e2 = assert_parity(night, f['night'], RTOL[mode], 'night')
    res = m16.evapotranspiration_raster(
        f['table'], f['cls'], *drv, separate=True, math=math_flag(m16, mode))
    e3 = check_sep(res, f, RTOL[mode], 'f3')
    print('\n[f3 %s] max rel err: day %.2e night %.2e components %.2e' % (mode, e1, e2, e3))


@pytest.mark.parametrize('mode', ['fast', 'exact'])
def test_f3_per_pixel_parameter_arrays(m16, golden, mode):
    """The reference idiom itself: MOD16({k: bplut[k][cls]}) with array
    parameters must equal the in-kernel class look-up."""
    f = golden('f3_random64_f64')
    params = {k: f['table'][:, j][f['cls']] for j, k in enumerate(oracle.PARAM_NAMES)}
    m = m16.MOD16(params)
    m.math = math_flag(m16, mode)
    day, night = m.evapotranspiration(*list(f['drivers']))
    assert_parity(day, f['day'], RTOL[mode], 'day')
    assert_parity(night, f['night'], RTOL[mode], 'night')


def test_f5_float32_raster(m16, golden):
    """float32 in -> float32 out (float64 arithmetic inside); compared with
    the reference's own all-float32 run, whose own error against float64 is
    median 2e-7, p99 1.4e-5, max 1.6e-2: NaN masks identical,
    99th percentile of the relative difference below 1e-4."""
    f = golden('f5_random64_f32')
    drv = list(f['drivers'])
    day, night = m16.evapotranspiration_raster(
        f['table'], f['cls'], *drv)
    assert day.dtype == np.float32 and night.dtype == np.float32
    for got, want in ((day, f['day']), (night, f['night'])):
        assert np.array_equal(np.isnan(got), np.isnan(want))
        ok = np.isfinite(want) & (want != 0)
        err = np.abs(got[ok].astype(np.float64) - want[ok]) / np.abs(want[ok])
        assert np.percentile(err, 99) < 1e-4, np.percentile(err, 99)
        assert np.median(err) < 5e-6, np.median(err)


def test_f5_float32_exact_follows_reference_float32(m16, golden):
    """MATH_EXACT on float32 data does float32 arithmetic in the reference's
    operation order, like numpy does for the reference: masks identical, the
    bulk of the pixels within a few float32 ulps."""
    f = golden('f5_random64_f32')
    day, night = m16.evapotranspiration_raster(
        f['table'], f['cls'], *list(f['drivers']), math=m16._lib.MATH_EXACT)
    for got, want in ((day, f['day']), (night, f['night'])):
        assert got.dtype == np.float32
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert np.array_equal(got == 0, want == 0)
        ok = np.isfinite(want) & (want != 0)
        err = np.abs(got[ok].astype(np.float64) - want[ok]) / np.abs(want[ok])
        assert np.median(err) < 2e-7 and np.percentile(err, 99) < 2e-5, \
            (np.median(err), np.percentile(err, 99))


def test_float32_storage_float64_arithmetic(m16):
    """BASELINE.json configs[4] (tolerance check): the production float32
    kernel widens on load, computes in float64 and rounds once on store, so on
    float32-representable inputs it equals the float64 kernel up to that one
    rounding (2^-24), with identical masks."""
    cls, drv32 = synth.drivers((700, 900), seed=5, dtype=np.float32)
    f = np.load(__import__('os').path.join(__import__('conftest').GOLDEN, 'f3_random64_f64.npz'))
    d32, n32 = m16.evapotranspiration_raster(f['table'], cls, *drv32)
    d64, n64 = m16.evapotranspiration_raster(
        f['table'], cls, *[d.astype(np.float64) for d in drv32])
    assert d32.dtype == np.float32 and d64.dtype == np.float64
    for a, b in ((d32, d64), (n32, n64)):
        want = b.astype(np.float32)
        assert np.array_equal(np.isnan(a), np.isnan(b))
        assert np.array_equal(a, want, equal_nan=True)      # exactly the rounded float64 result
        ok = np.isfinite(b) & (b != 0)
        assert np.max(np.abs(a[ok] - b[ok]) / np.abs(b[ok])) <= 2.0**-24 * 1.0001


def test_class_code_out_of_range_raises(m16, golden):
    f = golden('f3_random64_f64')
    cls = f['cls'].copy()
    cls[3, 5] = 13
    with pytest.raises(IndexError):
        m16.evapotranspiration_raster(f['table'], cls, *list(f['drivers']))
    # the context stays usable and the flag is cleared
    day, _ = m16.evapotranspiration_raster(f['table'], f['cls'], *list(f['drivers']))
    assert_parity(day, f['day'], RTOL['fast'], 'day after error')


def test_missing_parameter_is_keyerror(m16):
    with pytest.raises(KeyError):
        m16.MOD16({'tmin_close': -8})


@pytest.mark.parametrize('n', [0, 1, 2, 3, 255, 257, 4099, 1000003])
def test_ragged_sizes(m16, golden, n):
    """Empty, odd and non-multiple-of-block sizes take the scalar tail path."""
    f = golden('f3_random64_f64')
    bplut = {k: f['table'][:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    cls, drv = synth.drivers((n,), seed=n)
    day, night = m16.evapotranspiration_raster(f['table'], cls, *drv)
    assert day.shape == (n,)
    if n:
        wd, wn = oracle.evapotranspiration_raster(bplut, cls, *drv)
        assert_parity(day, wd, RTOL['fast'], 'day')
        assert_parity(night, wn, RTOL['fast'], 'night')


def test_broadcast_scalars_and_rows(m16, golden):
    """Scalars (stride 0) and a (N,) row against (T, N) drivers, as the
    reference's notebooks call it (pressure, temp_annual per site)."""
    f = golden('f1_tests_scalars')
    p = dict(zip(oracle.PARAM_NAMES, f['params']))
    _, drv = synth.drivers((5, 40), seed=3, special=False)
    drv[3] = 0                       # sw_rad_night scalar
    drv[11] = drv[11][0]             # pressure (N,)
    drv[7] = drv[7][0]               # temp_annual (N,)
    drv[1] = -30.0                   # lw_net_night scalar
    day, night = m16.MOD16(p).evapotranspiration(*drv)
    wd, wn = oracle.evapotranspiration(p, *drv)
    assert day.shape == (5, 40)
    assert_parity(day, wd, RTOL['fast'], 'day')
    assert_parity(night, wn, RTOL['fast'], 'night')


@pytest.mark.parametrize('mode', ['fast', 'exact'])
def test_conus_tile_1200(m16, golden, mode):
    """BASELINE.json configs[1]: 1200 x 1200 tile, float64, vs the oracle."""
    f = golden('f3_random64_f64')
    bplut = {k: f['table'][:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    cls, drv = synth.drivers((1200, 1200), seed=16)
    res = m16.evapotranspiration_raster(
        f['table'], cls, *drv, separate=True, math=math_flag(m16, mode))
    want = oracle.evapotranspiration_raster(bplut, cls, *drv, separate=True)
    errs = []
    for a, b, name in zip(list(res[0]) + list(res[1]),
                          list(want[0]) + list(want[1]), SEP):
        errs.append(assert_parity(a, b, RTOL[mode], name))
        ok = np.isfinite(b) & (b != 0)
        med = np.median(np.abs(a[ok] - b[ok]) / np.abs(b[ok]))
        assert med < MEDIAN[mode], (name, med)
    day, night = m16.evapotranspiration_raster(
        f['table'], cls, *drv, math=math_flag(m16, mode))
    wd, wn = oracle.evapotranspiration_raster(bplut, cls, *drv)
    e_d = assert_parity(day, wd, RTOL[mode], 'day')
    e_n = assert_parity(night, wn, RTOL[mode], 'night')
    print('\n[1200x1200 %s] max rel err: components %.2e day %.2e night %.2e'
          % (mode, max(errs), e_d, e_n))
    assert max(e_d, e_n) < 1e-5   # the north_star bar


@pytest.mark.parametrize('mode', ['fast', 'exact'])
def test_potential_et(m16, golden, mode):
    """PET from the same pass. The checker composes
    the reference's own component methods (oracle.potential_et); the reference
    vectors of those components are pinned in f6 / test_gpu_methods."""
    f = golden('f3_random64_f64')
    bplut = {k: f['table'][:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    drv = list(f['drivers'])
    day, night, pet_d, pet_n = m16.evapotranspiration_raster(
        f['table'], f['cls'], *drv, pet=True, math=math_flag(m16, mode))
    assert_parity(day, f['day'], RTOL[mode], 'day')
    assert_parity(night, f['night'], RTOL[mode], 'night')
    want = oracle.potential_et(oracle.gather_params(bplut, f['cls']), *drv)
    assert_parity(pet_d, want[0], RTOL[mode], 'pet day')
    assert_parity(pet_n, want[1], RTOL[mode], 'pet night')
    # scalar parameters + edge cases through the class method
    g = golden('f4_edge_cases')
    p = dict(zip(oracle.PARAM_NAMES, g['params']))
    m = m16.MOD16(p)
    m.math = math_flag(m16, mode)
    res = m.evapotranspiration_and_pet(*list(g['drivers']))
    want = oracle.potential_et(p, *list(g['drivers']))
    for i, case in enumerate(g['names']):
        assert_parity(res[2][i:i + 1], want[0][i:i + 1], RTOL[mode], 'pet day ' + str(case))
        assert_parity(res[3][i:i + 1], want[1][i:i + 1], RTOL[mode], 'pet night ' + str(case))
    assert_parity(res[0], g['day'], RTOL[mode], 'day')


@pytest.mark.parametrize('mode', ['fast', 'exact'])
def test_raw_driver_forward_run(m16, golden, mode):
    """raw reanalysis fields and uint8 fPAR / LAI in,
    ET out, against the reference's own pre-processing + forward run (f8)."""
    f = golden('f8_raw_drivers')
    raw = list(f['raw'])
    res = m16.evapotranspiration_raw(
        f['table'], f['cls'], *raw, f['fpar_pct'], f['lai_x10'],
        day_hours=f['day_hours'], math=math_flag(m16, mode))
    assert res[0].shape == (48, 50)
    assert_parity(res[0], f['day'], RTOL[mode], 'day')
    assert_parity(res[1], f['night'], RTOL[mode], 'night')
    assert_parity(res[2], f['total8'], RTOL[mode], 'total8')
    day, night = m16.evapotranspiration_raw(
        f['table'], f['cls'], *raw, f['fpar_pct'], f['lai_x10'], math=math_flag(m16, mode))
    assert np.array_equal(day, res[0], equal_nan=True)
    # scalar raw drivers broadcast like everywhere else
    raw2 = list(raw)
    raw2[13] = 350.0
    d2, _ = m16.evapotranspiration_raw(
        f['table'], f['cls'], *raw2, f['fpar_pct'], f['lai_x10'], math=math_flag(m16, mode))
    bplut = {k: f['table'][:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    raw2[13] = np.full((48, 50), 350.0)
    w2, _ = oracle.evapotranspiration_raw(bplut, f['cls'], raw2, f['fpar_pct'], f['lai_x10'])
    assert_parity(d2, w2, RTOL[mode], 'scalar elevation')


def test_rasters_on_disk(m16, tmp_path):
    """SURVEY.md 8f N4 (.npy memory maps through the HOST mode): same bits as
    the in-memory call, outputs written in place."""
    from mod16_amd import io as m16io
    f = np.load(__import__('os').path.join(__import__('conftest').GOLDEN, 'f3_random64_f64.npz'))
    cls, drv = synth.drivers((1500, 1700), seed=8)
    paths = {}
    for name, d in zip(m16io.DRIVER_NAMES, drv):
        paths[name] = str(tmp_path / (name + '.npy'))
        np.save(paths[name], d)
    np.save(str(tmp_path / 'cls.npy'), cls)
    day, night = m16io.evapotranspiration_npy(
        f['table'], str(tmp_path / 'cls.npy'), paths, str(tmp_path / 'day.npy'), str(tmp_path / 'night.npy'))
    want_d, want_n = m16.evapotranspiration_raster(f['table'], cls, *drv)
    assert np.array_equal(np.load(str(tmp_path / 'day.npy')), want_d, equal_nan=True)
    assert np.array_equal(np.load(str(tmp_path / 'night.npy')), want_n, equal_nan=True)
    assert day.shape == (1500, 1700) and isinstance(day, np.memmap)
    # ... and dealt over two contexts of the GPU (mod16_amd.multi): memory-mapped inputs and outputs
    # shared by the shards, the same bytes in the files
    m16io.evapotranspiration_npy(f['table'], str(tmp_path / 'cls.npy'), paths, str(tmp_path / 'day2.npy'),
                                 str(tmp_path / 'night2.npy'), devices=[0, 0])
    assert np.array_equal(np.load(str(tmp_path / 'day2.npy')), want_d, equal_nan=True)
    assert np.array_equal(np.load(str(tmp_path / 'night2.npy')), want_n, equal_nan=True)
    with pytest.raises(KeyError):
        m16io.evapotranspiration_npy(f['table'], str(tmp_path / 'cls.npy'), {'lai': paths['lai']},
                                     str(tmp_path / 'd2.npy'), str(tmp_path / 'n2.npy'))
    # out= of the in-memory interface: wrong shape / dtype is refused
    with pytest.raises(ValueError):
        m16.evapotranspiration_raster(f['table'], cls, *drv, out=[np.empty((3, 3)), np.empty((3, 3))])


def test_float32_class_interface_is_float64_arithmetic_rounded_once(m16, golden):
    """MOD16(params).evapotranspiration on float32 arrays (scalar and per-pixel
    parameters: the plain kernels, not the class-raster pipeline), totals,
    components and potential ET: the float64 result rounded once, bit for bit."""
    f = golden('f3_random64_f64')
    cls, drv32 = synth.drivers((333, 517), seed=9, dtype=np.float32)
    drv64 = [d.astype(np.float64) for d in drv32]
    bplut = {k: f['table'][:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    # scalar parameters reach the float32 kernel rounded to float32: give the float64 run the same values
    scalar = m16.MOD16({k: float(np.float32(bplut[k][7])) for k in oracle.PARAM_NAMES})
    per_pixel32 = m16.MOD16({k: bplut[k][cls].astype(np.float32) for k in oracle.PARAM_NAMES})
    per_pixel64 = m16.MOD16({k: bplut[k][cls].astype(np.float32).astype(np.float64) for k in oracle.PARAM_NAMES})
    for m32, m64 in ((scalar, scalar), (per_pixel32, per_pixel64)):
        got = m32.evapotranspiration(*drv32)
        want = m64.evapotranspiration(*drv64)
        for a, b in zip(got, want):
            assert a.dtype == np.float32
            assert np.array_equal(a, b.astype(np.float32), equal_nan=True)
        got = m32.evapotranspiration(*drv32, separate=True)
        want = m64.evapotranspiration(*drv64, separate=True)
        for a, b in zip(got[0] + got[1], want[0] + want[1]):
            assert np.array_equal(a, b.astype(np.float32), equal_nan=True)
        got = m32.evapotranspiration_and_pet(*drv32)
        want = m64.evapotranspiration_and_pet(*drv64)
        for a, b in zip(got, want):
            assert np.array_equal(a, b.astype(np.float32), equal_nan=True)


def test_scalar_drivers_over_many_tiles(m16, golden):
    """Broadcast scalars (here sw_rad_night = 0 and one albedo) with rasters that span
    several staged tiles: the plain kernels under the staging threads against the same
    call with the scalars expanded (the pipeline), totals and potential ET."""
    f = golden('f3_random64_f64')
    n = 3 * (1 << 21) + 7
    cls, drv = synth.drivers((n,), seed=10)
    drv_s = list(drv)
    drv_s[3] = 0.0
    drv_s[4] = 0.17
    drv_d = list(drv)
    drv_d[3] = np.zeros(n)
    drv_d[4] = np.full(n, 0.17)
    got = m16.evapotranspiration_raster(f['table'], cls, *drv_s, pet=True)
    want = m16.evapotranspiration_raster(f['table'], cls, *drv_d, pet=True)
    for a, b, what in zip(got, want, ('day', 'night', 'pet day', 'pet night')):
        assert_parity(a, b, 1e-11, what)     # two instantiations of one pixel function: contraction may differ


def test_f9_rows_and_columns_against_the_reference(m16, golden):
    """(N,) rows, a (T, 1) column and a scalar against (T, N) drivers with per-site
    (N,) parameter arrays -- the reference's own outputs (tests/golden/f9_round2.npz),
    through mod16_et2_* (nothing made dense on the host)."""
    f = golden('f9_round2')
    names = m16.MOD16.required_parameters
    site_par = {k: f['bcast_site_params'][j] for j, k in enumerate(names)}
    dense = list(f['bcast_dense'])
    lw_d, lw_n, sw_d, t_d, t_n, tmin, vpd_d, vpd_n, fpar, lai = dense
    drv = [lw_d, lw_n, sw_d, 0, f['bcast_albedo'], t_d, t_n, f['bcast_temp_annual'], tmin,
           vpd_d, vpd_n, f['bcast_pressure'], fpar, lai]
    seen = []
    real = m16._lib.Context.et2

    def spy(self, dtype, cls, ckind, drivers, dkind, params, pkind, inner, n, *a, **k):
        seen.append((list(dkind), list(pkind) if pkind is not None else None, inner, n))
        return real(self, dtype, cls, ckind, drivers, dkind, params, pkind, inner, n, *a, **k)

    m16._lib.Context.et2 = spy
    try:
        for math, rtol in ((m16._lib.MATH_FAST, 1e-8), (m16._lib.MATH_EXACT, 1e-10)):
            model = m16.MOD16(site_par)
            model.math = math
            day, night = model.evapotranspiration(*drv)
            assert_parity(day, f['bcast_day'], rtol, 'day')
            assert_parity(night, f['bcast_night'], rtol, 'night')
            sep = model.evapotranspiration(*drv, separate=True)
            for name, got in zip(('canopy_day', 'soil_day', 'trans_day', 'canopy_night', 'soil_night',
                                  'trans_night'), list(sep[0]) + list(sep[1])):
                assert_parity(got, f['bcast_' + name], 10 * rtol, name)
    finally:
        m16._lib.Context.et2 = real
    T, N = f['bcast_day'].shape
    B = m16._lib
    assert seen and all(s[2] == N and s[3] == T * N for s in seen)
    assert seen[0][0] == [1, 1, 1, B.BC_SCALAR, B.BC_COL, 1, 1, B.BC_ROW, 1, 1, 1, B.BC_ROW, 1, 1]
    assert seen[0][1] == [B.BC_ROW] * 11
    # the class-raster form: a per-site PFT vector against (T, N) drivers
    table = np.full((13, 11), np.nan)
    for j in range(N):
        table[int(f['bcast_site_cls'][j])] = f['bcast_site_params'][:, j]
    day, night = m16.evapotranspiration_raster(table, f['bcast_site_cls'], *drv)
    assert_parity(day, f['bcast_day'], 1e-8, 'day (class vector)')
    assert_parity(night, f['bcast_night'], 1e-8, 'night (class vector)')


def test_rows_and_columns_over_several_staged_tiles(m16):
    """The same kinds on a raster of more than one staged tile (2 Mi pixels), float64
    and float32, against the oracle on the broadcast inputs."""
    from oracle import synth
    from mod16_amd.utils import restore_bplut, bplut_table
    from mod16_amd.models import COLLECTION61_BPLUT
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    T, N = 37, 70001
    cls, drv = synth.drivers((T, N), seed=61)
    site_cls = cls[0].copy()
    mixed = list(drv)
    mixed[7] = drv[7][0].copy()            # temp_annual (N,)
    mixed[11] = drv[11][0].copy()          # pressure (N,)
    mixed[4] = drv[4][:, :1].copy()        # albedo (T, 1)
    mixed[3] = 0.0
    full = [np.broadcast_to(np.asarray(v, np.float64), (T, N)) for v in mixed]
    want = oracle.evapotranspiration_raster(bplut, np.broadcast_to(site_cls, (T, N)), *full)
    got = m16.evapotranspiration_raster(table, site_cls, *mixed)
    assert_parity(got[0], want[0], 1e-8, 'day')
    assert_parity(got[1], want[1], 1e-8, 'night')
    m32 = [np.asarray(v, np.float32) if isinstance(v, np.ndarray) else v for v in mixed]
    got32 = m16.evapotranspiration_raster(table, site_cls, *m32)
    want32 = oracle.evapotranspiration_raster(
        bplut, np.broadcast_to(site_cls, (T, N)),
        *[np.broadcast_to(np.asarray(v, np.float64), (T, N)) for v in m32])
    assert got32[0].dtype == np.float32
    assert_parity(got32[0], want32[0].astype(np.float32), 1e-6, 'day f32')
    assert_parity(got32[1], want32[1].astype(np.float32), 1e-6, 'night f32')


def _special_value_rasters(values, per=200, seed=123):
    """Plausible drivers with one special value in one driver per pixel: `per` pixels for
    every (driver, value) pair. Returns (class raster, 14 drivers, pair index per pixel)."""
    rng = np.random.default_rng(seed)
    n = per * 14 * len(values)
    t_d = rng.uniform(255, 305, n)
    t_n = t_d - rng.uniform(0, 12, n)
    es = lambda t: 610.8 * np.exp(17.27 * (t - 273.15) / (t - 273.15 + 237.3))
    drv = [rng.uniform(-100, 0, n), rng.uniform(-50, 0, n), rng.uniform(0, 360, n), np.zeros(n),
           rng.uniform(0.1, 0.22, n), t_d, t_n, rng.uniform(265, 300, n), t_n - rng.uniform(0, 3, n),
           es(t_d) * (1 - rng.uniform(0.05, 1, n)), es(t_n) * (1 - rng.uniform(0.05, 1, n)),
           rng.uniform(7e4, 101340, n), rng.uniform(0.02, 0.89, n), rng.uniform(0.13, 5.34, n)]
    which = np.repeat(np.arange(14 * len(values)), per)
    for j in range(14):
        for s, v in enumerate(values):
            drv[j][which == j * len(values) + s] = v
    cls = rng.choice(np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12], np.uint8), n)
    return cls, drv, which


def test_special_values_reference_order_kernel(m16, golden):
    """Zeros, NaN, infinities, fill values, the pole of the Tetens formula (35.85 K), the
    largest float32, 1e+-300 -- one of them in one driver per pixel, every driver: the kernel
    that keeps the reference's operation order reproduces the oracle's NaN, zero AND inf masks
    on all of them and the values to 1e-10 (whatever garbage the reference computes from
    garbage, this computes the same)."""
    values = [0.0, -0.0, np.nan, -9999.0, 65535.0, 1.0, -1.0, 1e-7, 273.15, 35.85, 34.15, 3.4e38,
              -3.4e38, 1e300, -1e300, 1e-300, np.inf, -np.inf]
    cls, drv, _ = _special_value_rasters(values)
    table = golden('f3_random64_f64')['table']
    bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    with np.errstate(all='ignore'):
        want = oracle.evapotranspiration_raster(bplut, cls, *drv)
    got = m16.evapotranspiration_raster(table, cls, *drv, math=m16._lib.MATH_EXACT)
    assert_parity(got[0], want[0], 1e-10, 'day')
    assert_parity(got[1], want[1], 1e-10, 'night')


SPECIAL_VALUES = [0.0, -0.0, np.nan, -9999.0, 65535.0, 1.0, -1.0, 1e-7, 273.15, 35.85, 34.15, 3.4e38,
                  -3.4e38, 1e300, -1e300, 1e-300, np.inf, -np.inf, 1e15]


def test_special_values_fast_kernel(m16, golden):
    """The default (strength-reduced) arithmetic on the same construction, every value of the list
    in every driver, nothing masked out: NaN, zero and inf masks identical to the oracle's, values
    to 1e-8. Inside its domain the fast form computes them itself; a pixel outside it (an
    infinity, a fill value left in a temperature or the pressure, the pole of the Tetens formula
    -- mod16_physics.hpp, "domain guard") is computed again in the reference's operation order
    inside the same kernel. MERRA-2's 1e15 fill is in the list as well."""
    values = SPECIAL_VALUES
    cls, drv, which = _special_value_rasters(values)
    table = golden('f3_random64_f64')['table']
    bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    with np.errstate(all='ignore'):
        want = oracle.evapotranspiration_raster(bplut, cls, *drv)
    got = m16.evapotranspiration_raster(table, cls, *drv, math=m16._lib.MATH_FAST)
    assert_parity(got[0], want[0], 1e-8, 'day')
    assert_parity(got[1], want[1], 1e-8, 'night')
    # the same pixels through the six components and the potential-ET outputs (other instances
    # of the pipeline, same guard)
    with np.errstate(all='ignore'):
        want6 = oracle.evapotranspiration_raster(bplut, cls, *drv, separate=True)
    got6 = m16.evapotranspiration_raster(table, cls, *drv, separate=True)
    for g3, w3, period in zip(got6, want6, ('day', 'night')):
        for g, w, part in zip(g3, w3, ('canopy', 'soil', 'transpiration')):
            assert_parity(g, w, 1e-8, period + ' ' + part)
    # one pixel per thread (ragged tail / unaligned) and per-pixel parameter arrays
    n = cls.size - 3
    params = {k: bplut[k][cls[1:n]] for k in oracle.PARAM_NAMES}
    got1 = m16.MOD16(params).evapotranspiration(*[d[1:n] for d in drv])
    assert_parity(got1[0], want[0][1:n], 1e-8, 'day, parameter arrays')
    assert_parity(got1[1], want[1][1:n], 1e-8, 'night, parameter arrays')


def test_special_values_float32_rasters(m16, golden):
    """float32 rasters, FAST (float64 arithmetic, rounded once) and MIXED: every special value a
    float32 can hold in every driver against the float64 oracle on the widened inputs. FAST: masks
    identical, values to 1e-6. MIXED: masks identical; its values are held to the mixed form's
    tolerance (1e-3 here; test_gpu_mixed.py has the distribution) -- outside its (tight) domain it
    hands the pixel to the reference-order arithmetic like the FAST form does."""
    values = [v for v in SPECIAL_VALUES if not np.isfinite(v) or v == 0 or 1e-37 < abs(v) < 3.41e38]
    cls, drv, which = _special_value_rasters(values)
    drv = [d.astype(np.float32) for d in drv]
    table = golden('f3_random64_f64')['table']
    bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    with np.errstate(all='ignore'):
        want = [w.astype(np.float32) for w in
                oracle.evapotranspiration_raster(bplut, cls, *[d.astype(np.float64) for d in drv])]
    got = m16.evapotranspiration_raster(table, cls, *drv, math=m16._lib.MATH_FAST)
    assert_parity(got[0], want[0], 1e-6, 'day, fast')
    assert_parity(got[1], want[1], 1e-6, 'night, fast')
    got = m16.evapotranspiration_raster(table, cls, *drv, math=m16._lib.MATH_MIXED)
    assert_mixed_parity(got[0], want[0], 'day, mixed')
    assert_mixed_parity(got[1], want[1], 'night, mixed')


def _raw_special_rasters(values, per=100, seed=321):
    """Plausible raw drivers (mod16_raw_driver order) with one special value in one field per
    pixel; uint8 fPAR / LAI with their fill codes sprinkled in."""
    rng = np.random.default_rng(seed)
    n = per * 14 * len(values)
    t_d = rng.uniform(255, 305, n)
    t_n = t_d - rng.uniform(0, 12, n)
    raw = [rng.uniform(-100, 0, n), rng.uniform(-50, 0, n), rng.uniform(0, 360, n), np.zeros(n),
           rng.uniform(0.1, 0.22, n), t_d, t_n, rng.uniform(265, 300, n), t_n - rng.uniform(0, 3, n),
           rng.uniform(5e-4, 2e-2, n), rng.uniform(5e-4, 2e-2, n),
           rng.uniform(70000, 101340, n), rng.uniform(70000, 101340, n), rng.uniform(-50, 4500, n)]
    which = np.repeat(np.arange(14 * len(values)), per)
    for j in range(14):
        for s, v in enumerate(values):
            raw[j][which == j * len(values) + s] = v
    fpar = rng.integers(0, 101, n).astype(np.uint8)
    lai = rng.integers(0, 70, n).astype(np.uint8)
    fpar[rng.random(n) < 0.02] = 255
    lai[rng.random(n) < 0.02] = 250
    cls = rng.choice(np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12], np.uint8), n)
    hours = rng.uniform(6, 18, n)
    return cls, raw, fpar, lai, hours


def test_special_values_raw_drivers(m16, golden):
    """The raw-driver forms (pre-processing fused into the kernel) on special values in every raw
    field -- specific humidity, surface pressure and elevation included -- against the oracle's
    restatement of the reference's pre-processing + forward run: masks identical, values to 1e-8
    (float64), 1e-6 (float32 rasters, float64 arithmetic), the mixed form's tolerance (parity.assert_mixed_parity)."""
    values = SPECIAL_VALUES + [-1.7, 44330.0, 5e4, -5e4]
    cls, raw, fpar, lai, hours = _raw_special_rasters(values)
    table = golden('f3_random64_f64')['table']
    bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    with np.errstate(all='ignore'):
        want = oracle.evapotranspiration_raw(bplut, cls, raw, fpar, lai, day_hours=hours)
    got = m16.evapotranspiration_raw(table, cls, *raw, fpar, lai, day_hours=hours)
    for g, w, what in zip(got, want, ('day', 'night', 'total8')):
        assert_parity(g, w, 1e-8, what)
    v32 = [v for v in values if not np.isfinite(v) or v == 0 or 1e-37 < abs(v) < 3.41e38]
    cls, raw, fpar, lai, hours = _raw_special_rasters(v32)
    raw32 = [a.astype(np.float32) for a in raw]
    with np.errstate(all='ignore'):
        want = [w.astype(np.float32) for w in oracle.evapotranspiration_raw(
            bplut, cls, [a.astype(np.float64) for a in raw32], fpar, lai)]
    got = m16.evapotranspiration_raw(table, cls, *raw32, fpar, lai, math=m16._lib.MATH_FAST)
    for g, w, what in zip(got, want, ('day', 'night')):
        assert_parity(g, w, 1e-6, what + ', float32 rasters')
    got = m16.evapotranspiration_raw(table, cls, *raw32, fpar, lai, math=m16._lib.MATH_MIXED)
    for g, w, what in zip(got, want, ('day', 'night')):
        assert_mixed_parity(g, w, what + ', mixed')


def test_special_value_pairs_fast_kernel(m16, golden):
    """TWO special values in two different drivers of a pixel (an infinity next to a NaN, to a zero,
    to the albedo 1 that turns inf * (1 - albedo) into NaN, 1e-300 next to 3.4e38 ...): the guard was
    drawn from single values, it has to hold for combinations -- totals and the six components
    against the oracle, masks identical, values to 1e-8. (tests/fuzz_domain.py runs 1.2 M pairs; the
    first version of the guard failed 242 of 400 k: a pressure of 1e-300 and sw_rad_night = 1e300
    next to large values.)"""
    values = np.array([0.0, -0.0, np.nan, np.inf, -np.inf, -9999.0, 65535.0, 1e15, 3.4e38, -3.4e38, 1e300,
                       -1e300, 1e-300, -1e-300, 1e-7, 1.0, -1.0, 35.85, 34.15, 1400.0, 1e40, 1e49, 1e60, 1e150])
    rng = np.random.default_rng(77)
    n = 300000
    cls, drv, _ = _special_value_rasters([1.0], per=(n + 13) // 14)
    cls, drv = cls[:n], [d[:n].copy() for d in drv]
    a = rng.integers(0, 14, n)
    b = (a + rng.integers(1, 14, n)) % 14
    va, vb = values[rng.integers(0, len(values), n)], values[rng.integers(0, len(values), n)]
    for k in range(14):
        drv[k][a == k] = va[a == k]
        drv[k][b == k] = vb[b == k]
    table = golden('f3_random64_f64')['table']
    bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    with np.errstate(all='ignore'):
        want = oracle.evapotranspiration_raster(bplut, cls, *drv)
        want6 = oracle.evapotranspiration_raster(bplut, cls, *drv, separate=True)
    got = m16.evapotranspiration_raster(table, cls, *drv)
    assert_parity(got[0], want[0], 1e-8, 'day')
    assert_parity(got[1], want[1], 1e-8, 'night')
    got6 = m16.evapotranspiration_raster(table, cls, *drv, separate=True)
    for g3, w3, period in zip(got6, want6, ('day', 'night')):
        for g, w, part in zip(g3, w3, ('canopy', 'soil', 'transpiration')):
            assert_parity(g, w, 1e-8, period + ' ' + part)


@pytest.mark.parametrize('n', [1, 2, 3, 5, 365, 1025, 65535, 65536, 65537])
def test_small_calls_give_the_bits_of_the_staged_path(m16, golden, n):
    """HOST mode, calls of up to 65536 pixels: no copy commands -- the kernel reads its inputs
    from a page-locked buffer and writes its outputs there (run_host_small, mod16_capi.hip).
    Same kernels, same values: every output of every form equals the staged path's
    (MOD16_SMALL_PIXELS=0) bit for bit, float64 and float32, class raster and parameter
    arrays, ragged sizes (the buffer pads to whole vectors), one pixel above the limit."""
    f = golden('f3_random64_f64')
    table = f['table']
    cls, drv = synth.drivers((n,), seed=1000 + n)
    params = [table[cls.astype(int) % 13, j] if j % 2 else float(table[7, j]) for j in range(11)]
    def run(dtype):
        d = [a.astype(dtype) for a in drv]
        model = m16.MOD16(dict(zip(oracle.PARAM_NAMES, [np.asarray(p, dtype) if np.ndim(p) else p for p in params])))
        scal = list(d)
        scal[7] = float(d[7][0])        # a broadcast scalar among the dense drivers
        out = list(m16.evapotranspiration_raster(table, cls, *d))
        out += list(m16.evapotranspiration_raster(table, cls, *scal))
        sep = m16.evapotranspiration_raster(table, cls, *d, separate=True)
        out += list(sep[0]) + list(sep[1])
        out += list(model.evapotranspiration(*d))
        sep = model.evapotranspiration(*scal, separate=True)
        out += list(sep[0]) + list(sep[1])
        out += list(model.evapotranspiration_and_pet(*d))
        out.append(model.evaporation_soil(d[11], d[5], d[9], d[12], np.abs(d[2])))
        out.append(m16.MOD16.rhumidity(d[5], d[9]))
        # the calibration interface (mod16_et_static_*: its own small path, two kernels)
        plist = [np.asarray(p, dtype) if np.ndim(p) else p for p in params]
        out += list(m16.MOD16._evapotranspiration(plist, *d))
        return out

    for dtype in (np.float64, np.float32):
        small = run(dtype)
        staged = in_a_fresh_thread(lambda: run(dtype), {'MOD16_SMALL_PIXELS': '0'})
        assert len(small) == len(staged) == 26
        for i, (a, b) in enumerate(zip(small, staged)):
            assert a.dtype == dtype and same_bits(a, b), (n, dtype, i)
    # and the oracle, value by value (the staged path's own tests hold it elsewhere)
    bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    wd, wn = oracle.evapotranspiration_raster(bplut, cls, *drv)
    small_f64 = m16.evapotranspiration_raster(table, cls, *drv)[0]
    assert_parity(small_f64, wd, RTOL['fast'], 'day')
    # a class code numpy would refuse: IndexError from both paths, nothing left behind
    bad = cls.copy()
    bad[n // 2] = 13
    for env in ({}, {'MOD16_SMALL_PIXELS': '0'}):
        with pytest.raises(IndexError):
            in_a_fresh_thread(lambda: m16.evapotranspiration_raster(table, bad, *drv), env)
    assert same_bits(m16.evapotranspiration_raster(table, cls, *drv)[0], small_f64)


def test_scalar_site_call_through_the_small_path(m16, golden):
    """BASELINE.json configs[0]: the flux-tower scalars of the reference's tests (F1) -- numpy
    scalars out, the reference's values, and the same again after calls of other sizes and
    dtypes have reshaped the page-locked buffer."""
    f = golden('f1_tests_scalars')
    m = model(m16, [float(p) for p in f['params']], 'fast')     # (Python floats: weak, as in numpy)
    drivers = [float(x) for x in f['drivers']]
    for _ in range(2):
        day, night = m.evapotranspiration(*drivers)
        assert np.ndim(day) == 0 and isinstance(day, np.floating)
        assert_parity(np.asarray(day), f['day'], RTOL['fast'], 'day')
        assert_parity(np.asarray(night), f['night'], RTOL['fast'], 'night')
        big = [np.full(3000, x, np.float32) for x in drivers]
        d32, _ = m.evapotranspiration(*big)
        assert d32.dtype == np.float32 and d32.shape == (3000,)
        assert abs(float(d32[0]) / float(day) - 1) < 1e-6


def test_small_calls_from_several_threads(m16, golden):
    """The reference's functions are pure and may be called from several threads (SURVEY.md
    section 8b): every thread has its own context -- and its own page-locked buffer for small
    calls. Four threads, interleaved scalar / site-year / window calls with their own inputs:
    each result equals the one the main thread computes for those inputs."""
    import threading
    f = golden('f1_tests_scalars')
    m = model(m16, [float(p) for p in f['params']], 'fast')
    base = [float(x) for x in f['drivers']]

    def inputs(seed, shape):
        rng = np.random.default_rng(seed)
        return [b * (1 + 0.01 * rng.uniform(-1, 1, shape)) if shape else b * (1 + 0.001 * seed) for b in base]
    cases = [(seed, shape) for seed in range(12) for shape in ((), (365,), (40, 50))]
    want = {c: m.evapotranspiration(*inputs(*c)) for c in cases}
    errors = []

    def worker(k):
        try:
            for rep in range(20):
                for c in cases[k::4]:
                    got = m.evapotranspiration(*inputs(*c))
                    for g, w in zip(got, want[c]):
                        if not same_bits(g, w):
                            errors.append((k, c))
        except BaseException as exc:      # noqa: BLE001 -- reported below
            errors.append((k, repr(exc)))
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]


@pytest.mark.parametrize('dtype', ['float64', 'float32'])
def test_device_tensors_through_the_class(m16, golden, dtype):
    """MOD16.evapotranspiration on torch tensors that live on the GPU (extension): DEVICE mode of
    the same entry points, torch tensors back -- the numpy call's bits for the same values, with
    the reference's shapes: scalars among the tensors, a (1, N) row against (T, N), a tensor
    parameter, separate components, potential ET, a side stream."""
    import torch
    f = golden('f1_tests_scalars')
    T, N = 37, 211
    rng = np.random.default_rng(5)
    np_dtype = np.dtype(dtype)
    host = [np.asarray(float(x) * (1 + 0.02 * rng.uniform(-1, 1, (T, N))), np_dtype) for x in f['drivers']]
    host[7] = np.ascontiguousarray(host[7][:1])              # temp_annual: a (1, N) row
    host[11] = float(f['drivers'][11])                       # pressure: a Python number
    params = [float(p) for p in f['params']]
    m = model(m16, params, 'fast')
    dev = [torch.from_numpy(a).cuda() if isinstance(a, np.ndarray) else a for a in host]
    want = m.evapotranspiration(*host)
    got = m.evapotranspiration(*dev)
    torch.cuda.synchronize()
    for g, w in zip(got, want):
        assert g.is_cuda and g.dtype == getattr(torch, dtype) and tuple(g.shape) == (T, N)
        assert same_bits(g.cpu().numpy(), w)
    ws = m.evapotranspiration(*host, separate=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gs = m.evapotranspiration(*dev, separate=True)
        gp = m.evapotranspiration_and_pet(*dev)
    side.synchronize()
    for g, w in zip(list(gs[0]) + list(gs[1]), list(ws[0]) + list(ws[1])):
        assert same_bits(g.cpu().numpy(), w)
    for g, w in zip(gp, m.evapotranspiration_and_pet(*host)):
        assert same_bits(g.cpu().numpy(), w)
    # a per-pixel parameter as a device tensor, a 0-d tensor among the drivers
    csl = np.asarray(params[7] * (1 + 0.1 * rng.uniform(-1, 1, (T, N))), np_dtype)
    pt = dict(zip(oracle.PARAM_NAMES, params))
    m_host, m_dev = m16.MOD16(dict(pt, csl=csl)), m16.MOD16(dict(pt, csl=torch.from_numpy(csl).cuda()))
    dev0 = list(dev)
    dev0[11] = torch.tensor(host[11], dtype=getattr(torch, dtype), device='cuda')
    for g, w in zip(m_dev.evapotranspiration(*dev0), m_host.evapotranspiration(*host)):
        assert same_bits(g.cpu().numpy(), w)
    # fourteen dense drivers and one set of parameters (the reference's usual call): the production
    # pipeline behind a class raster of ones -- still the numpy call's bits; then other parameters
    dense_h = list(host)
    dense_h[7] = np.ascontiguousarray(np.broadcast_to(host[7], (T, N)))
    dense_h[11] = np.full((T, N), host[11], np_dtype)
    dense_d = [torch.from_numpy(a).cuda() for a in dense_h]
    for mm in (m, model(m16, [p * 1.01 for p in params], 'fast'), m):
        for g, w in zip(mm.evapotranspiration(*dense_d), mm.evapotranspiration(*dense_h)):
            assert same_bits(g.cpu().numpy(), w)
    gs, ws = m.evapotranspiration(*dense_d, separate=True), m.evapotranspiration(*dense_h, separate=True)
    for g, w in zip(list(gs[0]) + list(gs[1]), list(ws[0]) + list(ws[1])):
        assert same_bits(g.cpu().numpy(), w)
    for g, w in zip(m.evapotranspiration_and_pet(*dense_d), m.evapotranspiration_and_pet(*dense_h)):
        assert same_bits(g.cpu().numpy(), w)
    # ... and a LARGE raster with a scalar among its drivers: the scalar is written out for the pipeline
    big = (1025, 1024)
    big_h = [np.asarray(float(x) * (1 + 0.02 * rng.uniform(-1, 1, big)), np_dtype) for x in f['drivers']]
    big_h[11] = float(f['drivers'][11])
    big_h[4] = np.asarray([[float(f['drivers'][4])]], np_dtype)          # a (1, 1) array: one value as well
    big_d = [torch.from_numpy(a).cuda() if isinstance(a, np.ndarray) else a for a in big_h]
    for g, w in zip(m.evapotranspiration(*big_d), m.evapotranspiration(*big_h)):
        assert tuple(g.shape) == big and same_bits(g.cpu().numpy(), w)
    # host arrays cannot be mixed in
    mixed = list(dev)
    mixed[0] = host[0]
    with pytest.raises(TypeError):
        m.evapotranspiration(*mixed)


def test_scalar_calls_return_what_the_reference_returns(m16, golden):
    """All-scalar input (the reference's tests call it that way, tests/tests.py:64-121): the
    totals (:792) and the soil component (e / lhv, :864) are numpy scalars, the wet-canopy and
    transpiration components -- np.where results, :961 and :1258 -- are 0-d arrays, as the
    sub-methods themselves return them."""
    f = golden('f1_tests_scalars')
    m = model(m16, [float(p) for p in f['params']], 'fast')
    drivers = [float(x) for x in f['drivers']]
    day, night = m.evapotranspiration(*drivers)
    assert isinstance(day, np.float64) and isinstance(night, np.float64)
    for period in m.evapotranspiration(*drivers, separate=True):
        canopy, soil, trans = period
        assert isinstance(canopy, np.ndarray) and canopy.shape == () and canopy.dtype == np.float64
        assert isinstance(trans, np.ndarray) and trans.shape == ()
        assert isinstance(soil, np.float64)
        assert float((canopy + soil) + trans) in (float(day), float(night))


def test_a_loop_over_parameter_sets_on_device_tensors(m16, golden):
    """The reference's usual pattern on device tensors -- one MOD16 per plant functional type,
    called in a loop -- with every call ASYNCHRONOUS on a side stream and the numpy entry point (which
    sets the table of the thread's own context with a blocking copy) called in between: every
    parameter set has a context of its own whose table is written once (a table
    changing under a launch in flight gave silently wrong parameters), so each result is the
    numpy call's, bit for bit, and the second pass over the sets makes no new context."""
    import torch
    import mod16_amd
    from mod16_amd.utils import restore_bplut
    from mod16_amd.models import COLLECTION61_BPLUT
    f = golden('f1_tests_scalars')
    bplut = restore_bplut(COLLECTION61_BPLUT)
    rng = np.random.default_rng(9)
    shape = (1100, 1000)                    # > 2^20 pixels: the pipeline behind a class raster of ones
    host = [np.asarray(float(x) * (1 + 0.02 * rng.uniform(-1, 1, shape))) for x in f['drivers']]
    dev = [torch.from_numpy(a).cuda() for a in host]
    pfts = [1, 4, 7, 10, 12]
    models = [m16.MOD16(dict({k: float(bplut[k][c]) for k in oracle.PARAM_NAMES if k != 'beta'}, beta=250.0))
              for c in pfts]
    want = [mm.evapotranspiration(*host) for mm in models]
    mod16_amd.release_device_cache()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    got = []
    for rounds in range(2):
        for mm in models:
            with torch.cuda.stream(side):
                got.append(mm.evapotranspiration(*dev))
            # the numpy path of this thread, with another table, while the launch may still run
            mm.evapotranspiration(*[a[:7, :9] for a in host])
        if rounds == 0:
            made = len(mod16_amd._tensor_local.contexts)
    assert made == len(pfts) == len(mod16_amd._tensor_local.contexts)
    side.synchronize()
    for k, res in enumerate(got):
        for g, w in zip(res, want[k % len(pfts)]):
            assert same_bits(g.cpu().numpy(), w), k
    mod16_amd.release_device_cache()
    assert not mod16_amd._tensor_local.contexts


@pytest.mark.parametrize('dtype', ['float64', 'float32'])
def test_gathered_parameter_tensors_take_the_pipeline(m16, dtype):
    """The reference's multi-class idiom on device tensors (notebook cell 32):
    MOD16({k: bplut[k][pft_map]}).evapotranspiration(*tensors) -- eleven parameter rasters that are a
    gather of the table's rows (all 11 valid classes + the two invalid ones, whose rows are NaN) are
    recognised (mod16_classify_*), turned into a class raster on the device and run through the
    production pipeline: the numpy call's bits (which are the plain kernel's, per-pixel parameters).
    The recognition is kept with the model for as long as the tensors are unchanged; genuinely
    per-pixel parameters (more than 13 distinct rows) keep taking the plain kernel."""
    import torch
    import mod16_amd
    from mod16_amd.utils import restore_bplut, bplut_table
    from mod16_amd.models import COLLECTION61_BPLUT
    np_dtype = np.dtype(dtype)
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    shape = (1030, 1024)                                    # > 2^20 pixels
    cls, drv = synth.drivers(shape, seed=21, dtype=np_dtype)
    cls[5, 7:9] = [0, 11]                                   # the invalid classes are there
    cls[cls == 3] = 4
    cls[1000, 1000] = 3                                     # ... and class 3 only where the sample does not look
    names = mod16_amd.MOD16.required_parameters
    par_h = {k: table[:, j][cls].astype(np_dtype) for j, k in enumerate(names)}
    par_d = {k: torch.from_numpy(v).cuda() for k, v in par_h.items()}
    par_d['beta'] = 250.0                                   # one of them a plain number
    dev = [torch.from_numpy(a).cuda() for a in drv]
    m_host, m_dev = m16.MOD16(par_h), m16.MOD16(par_d)
    want = m_host.evapotranspiration(*drv)
    got = m_dev.evapotranspiration(*dev)
    torch.cuda.synchronize()
    answer = m_dev._gather_cache['answer']
    assert answer is not None and answer[1].dtype == torch.uint8
    found = answer[0][:, 10] == 250.0                        # (rows in use: beta is a number in every one)
    assert np.isfinite(answer[0][found, 0]).sum() == 11 and np.isnan(answer[0][found, 0]).sum() == 1   # 11 classes + the NaN row
    for g, w in zip(got, want):
        assert g.dtype == getattr(torch, dtype) and same_bits(g.cpu().numpy(), w)
    # the class raster the library made: every pixel's row IS its parameters
    back = answer[0][answer[1].cpu().numpy().reshape(shape)]
    assert same_bits(back[..., 7].astype(np_dtype), par_h['csl'])
    # again: the recognition is reused (same objects, unchanged) -- and redone after an in-place change
    first = answer[1].data_ptr()
    for g, w in zip(m_dev.evapotranspiration(*dev, separate=True)[0], m_host.evapotranspiration(*drv, separate=True)[0]):
        assert same_bits(g.cpu().numpy(), w)
    assert m_dev._gather_cache['answer'][1].data_ptr() == first
    for j, f in ((0, 1.5), (1, 1.7), (2, 1.9)):               # now the raster holds more than 13 rows
        par_d['csl'][0, j] *= f
        par_h['csl'][0, j] = par_d['csl'][0, j].item()
    got = m_dev.evapotranspiration(*dev)
    torch.cuda.synchronize()
    assert m_dev._gather_cache['answer'] is None             # not a gather any more: the plain kernel
    for g, w in zip(got, m_host.evapotranspiration(*drv)):
        assert same_bits(g.cpu().numpy(), w)
    mod16_amd.release_device_cache()
