"""GPU tests of the mixed-precision form for float32 rasters (MOD16_MATH_MIXED,
mod16_amd/csrc/mod16_mixed.hpp; BASELINE.json configs[4]): float64 where a
mask is decided and for the humidity terms, packed float32 elsewhere.

Checked against the float64 arithmetic on the same float32 inputs (the FAST
kernel, which is itself within 1e-8 of the oracle): identical NaN masks,
identical exact-zero masks on these rasters, median relative error < 2e-7,
99 % of the pixels < 3e-6, absolute error < 2e-6 of the largest value. The
relative error of the remaining pixels is not bounded by 1e-5: where
s*A + rho*Cp*vpd/r_a cancels (A < 0) float32 factors resolve the sum to
1e-7 * |s*A| only; those values are orders of magnitude below the typical one."""
import numpy as np
import pytest

from oracle import mod16_oracle as oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    from mod16_amd import _lib
    from mod16_amd.raster import RasterEngine
    from mod16_amd.utils import restore_bplut, bplut_table
    from mod16_amd.models import COLLECTION61_BPLUT
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    return torch, _lib, RasterEngine, table


def stats(got, want):
    got = got.astype(np.float64)
    want = want.astype(np.float64)
    assert np.array_equal(np.isnan(got), np.isnan(want)), 'NaN masks differ'
    zero_mismatch = int(((got == 0) != (want.astype(np.float32) == 0)).sum())
    m = np.isfinite(want) & (want != 0)
    rel = np.abs(got[m] - want[m]) / np.abs(want[m])
    scale = np.nanmax(np.abs(want))
    return {'zero_mismatch': zero_mismatch, 'median': float(np.median(rel)),
            'p99': float(np.percentile(rel, 99)), 'max': float(rel.max()),
            'abs_over_scale': float(np.nanmax(np.abs(got - want)) / scale)}


def test_mixed_against_float64_arithmetic(env):
    torch, _lib, RasterEngine, table = env
    n = 1200 * 1200 + 8
    mixed = RasterEngine(table, dtype='float32', math=_lib.MATH_MIXED)
    fast64 = RasterEngine(table, dtype='float64')
    cls, drv32 = mixed.synth(n, seed=51)
    d_m, n_m = mixed.run(cls, drv32)
    d_f, n_f = fast64.run(cls, [d.double() for d in drv32])
    mixed.check()
    fast64.check()
    for got, want, what in ((d_m, d_f, 'day'), (n_m, n_f, 'night')):
        s = stats(got.cpu().numpy(), want.cpu().numpy())
        assert s['zero_mismatch'] == 0, (what, s)
        assert s['median'] < 2e-7 and s['p99'] < 3e-6 and s['abs_over_scale'] < 2e-6, (what, s)
    # and against the oracle on a tile of it
    m = 300000
    bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    h_drv = [d[:m].cpu().numpy().astype(np.float64) for d in drv32]
    wd, wn = oracle.evapotranspiration_raster(bplut, cls[:m].cpu().numpy(), *h_drv)
    for got, want in ((d_m[:m], wd), (n_m[:m], wn)):
        s = stats(got.cpu().numpy(), want)
        assert s['zero_mismatch'] == 0 and s['median'] < 2e-7 and s['abs_over_scale'] < 2e-6, s


def test_mixed_edge_cases_keep_their_masks(env):
    """The reference's edge cases (F4: zero / NaN / saturated drivers, fpar 0 and
    1, lai 0, vpd <= 0 and beyond saturation, pressure 0 ...) as a class raster:
    every NaN and every exact zero of the reference is one here."""
    torch, _lib, RasterEngine, table = env
    import mod16_amd
    import os
    from conftest import GOLDEN
    f = np.load(os.path.join(GOLDEN, 'f4_edge_cases.npz'))
    reps = 64                                   # whole pieces: the raster takes the pipeline
    drv = [np.tile(d.astype(np.float32), reps) for d in f['drivers']]
    t = np.full((13, 11), np.nan)
    t[7] = f['params']
    cls = np.full(drv[0].shape, 7, np.uint8)
    day, night = mod16_amd.evapotranspiration_raster(t, cls, *drv, math=_lib.MATH_MIXED)
    d64, n64 = mod16_amd.evapotranspiration_raster(t, cls, *[d.astype(np.float64) for d in drv])
    assert day.dtype == np.float32
    for got, want in ((day, d64), (night, n64)):
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert np.array_equal(got == 0, want.astype(np.float32) == 0)
        ok = np.isfinite(want) & (want != 0)
        rel = np.abs(got[ok] - want[ok]) / np.abs(want[ok])
        assert rel.max() < 5e-6, rel.max()


def test_mixed_components_and_potential_et(env):
    """The component and potential-ET forms of the pipeline with the mixed arithmetic,
    against the float64 arithmetic on the same float32 rasters."""
    torch, _lib, RasterEngine, table = env
    n = 1200 * 900 + 4
    mixed = RasterEngine(table, dtype='float32', math=_lib.MATH_MIXED)
    fast64 = RasterEngine(table, dtype='float64')
    cls, drv32 = mixed.synth(n, seed=54)
    drv64 = [d.double() for d in drv32]
    sep_m, sep_f = mixed.empty(n, 6), fast64.empty(n, 6)
    d_m, n_m = mixed.empty(n, 2)
    mixed.run(cls, drv32, d_m, n_m, out_sep=sep_m)            # totals + six components
    fast64.run(cls, drv64, out_sep=sep_f)                    # six components
    pet_m = mixed.run_pet(cls, drv32)
    pet_f = fast64.run_pet(cls, drv64)
    mixed.check()
    fast64.check()
    for got, want in list(zip(sep_m, sep_f)) + list(zip(pet_m, pet_f)) + [(d_m, pet_f[0]), (n_m, pet_f[1])]:
        s = stats(got.cpu().numpy(), want.cpu().numpy())
        assert s['zero_mismatch'] == 0, s
        # single components carry the cancellation tail undiluted: 99 % within 2e-5
        assert s['median'] < 3e-7 and s['p99'] < 2e-5 and s['abs_over_scale'] < 2e-6, s
    # the totals of these forms are the totals of the totals form (same arithmetic; the
    # compiler may fuse multiply-adds differently per instantiation: a float32 ulp or two)
    d0, n0 = mixed.run(cls, drv32)
    for got, want in ((d_m, d0), (n_m, n0), (pet_m[0], d0), (pet_m[1], n0)):
        s = stats(got.cpu().numpy(), want.cpu().numpy())
        assert s['zero_mismatch'] == 0 and s['median'] < 1e-7 and s['abs_over_scale'] < 5e-7, s


@pytest.mark.parametrize('hours_kind', ['none', 'array', 'scalar'])
def test_mixed_raw_drivers(env, hours_kind):
    """Raw float32 drivers (specific humidity, surface pressure, elevation, uint8 fPAR /
    LAI with fill codes) through the mixed form against the float64 arithmetic."""
    torch, _lib, RasterEngine, table = env
    from test_gpu_stream import raw_inputs
    n = 64 * 8 * 2 * 37 * 4 + 64 * 5 + 3
    cls, raw, fpar, lai, hours = raw_inputs(n, 55, np.float32)
    dev = lambda a: torch.from_numpy(a).cuda()
    mixed = RasterEngine(table, dtype='float32', math=_lib.MATH_MIXED)
    fast64 = RasterEngine(table, dtype='float64')
    h32 = {'none': None, 'array': dev(hours), 'scalar': 11.5}[hours_kind]
    h64 = {'none': None, 'array': dev(hours.astype(np.float64)), 'scalar': 11.5}[hours_kind]
    got = mixed.run_raw(dev(cls), [dev(a) for a in raw], dev(fpar), dev(lai), day_hours=h32)
    want = fast64.run_raw(dev(cls), [dev(a.astype(np.float64)) for a in raw], dev(fpar), dev(lai), day_hours=h64)
    mixed.check()
    fast64.check()
    assert len(got) == len(want) == (2 if hours_kind == 'none' else 3)
    for g, w in zip(got, want):
        s = stats(g.cpu().numpy(), w.cpu().numpy())
        assert s['zero_mismatch'] == 0, s
        assert s['median'] < 3e-7 and s['p99'] < 5e-6 and s['abs_over_scale'] < 2e-6, s


def test_mixed_falls_back_to_fast_elsewhere(env):
    """float64 rasters run FAST."""
    torch, _lib, RasterEngine, table = env
    n = 100000
    e_m = RasterEngine(table, dtype='float64', math=_lib.MATH_MIXED)
    e_f = RasterEngine(table, dtype='float64')
    cls, drv = e_f.synth(n, seed=52)
    a, b = e_m.run(cls, drv)
    c, d = e_f.run(cls, drv)
    assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(c))
    assert torch.equal(torch.nan_to_num(b), torch.nan_to_num(d))


def test_mixed_diagnostics_and_class_range(env):
    torch, _lib, RasterEngine, table = env
    eng = RasterEngine(table, dtype='float32', math=_lib.MATH_MIXED)
    n = 500000
    cls, drv = eng.synth(n, seed=53)
    diag = torch.zeros(8, dtype=torch.float64, device='cuda')
    day, night = eng.run(cls, drv, diag=diag)
    want = eng.diagnostics(day, night)
    eng.check()
    d, w = diag.cpu().numpy(), want.cpu().numpy()
    assert np.array_equal(d[2:], w[2:])                       # counts and maxima exactly
    np.testing.assert_allclose(d[:2], w[:2], rtol=1e-12)      # sums: another (fixed) order
    cls[777] = 99
    eng.run(cls, drv)
    with pytest.raises(IndexError):
        eng.check()


def test_mixed_at_piece_and_run_boundaries(env):
    """The mixed form at sizes around one vector, one piece, one run (work distribution
    and the scalar tail, which runs FAST): masks and absolute error against FAST."""
    torch, _lib, RasterEngine, table = env
    mixed = RasterEngine(table, dtype='float32', math=_lib.MATH_MIXED)
    fast = RasterEngine(table, dtype='float32')
    piece = 256
    sizes = [4, piece - 4, piece, piece + 1, 2 * piece, 8 * piece - 1, 8 * piece, 8 * piece + 5, 17 * piece + 3]
    cls, drv = mixed.synth(max(sizes), seed=56)
    for n in sizes:
        c, d = cls[:n], [x[:n] for x in drv]
        for got, want in zip(mixed.run(c, d), fast.run(c, d)):
            g, w = got.cpu().numpy(), want.cpu().numpy()
            assert np.array_equal(np.isnan(g), np.isnan(w)), n
            assert np.array_equal(g == 0, w == 0), n
            scale = np.nanmax(np.abs(w))
            assert np.nanmax(np.abs(g.astype(np.float64) - w)) <= 2e-6 * scale, n
    mixed.check()
    fast.check()


def test_mixed_special_values_keep_their_masks(env):
    """Zeros, signed zeros, NaN, the fill values (-9999, 65535, 1e15, +-3.4e38), infinities, 1,
    1e-7, the pole of the Tetens formula -- every one of them in every driver, nothing masked out:
    the mixed form's NaN, zero and inf masks are those of the float64 arithmetic (whose own are the
    oracle's: test_gpu_parity.py::test_special_values_float32_rasters). The mixed arithmetic
    itself covers physical drivers only; a pixel outside that domain (mod16_mixed.hpp, "domain
    guard") is computed in the reference's operation order in the same kernel."""
    import mod16_amd as m16
    from test_gpu_parity import _special_value_rasters, SPECIAL_VALUES
    torch, _lib, RasterEngine, table = env
    values = [v for v in SPECIAL_VALUES if not np.isfinite(v) or v == 0 or 1e-37 < abs(v) < 3.41e38]
    cls, drv, which = _special_value_rasters(values)
    drv = [d.astype(np.float32) for d in drv]
    fast = m16.evapotranspiration_raster(table, cls, *drv, math=_lib.MATH_FAST)
    mixed = m16.evapotranspiration_raster(table, cls, *drv, math=_lib.MATH_MIXED)
    for g, w, what in zip(mixed, fast, ('day', 'night')):
        assert np.array_equal(np.isnan(g), np.isnan(w)), what
        assert np.array_equal(g == 0, w == 0), what
        assert np.array_equal(np.isinf(g), np.isinf(w)), what


def _err_stats(got, want):
    """median, 99th percentile and maximum of |got - want| / |want| over want finite and != 0"""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    ok = np.isfinite(want) & (want != 0)
    e = np.abs(got[ok] - want[ok]) / np.abs(want[ok])
    return float(np.median(e)), float(np.percentile(e, 99)), float(e.max())


@pytest.mark.parametrize('math', ['mixed', 'fast'])
def test_f5_reference_float32_run(env, math):
    """configs[4] against the reference's OWN float32 vectors (fixture F5 = the reference run on
    all-float32 inputs, tests/golden/make_golden.py: mod16/__init__.py:675-793 in float32): the
    float32 kernels on F5's drivers through evapotranspiration_raster --
      * NaN and exact-zero masks identical to the reference's float32 outputs;
      * against the float64 fixture F3 (the same field generated in float64): no worse than the
        reference's own float32 run at the median, the 99th percentile and the maximum;
      * against the float64 arithmetic on F5's own inputs (the oracle on the widened drivers, which
        takes the inputs' rounding out of the comparison): likewise no worse than numpy's float32."""
    torch, _lib, RasterEngine, table = env
    import mod16_amd
    import os
    from conftest import GOLDEN
    f5 = np.load(os.path.join(GOLDEN, 'f5_random64_f32.npz'))
    f3 = np.load(os.path.join(GOLDEN, 'f3_random64_f64.npz'))
    flag = {'mixed': _lib.MATH_MIXED, 'fast': _lib.MATH_FAST}[math]
    drv = list(f5['drivers'])
    got = mod16_amd.evapotranspiration_raster(f5['table'], f5['cls'], *drv, math=flag)
    bplut = {k: f5['table'][:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    wide = oracle.evapotranspiration_raster(bplut, f5['cls'], *[d.astype(np.float64) for d in drv])
    for g, name, w64 in zip(got, ('day', 'night'), wide):
        ref32 = f5[name]
        assert g.dtype == np.float32 and g.shape == ref32.shape
        assert np.array_equal(np.isnan(g), np.isnan(ref32)), name
        assert np.array_equal(g == 0, ref32 == 0), name
        for truth, what in ((f3[name], 'F3'), (w64, 'float64 arithmetic on F5 inputs')):
            mine, theirs = _err_stats(g, truth), _err_stats(ref32, truth)
            print('\n[f5 %s %s vs %s] median / p99 / max: kernel %.2e %.2e %.2e, numpy float32 %.2e %.2e %.2e'
                  % (math, name, what, *mine, *theirs))
            if what == 'F3':
                # F5's drivers are the same field GENERATED in float32, not F3's rounded: both float32
                # runs differ from F3 mostly by their inputs, and the largest difference is one pixel's
                # input rounding through a cancellation (the exact float64 arithmetic on F5's inputs is
                # 1.8e-4 off F3 there, numpy's float32 happens to land closer): median and 99th
                # percentile no worse than numpy's (2 % for statistics of 4096 pixels), maximum within 3 x
                assert mine[0] <= 1.02 * theirs[0] and mine[1] <= 1.02 * theirs[1] and mine[2] <= 3 * theirs[2], \
                    (math, name, what, mine, theirs)
            else:
                assert all(a <= b for a, b in zip(mine, theirs)), (math, name, what, mine, theirs)


def test_class_surface_takes_the_mixed_form_on_float32_tensors(env):
    """``MOD16(params)`` with ``model.math = MATH_MIXED`` on float32 tensors resident on the GPU: the
    class surface runs the mixed-precision pipeline instance (1.3x faster than the float64 arithmetic
    on the same tensors) -- same NaN masks, the mixed form's accuracy; float64 tensors ignore the flag
    (same bits as FAST)."""
    torch, _lib, RasterEngine, table = env
    import mod16_amd
    n = 3 * 4096 * 1024 + 4 * 37          # dynamic schedule, ragged end
    eng = RasterEngine(table, dtype='float32')
    _, drv = eng.synth(n, seed=23)
    params = dict(zip(mod16_amd.MOD16.required_parameters, (float(v) for v in table[7])))
    fast, mixed = mod16_amd.MOD16(params), mod16_amd.MOD16(params)
    mixed.math = _lib.MATH_MIXED
    want = fast.evapotranspiration(*drv)
    got = mixed.evapotranspiration(*drv)
    torch.cuda.synchronize()
    for g, w, what in zip(got, want, ('day', 'night')):
        assert g.dtype == torch.float32 and g.shape == w.shape
        assert torch.equal(torch.isnan(g), torch.isnan(w)), what
        rel = torch.nan_to_num((g.double() - w.double()).abs() / w.double().abs(), nan=0.0, posinf=0.0)
        assert float(rel.max()) < 1e-3, (what, float(rel.max()))
        assert float(rel.median()) < 2e-7, (what, float(rel.median()))
        assert not torch.equal(g, w), 'the flag did not reach the kernel'
    drv64 = [d.double() for d in drv[:14]]
    a = fast.evapotranspiration(*drv64)
    b = mixed.evapotranspiration(*drv64)
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert torch.equal(x.view(torch.int64), y.view(torch.int64))
