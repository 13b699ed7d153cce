"""Tiled rasters (mod16_layout, RasterEngine.alloc_tiled): the engine's own
layout for device-resident rasters. Same pixels, same arithmetic, same schedule
as the plain-array path -- so everything must agree with it bit for bit -- and
the oracle is checked on windows copied back from the tiled storage."""
import numpy as np
import pytest

from oracle import mod16_oracle as oracle
from parity import assert_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    from mod16_amd import _lib
    from mod16_amd.raster import RasterEngine
    from mod16_amd.utils import restore_bplut, bplut_table
    from mod16_amd.models import COLLECTION61_BPLUT
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    return torch, RasterEngine, table, _lib


def same(a, b):
    import torch
    return torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))


@pytest.mark.parametrize('dtype,math', [('float64', 'fast'), ('float32', 'fast'), ('float32', 'mixed')])
@pytest.mark.parametrize('n', [8192 * 40, 8192 * 40 + 4 * 777, 4 * 500, 16384 * 3, 1200 * 1200])
def test_tiled_run_equals_plain_run(env, dtype, math, n):
    """Generator, forward run and in-kernel diagnostics on a tiled raster against
    the same on plain arrays: identical bits (sizes: whole tiles, ragged last
    tile, less than one tile, one 1200 x 1200 tile of configs[1])."""
    torch, RasterEngine, table, _lib = env
    m = {'fast': _lib.MATH_FAST, 'mixed': _lib.MATH_MIXED}[math]
    eng = RasterEngine(table, dtype=dtype, math=m)
    cls, drv = eng.synth(n, seed=5, step=2, pixel_offset=1000)
    d_plain = torch.zeros(8, dtype=torch.float64, device='cuda')
    day, night = eng.run(cls, drv, diag=d_plain)
    r = eng.alloc_tiled(n)
    r.slab.fill_(0xff)
    eng.synth_tiled(r, seed=5, step=2, pixel_offset=1000)
    assert torch.equal(r.flat(r.cls), cls)
    for k in range(14):
        assert same(r.flat(r.drivers[k]), drv[k]), k
    d_tiled = torch.zeros(8, dtype=torch.float64, device='cuda')
    eng.run_tiled(r, diag=d_tiled)
    eng.check()
    assert same(r.flat(r.day), day) and same(r.flat(r.night), night)
    assert torch.equal(d_tiled, d_plain)
    # without diagnostics, and through to_tiled (strided copies of plain arrays)
    r2 = eng.to_tiled(cls, drv)
    eng.run_tiled(r2)
    eng.check()
    assert same(r2.flat(r2.day), day) and same(r2.flat(r2.night), night)


def test_tiled_raster_against_the_oracle(env):
    torch, RasterEngine, table, _lib = env
    eng = RasterEngine(table)
    n = 1200 * 1200 + 8192 * 3
    r = eng.synth_tiled(eng.alloc_tiled(n), seed=77)
    diag = torch.zeros(8, dtype=torch.float64, device='cuda')
    eng.run_tiled(r, diag=diag)
    eng.check()
    bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    lo, hi = 5000, 5000 + 1200 * 1200
    h_drv = [r.flat(d, lo, hi).cpu().numpy() for d in r.drivers]
    want = oracle.evapotranspiration_raster(bplut, r.flat(r.cls, lo, hi).cpu().numpy(), *h_drv)
    assert_parity(r.flat(r.day, lo, hi).cpu().numpy(), want[0], 1e-8, 'day')
    assert_parity(r.flat(r.night, lo, hi).cpu().numpy(), want[1], 1e-8, 'night')
    hd, hn = r.flat(r.day).cpu().numpy(), r.flat(r.night).cpu().numpy()
    g = diag.cpu().numpy()
    np.testing.assert_allclose(g[:2], [np.nansum(hd), np.nansum(hn)], rtol=1e-12)
    assert g[4] == np.isnan(hd).sum() and g[2] == n - g[4]
    assert g[6] == np.nanmax(hd) and g[7] == np.nanmax(hn)


def test_bound_tiled_step_and_other_tile_sizes(env):
    torch, RasterEngine, table, _lib = env
    eng = RasterEngine(table)
    n = 700000
    ref = eng.synth_tiled(eng.alloc_tiled(n), seed=3)
    d0 = torch.zeros(8, dtype=torch.float64, device='cuda')
    eng.run_tiled(ref, diag=d0)
    for tile in (1024, 4096, 65536):
        r = eng.synth_tiled(eng.alloc_tiled(n, tile=tile), seed=3)
        d = torch.zeros(8, dtype=torch.float64, device='cuda')
        step = eng.bind_tiled(r, d)
        for _ in range(3):
            step()
        eng.check()
        assert same(r.flat(r.day), ref.flat(ref.day)) and same(r.flat(r.night), ref.flat(ref.night))
        assert torch.equal(d, d0), tile
        assert step.time(5) > 0
    with pytest.raises(ValueError):
        eng.alloc_tiled(n, tile=3000)
    with pytest.raises(ValueError):
        eng.alloc_tiled(n + 1)


def test_tiled_layout_argument_checks(env):
    torch, RasterEngine, table, _lib = env
    import ctypes as C
    eng = RasterEngine(table)
    r = eng.synth_tiled(eng.alloc_tiled(100000), seed=3)
    fn = eng.ctx.lib.mod16_et_tiled_f64
    ptrs = _lib.ptr_array([d.data_ptr() for d in r.drivers])
    bad = _lib.Layout(3000, 14 * 8192, 2 * 8192, 8192)
    assert fn(eng.ctx.handle, C.byref(bad), r.cls.data_ptr(), ptrs, r.n, r.day.data_ptr(),
              r.night.data_ptr(), 0, None, None) == _lib.ERR_ARG
    short = _lib.Layout(8192, 4096, 2 * 8192, 8192)
    assert fn(eng.ctx.handle, C.byref(short), r.cls.data_ptr(), ptrs, r.n, r.day.data_ptr(),
              r.night.data_ptr(), 0, None, None) == _lib.ERR_ARG
    assert fn(eng.ctx.handle, C.byref(r.layout), r.cls.data_ptr(), ptrs, r.n, r.day.data_ptr(),
              r.night.data_ptr(), _lib.MATH_EXACT, None, None) == _lib.ERR_ARG
    assert fn(eng.ctx.handle, C.byref(r.layout), r.cls.data_ptr(), ptrs, r.n, r.day.data_ptr() + 8,
              r.night.data_ptr(), 0, None, None) == _lib.ERR_ARG


def test_tiled_series_matches_the_oracle(env):
    """configs[3] on tiled rasters: 46 steps through a two-slot ring, windows of the
    first, a middle and the last step checked against the numpy oracle."""
    torch, RasterEngine, table, _lib = env
    eng = RasterEngine(table)
    n, steps = 600 * 1440, 46
    lo, hi = 123 * 1440 + 64, 123 * 1440 + 64 + 320000
    bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    grabbed = {}

    def grab(s, r):
        if s in (0, 23, steps - 1):
            grabbed[s] = ([r.flat(d, lo, hi) for d in r.drivers], r.flat(r.cls, lo, hi),
                          r.flat(r.day, lo, hi), r.flat(r.night, lo, hi))

    diag, last = eng.run_series_tiled(n, steps, seed=16, on_step=grab)
    eng.check()
    assert sorted(grabbed) == [0, 23, steps - 1]
    for s, (drv, cls, day, night) in grabbed.items():
        want = oracle.evapotranspiration_raster(bplut, cls.cpu().numpy(), *[d.cpu().numpy() for d in drv])
        assert_parity(day.cpu().numpy(), want[0], 1e-8, 'step %d day' % s)
        assert_parity(night.cpu().numpy(), want[1], 1e-8, 'step %d night' % s)
    # every step's diagnostics equal an independent run of that step
    for s in (0, 1, 22, 45):
        r = eng.synth_tiled(eng.alloc_tiled(n), seed=16, step=s)
        d = torch.zeros(8, dtype=torch.float64, device='cuda')
        eng.run_tiled(r, diag=d)
        assert torch.equal(d, diag[s]), s
    assert not torch.equal(diag[0], diag[1])


def _raw_fields(torch, eng, n, drv, seed=3):
    g = torch.Generator(device='cuda').manual_seed(seed)
    u = lambda lo, hi: torch.empty(n, dtype=eng.dtype, device='cuda').uniform_(lo, hi, generator=g)
    raw = list(drv[:9]) + [u(0.001, 0.02), u(0.001, 0.02), u(7e4, 1.0134e5), u(7e4, 1.0134e5), u(0, 3500)]
    fpar = torch.randint(0, 101, (n,), dtype=torch.uint8, device='cuda', generator=g)
    lai = torch.randint(0, 71, (n,), dtype=torch.uint8, device='cuda', generator=g)
    fpar[::97] = 255                                    # MODIS fill codes -> NaN
    lai[::89] = 250
    return raw, fpar, lai, u(8, 16)


@pytest.mark.parametrize('dtype,math', [('float64', 'fast'), ('float32', 'fast'), ('float32', 'mixed')])
@pytest.mark.parametrize('n', [8192 * 24 + 4 * 333, 4 * 700])
def test_every_form_on_a_tiled_raster_equals_the_plain_arrays(env, dtype, math, n):
    """mod16_et_form_tiled_*: potential ET, components and raw drivers on the tiled layout
    against mod16_et_pet_* / out_sep / mod16_et_raw_* on plain arrays -- same kernel, same
    arithmetic: identical bits, every output of every form."""
    torch, RasterEngine, table, _lib = env
    m = {'fast': _lib.MATH_FAST, 'mixed': _lib.MATH_MIXED}[math]
    eng = RasterEngine(table, dtype=dtype, math=m)
    cls, drv = eng.synth(n, seed=11, step=1)
    raw, fpar, lai, hours = _raw_fields(torch, eng, n, drv)

    def tiled(form, wide, rasters, day_hours=None):
        r = eng.alloc_tiled(n, form=form)
        r.slab.fill_(0xff)
        for dst, src in zip(r.wide, wide):
            r.put(dst, src)
        for dst, src in zip(r.bytes, rasters):
            r.put(dst, src)
        outs = eng.run_form_tiled(r, day_hours=day_hours)
        eng.check()
        return [r.flat(o) for o in outs]

    day, night = eng.run(cls, drv)
    got = tiled(_lib.FORM_TOTALS, drv, [cls])
    assert same(got[0], day) and same(got[1], night)

    want = eng.run_pet(cls, drv)
    got = tiled(_lib.FORM_PET, drv, [cls])
    for k in range(4):
        assert same(got[k], want[k]), ('pet', k)

    sep = eng.empty(n, 6)
    eng.run(cls, drv, out_sep=sep)
    got = tiled(_lib.FORM_COMPONENTS, drv, [cls])
    for k in range(6):
        assert same(got[k], sep[k]), ('components', k)
    d2, n2 = eng.run(cls, drv, *eng.empty(n, 2), out_sep=sep)
    got = tiled(_lib.FORM_TOTALS_COMPONENTS, drv, [cls])
    assert same(got[0], d2) and same(got[1], n2)
    for k in range(6):
        assert same(got[2 + k], sep[k]), ('totals + components', k)

    want = eng.run_raw(cls, raw, fpar, lai)
    got = tiled(_lib.FORM_RAW, raw, [cls, fpar, lai])
    assert same(got[0], want[0]) and same(got[1], want[1])
    # (one value for the hours of daylight: on plain device arrays the pipeline takes it
    # per pixel -- same arithmetic)
    want = eng.run_raw(cls, raw, fpar, lai, day_hours=torch.full_like(hours, 11.5))
    got = tiled(_lib.FORM_RAW_TOTAL8, raw, [cls, fpar, lai], day_hours=11.5)
    for k in range(3):
        assert same(got[k], want[k]), ('raw total8', k)
    want = eng.run_raw(cls, raw, fpar, lai, day_hours=hours)
    got = tiled(_lib.FORM_RAW_TOTAL8_HOURS, raw + [hours], [cls, fpar, lai])
    for k in range(3):
        assert same(got[k], want[k]), ('raw total8, hours per pixel', k)
    eng.check()


def test_form_tiled_argument_errors(env):
    torch, RasterEngine, table, _lib = env
    eng = RasterEngine(table)
    r = eng.alloc_tiled(8192, form=_lib.FORM_RAW_TOTAL8)
    with pytest.raises(ValueError):
        eng.run_form_tiled(r)                           # hours of daylight missing
    r = eng.alloc_tiled(8192, form=_lib.FORM_PET)
    r.form = 17
    with pytest.raises(Exception) as e:
        eng.run_form_tiled(r)
    assert 'unknown form' in str(e.value)
