"""GPU parity of the production pipeline's other forms (et_stream_kernel,
mod16_amd/csrc/mod16_stream.hpp): potential ET, separate components and raw
drivers on dense device-resident rasters, against the oracle and against the
plain kernels of the same library (MOD16_NO_DMA=1 context).

float64 tolerance as in test_gpu_parity.py: worst pixel of the FAST
arithmetic within 1e-8 of the oracle, identical NaN / exact-zero masks; the
two kernel forms are instantiated from one pixel function whose contractions are
all written out (`#pragma clang fp contract(off)` + explicit fma), so they must
agree BIT FOR BIT -- asserted here for every form (tools/forms_agree.py prints the
count of differing values; round 2 allowed 1e-11 without having looked)."""
import os

import numpy as np
import pytest

from oracle import mod16_oracle as oracle
from parity import assert_parity, in_a_fresh_thread, same_bits

pytestmark = pytest.mark.gpu

RTOL = 1e-8
N = 64 * 16 * 2 * 37 + 64 * 5 + 3      # several runs per wave, a ragged piece, a scalar tail


@pytest.fixture(scope='module')
def env():
    import torch
    from mod16_amd.raster import RasterEngine
    from mod16_amd.utils import restore_bplut, bplut_table
    from mod16_amd.models import COLLECTION61_BPLUT
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    return torch, RasterEngine, table


def plain_engine(RasterEngine, table, dtype='float64'):
    """An engine whose context never takes the LDS-DMA pipelines."""
    os.environ['MOD16_NO_DMA'] = '1'
    try:
        from mod16_amd import _lib
        eng = RasterEngine(table, dtype=dtype)
        eng.ctx = _lib.Context(0, experiments=True)          # a context of its own (contexts are cached per device)
        eng.ctx.set_bplut(np.ascontiguousarray(table, np.float64))
        return eng
    finally:
        del os.environ['MOD16_NO_DMA']


def assert_same_bits(got, want, what):
    got, want = np.asarray(got), np.asarray(want)
    assert got.dtype == want.dtype and got.shape == want.shape, what
    assert np.array_equal(got, want, equal_nan=True), \
        '%s: %d values differ' % (what, int((~((got == want) | (np.isnan(got) & np.isnan(want)))).sum()))


def to_np(ts):
    return [t.cpu().numpy() for t in ts]


def bplut_of(table):
    return {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}


def test_potential_et_on_device(env):
    torch, RasterEngine, table = env
    eng = RasterEngine(table)
    cls, drv = eng.synth(N, seed=31)
    got = to_np(eng.run_pet(cls, drv))
    eng.check()
    h_cls, h_drv = cls.cpu().numpy(), to_np(drv)
    params = oracle.gather_params(bplut_of(table), h_cls)
    wd, wn = oracle.evapotranspiration(params, *h_drv)
    pd, pn = oracle.potential_et(params, *h_drv)
    for g, w, what in zip(got, (wd, wn, pd, pn), ('day', 'night', 'pet day', 'pet night')):
        assert_parity(g, w, RTOL, what)
    ref = to_np(plain_engine(RasterEngine, table).run_pet(cls, drv))
    for g, w in zip(got, ref):
        assert_same_bits(g, w, 'pipeline vs plain kernel')


@pytest.mark.parametrize('totals', [True, False])
def test_components_on_device(env, totals):
    torch, RasterEngine, table = env
    eng = RasterEngine(table)
    cls, drv = eng.synth(N, seed=32)
    sep = eng.empty(N, 6)
    day, night = eng.empty(N, 2) if totals else (None, None)
    eng.run(cls, drv, day, night, out_sep=sep)
    eng.check()
    h_cls, h_drv = cls.cpu().numpy(), to_np(drv)
    ws = oracle.evapotranspiration_raster(bplut_of(table), h_cls, *h_drv, separate=True)
    for got, want in zip(to_np(sep), list(ws[0]) + list(ws[1])):
        assert_parity(got, want, RTOL, 'component')
    if totals:
        wd, wn = oracle.evapotranspiration_raster(bplut_of(table), h_cls, *h_drv)
        assert_parity(day.cpu().numpy(), wd, RTOL, 'day')
        assert_parity(night.cpu().numpy(), wn, RTOL, 'night')
        # the totals of this form are the totals of the production kernel, bit for bit
        d2, n2 = eng.run(cls, drv)
        assert torch.equal(torch.nan_to_num(day), torch.nan_to_num(d2))
        assert torch.equal(torch.nan_to_num(night), torch.nan_to_num(n2))


def raw_inputs(n, seed, dtype):
    rng = np.random.default_rng(seed)
    t_d = rng.uniform(255, 305, n)
    t_n = t_d - rng.uniform(0, 12, n)
    raw = [rng.uniform(-100, 0, n), rng.uniform(-50, 0, n), rng.uniform(0, 360, n), np.zeros(n),
           rng.uniform(0.1, 0.22, n), t_d, t_n, rng.uniform(265, 300, n), t_n - rng.uniform(0, 3, n),
           rng.uniform(5e-4, 2e-2, n), rng.uniform(5e-4, 2e-2, n),
           rng.uniform(70000, 101340, n), rng.uniform(70000, 101340, n), rng.uniform(-50, 4500, n)]
    raw = [np.ascontiguousarray(a, dtype) for a in raw]
    fpar = rng.integers(0, 101, n).astype(np.uint8)
    lai = rng.integers(0, 70, n).astype(np.uint8)
    fill = rng.random(n) < 0.01
    fpar[fill] = rng.integers(249, 256, fill.sum()).astype(np.uint8)
    lai[fill] = 255
    fpar[rng.random(n) < 0.01] = 0
    lai[rng.random(n) < 0.01] = 0
    cls = rng.choice(np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12], np.uint8), n)
    hours = np.ascontiguousarray(rng.uniform(6, 18, n), dtype)
    return cls, raw, fpar, lai, hours


@pytest.mark.parametrize('hours_kind', ['none', 'array', 'scalar'])
def test_raw_drivers_on_device(env, hours_kind):
    torch, RasterEngine, table = env
    eng = RasterEngine(table)
    cls, raw, fpar, lai, hours = raw_inputs(N, 33, np.float64)
    dev = lambda a: torch.from_numpy(a).cuda()
    d_raw = [dev(a) for a in raw]
    h = {'none': None, 'array': dev(hours), 'scalar': 11.5}[hours_kind]
    got = eng.run_raw(dev(cls), d_raw, dev(fpar), dev(lai), day_hours=h)
    eng.check()
    want = oracle.evapotranspiration_raw(
        bplut_of(table), cls, raw, fpar, lai,
        day_hours={'none': None, 'array': hours, 'scalar': np.full(N, 11.5)}[hours_kind])
    assert len(got) == len(want) == (2 if hours_kind == 'none' else 3)
    for g, w, what in zip(to_np(got), want, ('day', 'night', 'total8')):
        assert_parity(g, w, RTOL, what)
    ref = plain_engine(RasterEngine, table).run_raw(dev(cls), d_raw, dev(fpar), dev(lai), day_hours=h)
    for g, w in zip(to_np(got), to_np(ref)):
        assert_same_bits(g, w, 'pipeline vs plain kernel')


def test_raw_drivers_host_path_and_float32(env):
    """numpy in / numpy out (staged tiles) takes the same pipeline; float32
    rasters are computed in float64 and rounded once."""
    torch, RasterEngine, table = env
    import mod16_amd
    cls, raw, fpar, lai, hours = raw_inputs(N, 34, np.float64)
    want = oracle.evapotranspiration_raw(bplut_of(table), cls, raw, fpar, lai, day_hours=hours)
    got = mod16_amd.evapotranspiration_raw(table, cls, *raw, fpar, lai, day_hours=hours)
    for g, w, what in zip(got, want, ('day', 'night', 'total8')):
        assert_parity(g, w, RTOL, what)
    got_s = mod16_amd.evapotranspiration_raw(table, cls, *raw, fpar, lai, day_hours=11.5)
    want_s = oracle.evapotranspiration_raw(bplut_of(table), cls, raw, fpar, lai, day_hours=np.full(N, 11.5))
    assert_parity(got_s[2], want_s[2], RTOL, 'total8, scalar hours')
    raw32 = [a.astype(np.float32) for a in raw]
    h32 = hours.astype(np.float32)
    got32 = mod16_amd.evapotranspiration_raw(table, cls, *raw32, fpar, lai, day_hours=h32)
    ref64 = mod16_amd.evapotranspiration_raw(
        table, cls, *[a.astype(np.float64) for a in raw32], fpar, lai, day_hours=h32.astype(np.float64))
    for a, b in zip(got32, ref64):
        assert a.dtype == np.float32
        assert np.array_equal(a, b.astype(np.float32), equal_nan=True)


def test_float32_components_and_pet_on_device(env):
    torch, RasterEngine, table = env
    e32, e64 = RasterEngine(table, dtype='float32'), RasterEngine(table)
    cls, drv32 = e32.synth(N, seed=35)
    drv64 = [d.double() for d in drv32]
    got = to_np(e32.run_pet(cls, drv32))
    ref = to_np(e64.run_pet(cls, drv64))
    for a, b in zip(got, ref):
        assert np.array_equal(a, b.astype(np.float32), equal_nan=True)
    sep32, sep64 = e32.empty(N, 6), e64.empty(N, 6)
    e32.run(cls, drv32, out_sep=sep32)
    e64.run(cls, drv64, out_sep=sep64)
    e32.check()
    e64.check()
    for a, b in zip(to_np(sep32), to_np(sep64)):
        assert np.array_equal(a, b.astype(np.float32), equal_nan=True)


def test_class_range_flag_from_the_pipeline(env):
    torch, RasterEngine, table = env
    eng = RasterEngine(table)
    cls, drv = eng.synth(4096, seed=1)
    cls[1234] = 200
    eng.run_pet(cls, drv)
    with pytest.raises(IndexError):
        eng.check()


@pytest.mark.parametrize('dtype', ['float64', 'float32'])
def test_pipeline_at_piece_run_and_chip_boundaries(env, dtype):
    """Sizes around one vector, one 64-vector piece, one run, one run per wave of
    the chip and one more: the pipeline's work distribution (run length, claimed
    runs, ragged last piece, scalar tail) against the plain kernels, all forms."""
    torch, RasterEngine, table = env
    eng = RasterEngine(table, dtype=dtype)
    plain = plain_engine(RasterEngine, table, dtype)
    v = 2 if dtype == 'float64' else 4
    piece = 64 * v
    chip = 2048 * 2 * piece                      # every wave of the chip gets a 2-piece run
    sizes = [v, piece - v, piece, piece + 1, 2 * piece, 8 * piece - 1, 8 * piece, 8 * piece + v + 1,
             17 * piece + 3, chip - piece, chip, chip + piece + 1, 3 * chip + 5 * piece + 2]
    big = max(sizes)
    cls, drv = eng.synth(big, seed=71)
    for n in sizes:
        c, d = cls[:n], [x[:n] for x in drv]
        diag = torch.zeros(8, dtype=torch.float64, device='cuda')
        day, night = eng.run(c, d, diag=diag)
        pd, pn = plain.run(c, d)
        assert torch.equal(torch.nan_to_num(day), torch.nan_to_num(pd)), n
        assert torch.equal(torch.nan_to_num(night), torch.nan_to_num(pn)), n
        want = eng.diagnostics(day, night)
        assert torch.equal(diag[2:], want[2:]), n
        assert torch.allclose(diag[:2], want[:2], rtol=1e-12, atol=0), n
        if n <= 17 * piece + 3 or n == chip + piece + 1:
            sep, psep = eng.empty(n, 6), plain.empty(n, 6)
            eng.run(c, d, out_sep=sep)
            plain.run(c, d, out_sep=psep)
            for a, b in zip(sep, psep):
                assert_same_bits(a.cpu().numpy(), b.cpu().numpy(), 'components, n = %d' % n)
            for a, b in zip(eng.run_pet(c, d), plain.run_pet(c, d)):
                assert_same_bits(a.cpu().numpy(), b.cpu().numpy(), 'potential ET, n = %d' % n)
    eng.check()
    plain.check()
    # the raw-driver forms at the same boundaries
    rcls, raw, fpar, lai, hours = raw_inputs(17 * piece + 3, 72, np.dtype(dtype).type)
    dev = lambda a: torch.from_numpy(a).cuda()
    rcls, raw, fpar, lai, hours = dev(rcls), [dev(a) for a in raw], dev(fpar), dev(lai), dev(hours)
    for n in [s for s in sizes if s <= 17 * piece + 3]:
        for h in (None, hours[:n], 11.5):
            got = eng.run_raw(rcls[:n], [a[:n] for a in raw], fpar[:n], lai[:n], day_hours=h)
            want = plain.run_raw(rcls[:n], [a[:n] for a in raw], fpar[:n], lai[:n], day_hours=h)
            for a, b in zip(got, want):
                assert_same_bits(a.cpu().numpy(), b.cpu().numpy(), 'raw drivers, n = %d' % n)


@pytest.mark.parametrize('n', [1, 7, 365, 4099, 65536])
def test_small_raw_calls_give_the_bits_of_the_staged_path(env, n):
    """evapotranspiration_raw on numpy arrays of up to 65536 pixels: the kernel reads and writes one
    page-locked buffer (the small path of mod16_et_raw_*, mod16_capi.hip) -- the staged path's bits
    (MOD16_SMALL_PIXELS=0), every hours-of-daylight form, float64 and float32 (FAST and MIXED);
    a class code numpy would refuse is an IndexError from both."""
    torch, RasterEngine, table = env
    import mod16_amd
    _lib = mod16_amd._lib

    def run(dtype, math):
        cls, raw, fpar, lai, hours = raw_inputs(n, 77 + n, dtype)
        scal = list(raw)
        scal[13] = float(raw[13][0])           # elevation as a broadcast scalar
        out = list(mod16_amd.evapotranspiration_raw(table, cls, *raw, fpar, lai, day_hours=hours, math=math))
        out += list(mod16_amd.evapotranspiration_raw(table, cls, *scal, fpar, lai, day_hours=11.5, math=math))
        out += list(mod16_amd.evapotranspiration_raw(table, cls, *raw, fpar, lai, math=math))
        return out

    for dtype, math in ((np.float64, _lib.MATH_FAST), (np.float64, _lib.MATH_EXACT),
                        (np.float32, _lib.MATH_FAST), (np.float32, _lib.MATH_MIXED)):
        small = run(dtype, math)
        staged = in_a_fresh_thread(lambda: run(dtype, math), {'MOD16_SMALL_PIXELS': '0'})
        assert len(small) == len(staged) == 8
        for i, (a, b) in enumerate(zip(small, staged)):
            assert a.dtype == dtype
            if math == _lib.MATH_MIXED:
                # (a ragged end is computed by the one-pixel kernel in float64 arithmetic on the staged
                # path and, padded to whole vectors, by the mixed-precision pipeline here: the mixed
                # form's own tolerance between them, not bits)
                assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a == 0, b == 0), (n, i)
                assert np.allclose(a, b, rtol=2e-3, atol=1e-6 * np.nanmax(np.abs(b), initial=0.0), equal_nan=True), (n, i)
            else:
                assert same_bits(a, b), (n, dtype, math, i)
    cls, raw, fpar, lai, hours = raw_inputs(n, 77 + n, np.float64)
    cls[n - 1] = 13
    for envv in ({}, {'MOD16_SMALL_PIXELS': '0'}):
        with pytest.raises(IndexError):
            in_a_fresh_thread(lambda: mod16_amd.evapotranspiration_raw(table, cls, *raw, fpar, lai), envv)
