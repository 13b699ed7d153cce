"""The domain guard on device-resident rasters (mod16_physics.hpp, "domain guard"): pixels
outside the domain of the production arithmetic -- infinities, fill values in a temperature or the
pressure, the pole of the Tetens formula -- sprinkled over a synthetic raster. The production
kernel must give what the reference-order kernel gives on them (NaN / zero / inf masks identical,
values to 1e-9; everywhere else it is the production arithmetic, bit for bit what it is without
such pixels) and its in-kernel diagnostics must be those of its outputs -- on both schedules
(small rasters revisit their flagged pieces inside the kernel, large ones in
et_stream_redo_kernel), both layouts, float64 and float32 (FAST, MIXED), and through the oracle on
windows."""
import numpy as np
import pytest

from oracle import mod16_oracle as oracle
from parity import assert_mixed_parity, assert_parity

pytestmark = pytest.mark.gpu

FILLS = [np.inf, -np.inf, -9999.0, 65535.0, 1e15, 3.4e38, -3.4e38, 35.85, 0.0, 1400.0]


@pytest.fixture(scope='module')
def env():
    import torch
    from mod16_amd import _lib
    from mod16_amd.raster import RasterEngine
    from mod16_amd.utils import restore_bplut, bplut_table
    from mod16_amd.models import COLLECTION61_BPLUT
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    return torch, RasterEngine, table, _lib


def sprinkle(torch, drv, n, every, seed):
    """One fill value in one driver of every `every`-th pixel (and a block of 300 consecutive
    pixels, so that whole waves are flagged too). Returns the number of touched pixels."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    idx = torch.arange(every // 2, n, every)
    idx = torch.cat([idx, torch.arange(n // 3, min(n, n // 3 + 300))]).unique()
    which = torch.randint(0, 14, (idx.numel(),), generator=g)
    val = torch.tensor(FILLS, dtype=torch.float64)[torch.randint(0, len(FILLS), (idx.numel(),), generator=g)]
    for k in range(14):
        sel = which == k
        drv[k][idx[sel].cuda()] = val[sel].to(drv[k].dtype).cuda()
    return idx


def host_diag(torch, day, night):
    d = day.double()
    g = night.double()
    return [float(torch.nansum(d)), float(torch.nansum(g)), float((~torch.isnan(d)).sum()),
            float((~torch.isnan(g)).sum()), float(torch.isnan(d).sum()), float(torch.isnan(g).sum()),
            float(torch.where(torch.isnan(d), -np.inf, d).max()), float(torch.where(torch.isnan(g), -np.inf, g).max())]


@pytest.mark.parametrize('n,every', [(1200 * 1200, 997), (1200 * 1200, 7), (40_000_000, 100_003), (40_000_000, 61),
                                      (8192 * 40 + 4 * 777, 1),
                                      (1 << 24, 4099)])      # float64: exactly 16384 runs, the first dynamic size
@pytest.mark.parametrize('dtype,math', [('float64', 'fast'), ('float32', 'fast'), ('float32', 'mixed')])
def test_flagged_pixels_on_device_rasters(env, n, every, dtype, math):
    torch, RasterEngine, table, _lib = env
    m = {'fast': _lib.MATH_FAST, 'mixed': _lib.MATH_MIXED}[math]
    eng = RasterEngine(table, dtype=dtype, math=m)
    ref = RasterEngine(table, dtype=dtype, math=_lib.MATH_EXACT if dtype == 'float64' else _lib.MATH_FAST)
    cls, drv = eng.synth(n, seed=9, step=1)
    clean_day, clean_night = eng.run(cls, drv)
    clean_day, clean_night = clean_day.clone(), clean_night.clone()
    idx = sprinkle(torch, drv, n, every, seed=n % 1000 + every).cuda()
    diag = torch.zeros(8, dtype=torch.float64, device='cuda')
    day, night = eng.run(cls, drv, diag=diag)
    eng.check()
    # untouched pixels: what they were without the flagged ones around them, bit for bit
    keep = torch.ones(n, dtype=torch.bool, device='cuda')
    keep[idx] = False
    for got, was in ((day, clean_day), (night, clean_night)):
        if math == 'mixed':
            # (the mixed form redoes the radiation balance of a whole WAVE in float64 when one
            # of its pixels sits next to a discontinuity -- mod16_mixed.hpp -- and a fill value
            # does: the neighbours' float32 results move within the form's tolerance)
            assert_mixed_parity(got[keep].cpu().numpy(), was[keep].cpu().numpy(), 'untouched pixels',
                                rtol=1e-4, atol_of_max=1e-6)
        else:
            assert torch.equal(torch.nan_to_num(got[keep], nan=-7.0), torch.nan_to_num(was[keep], nan=-7.0))
    # touched pixels: the reference-order arithmetic's results
    sub_cls = cls[idx].contiguous()
    sub_drv = [d[idx].contiguous() for d in drv]
    want = ref.run(sub_cls, sub_drv)
    ref.check()
    for got, w, what in ((day[idx], want[0], 'day'), (night[idx], want[1], 'night')):
        g_np, w_np = got.cpu().numpy(), w.cpu().numpy()
        if math == 'mixed':
            assert_mixed_parity(g_np, w_np, what)
        else:
            assert_parity(g_np, w_np, 1e-9 if dtype == 'float64' else 1e-6, what)
    # ... and, on a window, the oracle's
    if dtype == 'float64':
        bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
        with np.errstate(all='ignore'):
            o_day, o_night = oracle.evapotranspiration_raster(
                bplut, cls[idx[:20000]].cpu().numpy(), *[d[idx[:20000]].cpu().numpy() for d in drv])
        assert_parity(day[idx[:20000]].cpu().numpy(), o_day, 1e-8, 'day vs oracle')
        assert_parity(night[idx[:20000]].cpu().numpy(), o_night, 1e-8, 'night vs oracle')
    # the diagnostics are those of the outputs
    got = diag.cpu().numpy()
    want_d = host_diag(torch, day, night)
    assert np.array_equal(got[2:6], want_d[2:6]), (got, want_d)
    for k in (0, 1):
        assert np.isclose(got[k], want_d[k], rtol=1e-11 if np.isfinite(want_d[k]) else 0, equal_nan=True) or \
            (np.isinf(want_d[k]) and got[k] == want_d[k]), (k, got[k], want_d[k])
    assert got[6] == want_d[6] and got[7] == want_d[7], (got, want_d)
    # the tiled layout: same bits, outputs and diagnostics
    r = eng.to_tiled(cls, drv)
    d_tiled = torch.zeros(8, dtype=torch.float64, device='cuda')
    eng.run_tiled(r, diag=d_tiled)
    eng.check()
    assert torch.equal(torch.nan_to_num(r.flat(r.day), nan=-7.0), torch.nan_to_num(day, nan=-7.0))
    assert torch.equal(torch.nan_to_num(r.flat(r.night), nan=-7.0), torch.nan_to_num(night, nan=-7.0))
    assert np.array_equal(d_tiled.cpu().numpy()[2:], got[2:])
    assert np.allclose(d_tiled.cpu().numpy()[:2], got[:2], rtol=1e-12, equal_nan=True)
    # repeated launches: same bits
    d2 = torch.zeros(8, dtype=torch.float64, device='cuda')
    eng.run(cls, drv, out_day=day, out_night=night, diag=d2)
    assert torch.equal(torch.nan_to_num(d2, nan=-7.0), torch.nan_to_num(diag, nan=-7.0))


@pytest.mark.parametrize('dtype', ['float64', 'float32'])
def test_flagged_pixels_in_the_other_forms_at_size(env, dtype):
    """Potential ET, the six components and the raw-driver form on rasters large enough for the
    dynamic schedule (flagged pieces revisited by et_stream_redo_kernel<T, form>), fill values
    sprinkled in: every output equals the plain kernels' (MOD16_NO_DMA context: same pixel functions,
    flagged pixels redone inline) bit for bit; and a step captured into a HIP graph replays the
    revisit as well."""
    import os
    torch, RasterEngine, table, _lib = env
    n = 24_000_000 if dtype == 'float64' else 40_000_000
    eng = RasterEngine(table, dtype=dtype)
    os.environ['MOD16_NO_DMA'] = '1'
    try:
        plain = RasterEngine(table, dtype=dtype)
        plain.ctx = _lib.Context(0, experiments=True)
        plain.ctx.set_bplut(np.ascontiguousarray(table, np.float64))
    finally:
        del os.environ['MOD16_NO_DMA']
    cls, drv = eng.synth(n, seed=4, step=3)
    idx = sprinkle(torch, drv, n, 9973, seed=11).cuda()

    def same(a, b):
        return torch.equal(torch.nan_to_num(a, nan=-7.0, posinf=1e300, neginf=-1e300),
                           torch.nan_to_num(b, nan=-7.0, posinf=1e300, neginf=-1e300))

    got, want = eng.run_pet(cls, drv), plain.run_pet(cls, drv)
    for k, (g, w) in enumerate(zip(got, want)):
        assert same(g, w), ('potential ET', k)
    a, b = eng.empty(n, 6), plain.empty(n, 6)
    eng.run(cls, drv, None, None, out_sep=a)
    plain.run(cls, drv, None, None, out_sep=b)
    for k in range(6):
        assert same(a[k], b[k]), ('components', k)
    del a, b, got, want
    # raw drivers: the first nine fields pass through; humidity, surface pressure, elevation
    g = torch.Generator(device='cuda').manual_seed(2)
    u = lambda lo, hi: torch.empty(n, dtype=eng.dtype, device='cuda').uniform_(lo, hi, generator=g)
    raw = drv[:9] + [u(0.001, 0.02), u(0.001, 0.02), u(7e4, 1.0134e5), u(7e4, 1.0134e5), u(0, 3500)]
    raw[13][idx[::3]] = 5e4                     # above the guard's 40 km
    raw[9][idx[1::3]] = 1.5                     # a specific humidity above 1 kg/kg
    fpar = torch.randint(0, 101, (n,), dtype=torch.uint8, device='cuda', generator=g)
    lai = torch.randint(0, 71, (n,), dtype=torch.uint8, device='cuda', generator=g)
    hours = u(8, 16)
    got = eng.run_raw(cls, raw, fpar, lai, day_hours=hours)
    want = plain.run_raw(cls, raw, fpar, lai, day_hours=hours)
    eng.check()
    plain.check()
    for k, (g_, w_) in enumerate(zip(got, want)):
        assert same(g_, w_), ('raw drivers', k)
    # a captured step: the graph holds the revisit
    day, night = eng.empty(n, 2)
    diag = torch.zeros(8, dtype=torch.float64, device='cuda')
    step = eng.bind(cls, drv, day, night, diag, graph=True)
    step()
    torch.cuda.synchronize()
    d2 = torch.zeros(8, dtype=torch.float64, device='cuda')
    rd, rn = eng.run(cls, drv, diag=d2)
    assert same(day, rd) and same(night, rn) and same(diag, d2)
    pd, pn = plain.run(cls, drv)
    assert same(day, pd) and same(night, pn)


@pytest.mark.parametrize('switches', [{'MOD16_STATIC_BELOW': '0'}, {'MOD16_STATIC_BELOW': '0', 'MOD16_RUN_SHIFT': '6'},
                                      {'MOD16_RUN_SHIFT': '6'}, {'MOD16_STATIC_BELOW': '64'},
                                      {'MOD16_STREAM_BLOCKS': '1'}, {'MOD16_STREAM_BLOCKS': '1', 'MOD16_STATIC_BELOW': '64'}])
def test_flags_under_other_schedules(env, switches):
    """The flag record under the schedules the experiment switches select: a small raster on the
    DYNAMIC schedule (et_stream_redo_kernel on few runs), runs of 64 pieces (more pieces per run than
    the 52 flag bits: the last bit stands for the rest), the static schedule on a larger raster (more
    than 52 iterations per wave likewise), half the waves (one block per CU: twice the iterations per
    wave). Same outputs and diagnostics as the default context."""
    import os
    torch, RasterEngine, table, _lib = env
    n = 3_000_000 if switches.get('MOD16_STATIC_BELOW') != '64' else 20_000_000
    base = RasterEngine(table)
    for k, v in switches.items():
        os.environ[k] = v
    try:
        eng = RasterEngine(table)
        eng.ctx = _lib.Context(0, experiments=True)
        eng.ctx.set_bplut(np.ascontiguousarray(table, np.float64))
    finally:
        for k in switches:
            del os.environ[k]
    cls, drv = base.synth(n, seed=12)
    sprinkle(torch, drv, n, 501, seed=3)
    drv[5][n // 2: n // 2 + 40000] = 65535.0            # whole runs of flagged pixels
    d0 = torch.zeros(8, dtype=torch.float64, device='cuda')
    d1 = torch.zeros(8, dtype=torch.float64, device='cuda')
    want = base.run(cls, drv, diag=d0)
    got = eng.run(cls, drv, diag=d1)
    eng.check()
    for g, w in zip(got, want):
        assert torch.equal(torch.nan_to_num(g, nan=-7.0, posinf=1e300, neginf=-1e300),
                           torch.nan_to_num(w, nan=-7.0, posinf=1e300, neginf=-1e300))
    a, b = d1.cpu().numpy(), d0.cpu().numpy()
    assert np.array_equal(a[2:6], b[2:6]) and a[6] == b[6] and a[7] == b[7]
    assert np.allclose(a[:2], b[:2], rtol=1e-11, equal_nan=True) or (np.isinf(b[:2]).any() and np.array_equal(a[:2], b[:2]))


def test_flagged_pixels_through_the_numpy_path_over_many_tiles(env):
    """numpy in -> numpy out (HOST mode: 2 Mi-pixel tiles staged by eight threads through one
    context's workspace) with fill values sprinkled in, float64 and float32: what the
    reference-order arithmetic gives, masks identical."""
    import mod16_amd as m16
    from oracle import synth
    torch, RasterEngine, table, _lib = env
    n = 5 * (1 << 21) + 12345
    cls, drv = synth.drivers((n,), seed=8)
    rng = np.random.default_rng(1)
    idx = rng.choice(n, 60000, replace=False)
    which = rng.integers(0, 14, idx.size)
    val = np.array(FILLS)[rng.integers(0, len(FILLS), idx.size)]
    for k in range(14):
        drv[k] = drv[k].copy()
        drv[k][idx[which == k]] = val[which == k]
    got = m16.evapotranspiration_raster(table, cls, *drv)
    with np.errstate(all='ignore'):
        want = m16.evapotranspiration_raster(table, cls, *drv, math=_lib.MATH_EXACT)
    assert_parity(got[0], want[0], 1e-9, 'day')
    assert_parity(got[1], want[1], 1e-9, 'night')
    bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    sub = idx[:30000]
    with np.errstate(all='ignore'):
        o = oracle.evapotranspiration_raster(bplut, cls[sub], *[d[sub] for d in drv])
    assert_parity(got[0][sub], o[0], 1e-8, 'day vs oracle')
    assert_parity(got[1][sub], o[1], 1e-8, 'night vs oracle')
    d32 = [d.astype(np.float32) for d in drv]
    got32 = m16.evapotranspiration_raster(table, cls, *d32)
    with np.errstate(all='ignore'):
        o32 = oracle.evapotranspiration_raster(bplut, cls[sub], *[d[sub].astype(np.float64) for d in d32])
    with np.errstate(over='ignore'):          # a float64 result beyond float32's range rounds to inf, as the kernel's does
        w32 = [w.astype(np.float32) for w in o32]
    assert_parity(got32[0][sub], w32[0], 1e-6, 'day, float32')
    assert_parity(got32[1][sub], w32[1], 1e-6, 'night, float32')


@pytest.mark.parametrize('n', [1200 * 1200, 24_000_000])
def test_signalling_nans_do_not_hide_their_neighbours(env, n):
    """A SIGNALLING NaN bit pattern (0x7ff0000000000001: what uninitialised or bit-packed memory
    can hold; numpy's own NaN is quiet) in a driver that enters the guard's chain LATE -- pressure,
    VPD, the temperatures -- next to an infinity or a huge value in one that enters it EARLY
    (radiation, fPAR, LAI). Under MODE.IEEE = 1 v_max_f64 would return the quieted NaN and the
    next link drop the running maximum: the pixel would stay on the fast path. The kernels clear
    the bit (mod16_math.hpp: ignore_signalling_nans), so the pixel is redone in the reference's
    order: masks and values of the reference-order kernel and of the oracle. Both schedules."""
    torch, RasterEngine, table, _lib = env
    eng = RasterEngine(table)
    ref = RasterEngine(table, math=_lib.MATH_EXACT)
    cls, drv = eng.synth(n, seed=31)
    snan = torch.tensor([0x7ff0000000000001, -0x000ffffffffffff], dtype=torch.int64).view(torch.float64).cuda()
    assert bool(torch.isnan(snan).all())
    g = torch.Generator(device='cpu').manual_seed(5)
    idx = torch.arange(17, n, max(1, n // 40000))
    late = torch.tensor([11, 9, 10, 5, 6])[torch.randint(0, 5, (idx.numel(),), generator=g)]
    early = torch.tensor([0, 2, 4, 1, 3, 12, 13])[torch.randint(0, 7, (idx.numel(),), generator=g)]
    big = torch.tensor([np.inf, -np.inf, 1e300, -1e300])[torch.randint(0, 4, (idx.numel(),), generator=g)].double()
    which = torch.randint(0, 2, (idx.numel(),), generator=g)
    for k in range(14):
        sel = late == k
        if sel.any():
            drv[k][idx[sel].cuda()] = snan[which[sel].cuda()]
        sel = early == k
        if sel.any():
            drv[k][idx[sel].cuda()] = big[sel].cuda()
    idx = idx.cuda()
    diag = torch.zeros(8, dtype=torch.float64, device='cuda')
    day, night = eng.run(cls, drv, diag=diag)
    eng.check()
    want = ref.run(cls[idx].contiguous(), [d[idx].contiguous() for d in drv])
    ref.check()
    assert_parity(day[idx].cpu().numpy(), want[0].cpu().numpy(), 1e-9, 'day')
    assert_parity(night[idx].cpu().numpy(), want[1].cpu().numpy(), 1e-9, 'night')
    bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    sub = idx[:20000]
    with np.errstate(all='ignore'):
        o_day, o_night = oracle.evapotranspiration_raster(bplut, cls[sub].cpu().numpy(), *[d[sub].cpu().numpy() for d in drv])
    assert_parity(day[sub].cpu().numpy(), o_day, 1e-8, 'day vs oracle')
    assert_parity(night[sub].cpu().numpy(), o_night, 1e-8, 'night vs oracle')
    want_d = host_diag(torch, day, night)
    got = diag.cpu().numpy()
    assert np.array_equal(got[2:6], want_d[2:6]) and got[6] == want_d[6] and got[7] == want_d[7]
    # the other kernels that evaluate the guard: one pixel per thread / ragged, and float32 MIXED
    m = 4099
    d1, n1 = eng.run(cls[idx[:m]].contiguous(), [d[idx[:m]].contiguous() for d in drv])
    assert torch.equal(torch.nan_to_num(d1, nan=-7.0, posinf=9e300, neginf=-9e300),
                       torch.nan_to_num(day[idx[:m]], nan=-7.0, posinf=9e300, neginf=-9e300))
    assert torch.equal(torch.nan_to_num(n1, nan=-7.0, posinf=9e300, neginf=-9e300),
                       torch.nan_to_num(night[idx[:m]], nan=-7.0, posinf=9e300, neginf=-9e300))


@pytest.mark.parametrize('dtype,math', [('float64', 'fast'), ('float32', 'fast'), ('float32', 'mixed')])
def test_trusted_domain_is_the_same_arithmetic_without_the_test(env, dtype, math):
    """MOD16_DOMAIN_TRUSTED (RasterEngine(trusted=True)): on drivers inside the domain -- NaN fill
    included -- the instance without the domain test gives the guarded instance's bits: outputs and
    diagnostics, plain arrays (both schedules) and the tiled layout, direct launches and a captured
    step. (Outside the domain it returns whatever the rearranged arithmetic gives: the caller's
    word is what the flag is.)"""
    torch, RasterEngine, table, _lib = env
    m = {'fast': _lib.MATH_FAST, 'mixed': _lib.MATH_MIXED}[math]
    eng = RasterEngine(table, dtype=dtype, math=m)
    fast = RasterEngine(table, dtype=dtype, math=m, trusted=True)
    assert fast.math == (m | _lib.DOMAIN_TRUSTED)
    same = lambda a, b: torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))
    if math == 'mixed':
        # the guarded mixed form also computes the pixels of its cancellation class again in float64
        # (mod16_mixed.hpp, period_mixed: values orders of magnitude below the typical one), the
        # trusted one revisits nothing: everywhere else the same bits -- at most 1 value in 1000 differs,
        # none of them above 2 % of the largest, NaN masks and exact zeros alike
        def same(a, b):
            a, b = torch.nan_to_num(a.double(), nan=-7.0), torch.nan_to_num(b.double(), nan=-7.0)
            if a.numel() == 8:                  # diagnostics: counts and maxima alike, sums to 1e-9
                return bool(torch.equal(a[2:], b[2:])) and bool(torch.allclose(a[:2], b[:2], rtol=1e-9, atol=0))
            diff = a != b
            if not bool(diff.any()):
                return True
            return (float(diff.double().mean()) < 1e-3 and bool(torch.equal(a == 0, b == 0)) and bool(torch.equal(a == -7.0, b == -7.0))
                    and float(torch.maximum(a[diff].abs(), b[diff].abs()).max()) < 0.02 * float(b.abs().max()))
    for n in (1200 * 1200, 30_000_000):
        cls, drv = eng.synth(n, seed=41)
        d0 = torch.zeros(8, dtype=torch.float64, device='cuda')
        d1 = torch.zeros(8, dtype=torch.float64, device='cuda')
        want = eng.run(cls, drv, diag=d0)
        got = fast.run(cls, drv, diag=d1)
        fast.check()
        assert same(got[0], want[0]) and same(got[1], want[1]) and same(d0, d1), n
        r = fast.to_tiled(cls, drv)
        d2 = torch.zeros(8, dtype=torch.float64, device='cuda')
        step = fast.bind_tiled(r, d2)
        step()
        torch.cuda.synchronize()
        assert same(r.flat(r.day), want[0]) and same(r.flat(r.night), want[1])
        assert np.array_equal(d2.cpu().numpy()[2:], d0.cpu().numpy()[2:])
        assert np.allclose(d2.cpu().numpy()[:2], d0.cpu().numpy()[:2], rtol=1e-9 if math == 'mixed' else 1e-12)
        del r, step
