"""GPU tests of the device-resident path (RasterEngine over the C ABI in
DEVICE mode): on-device generator, zero-copy forward run, diagnostics."""
import numpy as np
import pytest

from oracle import mod16_oracle as oracle
from parity import assert_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import torch
    from mod16_amd.raster import RasterEngine
    from mod16_amd.utils import restore_bplut, bplut_table
    from mod16_amd.models import COLLECTION61_BPLUT
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    return torch, RasterEngine, table


def to_np(ts):
    return [t.cpu().numpy() for t in ts]


def test_generator_is_tiling_invariant(env):
    """Any split of the raster over ranks sees the same field."""
    torch, RasterEngine, table = env
    eng = RasterEngine(table)
    n = 300000
    cls, drv = eng.synth(n, seed=16)
    parts = [eng.synth(m, seed=16, pixel_offset=off) for off, m in ((0, 100001), (100001, 199999))]
    assert torch.equal(cls, torch.cat([p[0] for p in parts]))
    for k in range(14):
        a = drv[k]
        b = torch.cat([p[1][k] for p in parts])
        assert torch.equal(torch.nan_to_num(a, nan=-1.0), torch.nan_to_num(b, nan=-1.0)), k
    # the field is what SURVEY.md section 8d prescribes
    h = to_np(drv)
    c = cls.cpu().numpy()
    assert set(np.unique(c)) <= set(range(13)) and (c == 0).any() and (c == 11).any()
    assert (h[3] == 0).all()                                 # sw_rad_night
    assert 254 < np.nanmin(h[5]) and np.nanmax(h[5]) < 306   # temp_day
    assert (h[5] - h[6] >= 0).all() and (h[6] - h[8] >= 0).all()
    assert np.isnan(h[12]).mean() > 0.002 and (h[12] == 0).any() and (h[12] == 1).any()
    assert (h[13] == 0).any() and np.isnan(h[13]).any()
    # different seed / step -> different field; land cover stays
    cls2, drv2 = eng.synth(n, seed=16, step=1)
    assert torch.equal(cls, cls2) and not torch.equal(drv[5], drv2[5])


@pytest.mark.parametrize('dtype,rtol', [('float64', 1e-8)])
def test_device_run_matches_oracle_and_host_path(env, dtype, rtol):
    torch, RasterEngine, table = env
    import mod16_amd
    eng = RasterEngine(table, dtype=dtype)
    n = 1200 * 1200
    cls, drv = eng.synth(n, seed=7)
    day, night = eng.run(cls, drv)
    eng.check()
    h_cls, h_drv = cls.cpu().numpy(), to_np(drv)
    bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    wd, wn = oracle.evapotranspiration_raster(bplut, h_cls, *h_drv)
    assert_parity(day.cpu().numpy(), wd, rtol, 'day')
    assert_parity(night.cpu().numpy(), wn, rtol, 'night')
    # numpy-in/numpy-out path gives bit-identical results (same kernel)
    hd, hn = mod16_amd.evapotranspiration_raster(table, h_cls, *h_drv)
    assert np.array_equal(hd, day.cpu().numpy(), equal_nan=True)
    assert np.array_equal(hn, night.cpu().numpy(), equal_nan=True)
    # components on request
    sep = eng.empty(n, 6)
    eng.run(cls, drv, out_sep=sep)
    ws = oracle.evapotranspiration_raster(bplut, h_cls, *h_drv, separate=True)
    for got, want in zip(to_np(sep), list(ws[0]) + list(ws[1])):
        assert_parity(got, want, rtol, 'component')


def test_exact_and_fast_kernels_agree_on_device(env):
    """Size-independent cross-check used at full raster sizes: the production
    kernel against the reference-order kernel, compared on the GPU."""
    torch, RasterEngine, table = env
    from mod16_amd import _lib
    n = 4 * 1000 * 1000
    fast = RasterEngine(table)
    exact = RasterEngine(table, math=_lib.MATH_EXACT)
    cls, drv = fast.synth(n, seed=3)
    d1, n1 = fast.run(cls, drv)
    d2, n2 = exact.run(cls, drv)
    fast.check()
    for a, b in ((d1, d2), (n1, n2)):
        assert torch.equal(torch.isnan(a), torch.isnan(b))
        assert torch.equal(a == 0, b == 0)
        ok = torch.isfinite(b) & (b != 0)
        rel = ((a[ok] - b[ok]).abs() / b[ok].abs())
        assert float(rel.max()) < 1e-7 and float(rel.median()) < 1e-13


def test_diagnostics_fused_standalone_numpy(env):
    torch, RasterEngine, table = env
    eng = RasterEngine(table)
    for n in (2 * 123457, 1200 * 1200 + 1):    # even: in-kernel reduction; odd: fallback
        cls, drv = eng.synth(n, seed=5)
        day, night = eng.empty(n, 2)
        fused = torch.zeros(8, dtype=torch.float64, device='cuda')
        eng.run(cls, drv, day, night, diag=fused)
        alone = eng.diagnostics(day, night)
        eng.check()
        d, g = day.cpu().numpy(), night.cpu().numpy()
        want = np.array([np.nansum(d), np.nansum(g), (~np.isnan(d)).sum(), (~np.isnan(g)).sum(),
                         np.isnan(d).sum(), np.isnan(g).sum(), np.nanmax(d), np.nanmax(g)])
        for got in (fused.cpu().numpy(), alone.cpu().numpy()):
            np.testing.assert_allclose(got[:2], want[:2], rtol=1e-12)
            assert np.array_equal(got[2:], want[2:]), (got, want)
        # deterministic: a second launch gives the same bits
        again = torch.zeros(8, dtype=torch.float64, device='cuda')
        eng.run(cls, drv, day, night, diag=again)
        assert torch.equal(fused, again)


@pytest.mark.parametrize('dtype', ['float64', 'float32'])
def test_small_rasters_finish_their_diagnostics_in_the_kernel(env, dtype):
    """A small raster's partials are added up by the pipeline kernel's last block (one
    partial per wave under the static schedule, per run otherwise, up to 16384 of them).
    Sizes: one wave, a few blocks, a 1200 x 1200 tile, the largest statically scheduled
    rasters and the first dynamically scheduled ones, plain arrays and tiled, launches back to
    back on one stream and alternating between two: always the stand-alone reduction's sums
    and counts, and the same bits every time (whichever block happens to finish last)."""
    torch, RasterEngine, table = env
    eng = RasterEngine(table, dtype=dtype)
    vec = 16 // eng.np_dtype.itemsize
    side = torch.cuda.Stream()
    for n in (vec * 64, vec * 64 * 9, 1200 * 1200, vec * 64 * 2 * 16000, vec * 64 * 2 * 16500,
              vec * 64 * 8 * 16300):
        for tiled in (False, True):
            if tiled:
                r = eng.synth_tiled(eng.alloc_tiled(n), seed=3)
                launch = lambda d: eng.run_tiled(r, diag=d)
                flat = lambda: (r.flat(r.day), r.flat(r.night))
            else:
                cls, drv, day, night = eng.alloc_raster(n)
                eng.synth(n, seed=3, out=(cls, drv))
                launch = lambda d: eng.run(cls, drv, day, night, diag=d)
                flat = lambda: (day, night)
            vecs = [torch.zeros(8, dtype=torch.float64, device='cuda') for _ in range(24)]
            for d in vecs[:16]:
                launch(d)
            torch.cuda.synchronize()
            for i, d in enumerate(vecs[16:]):           # two streams take turns
                if i % 2:
                    with torch.cuda.stream(side):
                        launch(d)
                else:
                    launch(d)
            torch.cuda.synchronize()
            alone = eng.diagnostics(*flat())
            eng.check()
            for d in vecs[1:]:
                assert torch.equal(vecs[0], d), (n, tiled)
            got, ref = vecs[0].cpu().numpy(), alone.cpu().numpy()
            np.testing.assert_allclose(got[:2], ref[:2], rtol=1e-12, err_msg=str((n, tiled)))
            assert np.array_equal(got[2:], ref[2:]), (n, tiled, got, ref)


def test_diagnostics_are_schedule_independent_at_size(env):
    """The production kernel hands runs of pixels to whichever wave is ready;
    its diagnostics are accumulated per run and summed in fixed order, so
    repeated launches must agree bit for bit even with many more runs than
    waves (here ~36000 runs for 2048 waves)."""
    torch, RasterEngine, table = env
    eng = RasterEngine(table)
    n = 1700 * 43200
    cls, drv, day, night = eng.alloc_raster(n)
    eng.synth(n, seed=11, out=(cls, drv))
    vecs = []
    for _ in range(4):
        d = torch.zeros(8, dtype=torch.float64, device='cuda')
        eng.run(cls, drv, day, night, diag=d)
        vecs.append(d)
    alone = eng.diagnostics(day, night)
    eng.check()
    for d in vecs[1:]:
        assert torch.equal(vecs[0], d)
    got, ref = vecs[0].cpu().numpy(), alone.cpu().numpy()
    np.testing.assert_allclose(got[:2], ref[:2], rtol=1e-12)
    assert np.array_equal(got[2:], ref[2:])


def test_deferred_class_range_error(env):
    torch, RasterEngine, table = env
    eng = RasterEngine(table)
    cls, drv = eng.synth(4096, seed=1)
    cls[77] = 200
    day, night = eng.run(cls, drv)
    with pytest.raises(IndexError):
        eng.check()
    assert bool(torch.isnan(day[77])) and bool(torch.isnan(night[77]))
    eng.check()      # flag cleared


def test_scalar_driver_on_device(env):
    """sw_rad_night = 0 as a broadcast scalar instead of a dense array."""
    torch, RasterEngine, table = env
    eng = RasterEngine(table)
    cls, drv = eng.synth(50000, seed=2)
    d1, n1 = eng.run(cls, drv)
    d2, n2 = eng.run(cls, drv[:3] + [0.0] + drv[4:])
    eng.check()
    assert torch.equal(torch.nan_to_num(d1), torch.nan_to_num(d2))
    assert torch.equal(torch.nan_to_num(n1), torch.nan_to_num(n2))


@pytest.mark.parametrize('world', [2, 8])
def test_band_sharding_composes_to_the_global_run(env, world):
    """BASELINE.json configs[2] on one device: the raster cut into `world` row
    bands (what N ranks do) gives bit-identical outputs to the single run, and
    the per-band diagnostics add up to the global ones."""
    torch, RasterEngine, table = env
    from mod16_amd import dist as tiles
    eng = RasterEngine(table)
    rows, cols = 2400, 1440      # small stand-in for 21600 x 43200 (divisible by 2 and 8 bands)
    n = rows * cols
    cls, drv = eng.synth(n, seed=16)
    gdiag = torch.zeros(8, dtype=torch.float64, device='cuda')
    gday, gnight = eng.run(cls, drv, diag=gdiag)
    day_parts, night_parts, diags = [], [], []
    for r in range(world):
        off, m = tiles.pixel_range(rows, cols, r, world)
        bcls, bdrv = eng.synth(m, seed=16, pixel_offset=off)      # each rank makes its own band
        assert torch.equal(bcls, cls[off:off + m])
        d = torch.zeros(8, dtype=torch.float64, device='cuda')
        bd, bn = eng.run(bcls, bdrv, diag=d)
        day_parts.append(bd)
        night_parts.append(bn)
        diags.append(d.clone())
    eng.check()
    assert torch.equal(torch.nan_to_num(torch.cat(day_parts), nan=-1.0),
                       torch.nan_to_num(gday, nan=-1.0))
    assert torch.equal(torch.nan_to_num(torch.cat(night_parts), nan=-1.0),
                       torch.nan_to_num(gnight, nan=-1.0))
    tot = torch.stack(diags)
    g = gdiag.cpu().numpy()
    np.testing.assert_allclose(tot[:, :2].sum(0).cpu().numpy(), g[:2], rtol=1e-12)
    assert np.array_equal(tot[:, 2:6].sum(0).cpu().numpy(), g[2:6])
    assert np.array_equal(tot[:, 6:].max(0).values.cpu().numpy(), g[6:])


def test_time_series_streaming_ring(env):
    """BASELINE.json configs[3] at test size: 46 steps through the two-slot
    ring give, step by step, exactly what independent runs give."""
    torch, RasterEngine, table = env
    eng = RasterEngine(table)
    n, steps = 64 * 1440, 46
    acc = torch.zeros(n, dtype=torch.float64, device='cuda')

    def accumulate(s, day, night):
        acc.add_(torch.nan_to_num(day) + torch.nan_to_num(night))

    diag, day, night = eng.run_series(n, steps, seed=16, on_step=accumulate)
    eng.check()
    assert diag.shape == (steps, 8)
    want_acc = torch.zeros_like(acc)
    for s in (0, 1, 2, 17, 45):
        cls, drv = eng.synth(n, seed=16, step=s)
        d = torch.zeros(8, dtype=torch.float64, device='cuda')
        wd, wn = eng.run(cls, drv, diag=d)
        assert torch.equal(d, diag[s]), s
        if s == steps - 1:
            assert torch.equal(torch.nan_to_num(wd), torch.nan_to_num(day))
            assert torch.equal(torch.nan_to_num(wn), torch.nan_to_num(night))
    for s in range(steps):
        cls, drv = eng.synth(n, seed=16, step=s)
        wd, wn = eng.run(cls, drv)
        want_acc.add_(torch.nan_to_num(wd) + torch.nan_to_num(wn))
    assert torch.equal(acc, want_acc)
    # steps differ from one another (the generator is keyed on the step)
    assert not torch.equal(diag[0], diag[1])


def test_time_series_steps_match_the_oracle(env):
    """configs[3] against the ORACLE, not against another HIP launch: windows of
    >= 300 k pixels of inputs and outputs of the first, a middle and the last
    step of a 46-step series are copied back while the ring streams on, and the
    numpy oracle runs on exactly those input bits (float64, 1e-8; NaN and
    exact-zero masks identical); the step's diagnostics are checked against
    numpy over the whole raster of that step."""
    torch, RasterEngine, table = env
    eng = RasterEngine(table)
    rows, cols, steps = 600, 1440, 46
    n = rows * cols
    w0, w1 = 123 * cols + 64, 123 * cols + 64 + 320000         # one >= 300 k-pixel window
    bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    bufs = eng.alloc_series(n)
    grabbed = {}

    def grab(s, day, night):
        if s in (0, 23, steps - 1):
            # enqueued on the compute stream right behind step s: the copies see the
            # slot of step s before the ingest stream may refill it
            grabbed[s] = ([d[w0:w1].clone() for d in bufs['ring'][s % 2]],
                          bufs['cls'][w0:w1].clone(), day[w0:w1].clone(), night[w0:w1].clone(),
                          day.clone(), night.clone())

    diag, _, _ = eng.run_series(n, steps, seed=16, on_step=grab, buffers=bufs)
    eng.check()
    assert sorted(grabbed) == [0, 23, steps - 1]
    for s, (drv, cls, day, night, fday, fnight) in grabbed.items():
        want_day, want_night = oracle.evapotranspiration_raster(bplut, cls.cpu().numpy(), *to_np(drv))
        assert_parity(day.cpu().numpy(), want_day, 1e-8, 'step %d day' % s)
        assert_parity(night.cpu().numpy(), want_night, 1e-8, 'step %d night' % s)
        hd, hn = fday.cpu().numpy(), fnight.cpu().numpy()
        g = diag[s].cpu().numpy()
        np.testing.assert_allclose(g[:2], [np.nansum(hd), np.nansum(hn)], rtol=1e-12)
        assert g[4] == np.isnan(hd).sum() and g[5] == np.isnan(hn).sum()
        assert g[2] == n - g[4] and g[3] == n - g[5]
        assert g[6] == np.nanmax(hd) and g[7] == np.nanmax(hn)
    # the windows differ from step to step (the ring really advanced)
    assert not torch.equal(torch.nan_to_num(grabbed[0][2]), torch.nan_to_num(grabbed[23][2]))


@pytest.mark.parametrize('graph', [True, False])
def test_bound_launch_equals_run(env, graph):
    torch, RasterEngine, table = env
    eng = RasterEngine(table)
    n = 200000
    cls, drv = eng.synth(n, seed=9)
    d1 = torch.zeros(8, dtype=torch.float64, device='cuda')
    a, b = eng.run(cls, drv, diag=d1)
    day, night = eng.empty(n, 2)
    d2 = torch.zeros(8, dtype=torch.float64, device='cuda')
    launch = eng.bind(cls, drv, day, night, d2, graph=graph)
    for _ in range(3):
        launch()
    eng.check()
    assert torch.equal(d1, d2)
    assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(day))
    assert torch.equal(torch.nan_to_num(b), torch.nan_to_num(night))
    # a bound launch follows the raster's contents (the graph holds pointers, not values)
    eng.synth(n, seed=10, out=(cls, drv))
    launch()
    a2, b2 = eng.run(cls, drv, diag=d1)
    eng.check()
    assert torch.equal(d1, d2) and not torch.equal(torch.nan_to_num(a), torch.nan_to_num(a2))
    assert torch.equal(torch.nan_to_num(a2), torch.nan_to_num(day))
    assert torch.equal(torch.nan_to_num(b2), torch.nan_to_num(night))


def test_bind_right_behind_the_generator(env):
    """`bind` validates its arguments with one launch on the library's own stream: it
    must wait for whatever the caller's stream is still writing into the raster (here the
    generator over memory full of 0xff bytes -- class codes >= 13 if read too early)."""
    torch, RasterEngine, table = env
    eng = RasterEngine(table)
    n = 40 * 1200 * 1200
    cls, drv, day, night = eng.alloc_raster(n)
    for t in drv + [cls.view(torch.uint8)]:
        t.view(torch.uint8).fill_(0xff)
    torch.cuda.synchronize()
    d = torch.zeros(8, dtype=torch.float64, device='cuda')
    eng.synth(n, seed=12, out=(cls, drv))
    step = eng.bind(cls, drv, day, night, d)          # no synchronisation in between
    step()
    eng.check()                                       # no IndexError: nothing was read early
    r = eng.synth_tiled(eng.alloc_tiled(n), seed=12)
    d2 = torch.zeros(8, dtype=torch.float64, device='cuda')
    r.slab.fill_(0xff)
    eng.synth_tiled(r, seed=12)
    step2 = eng.bind_tiled(r, d2)
    step2()
    eng.check()
    # same pixels; the sums run over partials of different run lengths on the two layouts
    assert torch.allclose(d, d2, rtol=1e-12, atol=0) and torch.equal(d[2:6], d2[2:6])


def test_bound_graph_survives_workspace_growth(env):
    """A captured graph owns its diagnostics workspace: launches of LARGER
    rasters on the same engine (DEVICE and HOST mode) make the context's own
    workspace grow -- it is freed and re-allocated -- and the graph bound before
    that must still replay correctly afterwards."""
    torch, RasterEngine, table = env
    import mod16_amd
    eng = RasterEngine(table)
    n = 1200 * 1200
    cls, drv = eng.synth(n, seed=21)
    day, night = eng.empty(n, 2)
    d_graph = torch.zeros(8, dtype=torch.float64, device='cuda')
    launch = eng.bind(cls, drv, day, night, d_graph, graph=True)
    launch()
    eng.check()
    first = d_graph.clone()
    # a raster 40 x larger through the same context: DEVICE mode with diagnostics ...
    big = 40 * n
    bcls, bdrv = eng.synth(big, seed=22)
    d_big = torch.zeros(8, dtype=torch.float64, device='cuda')
    eng.run(bcls, bdrv, diag=d_big)
    eng.check()
    # ... and HOST mode (stages 2 Mi-pixel tiles, reserves its own workspace size)
    hcls = bcls[:5 * n].cpu().numpy()
    hdrv = [d[:5 * n].cpu().numpy() for d in bdrv]
    ctx = eng.ctx
    out = [np.empty(5 * n), np.empty(5 * n)]
    ctx.et(np.float64, hcls.ctypes.data, [d.ctypes.data for d in hdrv], [1] * 14, None, None,
           5 * n, out[0].ctypes.data, out[1].ctypes.data, None)
    del bcls, bdrv
    torch.cuda.empty_cache()
    scratch = torch.full((64 << 20,), 7.0, dtype=torch.float64, device='cuda')   # reuse freed memory
    d_graph.zero_()
    for _ in range(3):
        launch()
    eng.check()
    assert torch.equal(d_graph, first)
    assert bool((scratch == 7.0).all())
    d_run = torch.zeros(8, dtype=torch.float64, device='cuda')
    a, b = eng.run(cls, drv, diag=d_run)
    eng.check()
    assert torch.equal(d_run, d_graph)
    assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(day))
    assert torch.equal(torch.nan_to_num(b), torch.nan_to_num(night))


def test_engines_with_different_tables_do_not_disturb_each_other(env):
    """Each RasterEngine has its own context and BPLUT copy: interleaved launches
    (and bound graphs) of two engines with different tables, and a numpy-path
    call with a third table in between, each compute with their own table."""
    torch, RasterEngine, table = env
    import mod16_amd
    table2 = table.copy()
    table2[:, 7] *= 0.5          # csl
    table2[:, 10] = 400.0        # beta
    e1, e2 = RasterEngine(table), RasterEngine(table2)
    n = 300000
    cls, drv = e1.synth(n, seed=31)
    want1 = [t.clone() for t in RasterEngine(table).run(cls, drv)]
    want2 = [t.clone() for t in RasterEngine(table2).run(cls, drv)]
    torch.cuda.synchronize()
    assert not torch.equal(torch.nan_to_num(want1[0]), torch.nan_to_num(want2[0]))
    d1, d2 = (torch.zeros(8, dtype=torch.float64, device='cuda') for _ in range(2))
    o1, o2 = e1.empty(n, 2), e2.empty(n, 2)
    g1 = e1.bind(cls, drv, o1[0], o1[1], d1)
    g2 = e2.bind(cls, drv, o2[0], o2[1], d2)
    table3 = table.copy()
    table3[:, 4] *= 2.0
    h = [d[:5000].cpu().numpy() for d in drv]
    for _ in range(2):
        g1()
        a2 = e2.run(cls, drv)
        mod16_amd.evapotranspiration_raster(table3, cls[:5000].cpu().numpy(), *h)
        g2()
        a1 = e1.run(cls, drv)
    torch.cuda.synchronize()
    for got, want in ((o1, want1), (a1, want1), (o2, want2), (a2, want2)):
        for g, w in zip(got, want):
            assert torch.equal(torch.nan_to_num(g), torch.nan_to_num(w))


def test_numpy_path_from_two_threads(env):
    """The reference's functions are pure and may be called from several threads
    (SURVEY.md section 8b). ctypes drops the GIL: two threads run
    MOD16.evapotranspiration / evapotranspiration_raster concurrently on different
    inputs and tables; both must equal the serial results."""
    import threading
    torch, RasterEngine, table = env
    import mod16_amd
    from oracle import synth
    table2 = table.copy()
    table2[:, 10] = 500.0
    jobs = []
    for k, tab in enumerate((table, table2)):
        cls, drv = synth.drivers((700, 3000), seed=40 + k)      # 2.1 M pixels: more than one staged tile
        jobs.append((tab, cls, drv))
    serial = [mod16_amd.evapotranspiration_raster(tab, cls, *drv) for tab, cls, drv in jobs]
    params = dict(zip(mod16_amd.MOD16.required_parameters, table[7]))
    serial_m = mod16_amd.MOD16(params).evapotranspiration(*jobs[0][2])
    results, errors = {}, []

    def work(i):
        try:
            tab, cls, drv = jobs[i % 2]
            for rep in range(3):
                if i == 2:
                    results[(i, rep)] = mod16_amd.MOD16(params).evapotranspiration(*jobs[0][2])
                else:
                    results[(i, rep)] = mod16_amd.evapotranspiration_raster(tab, cls, *drv)
        except Exception as exc:       # surfaced below
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for (i, rep), got in results.items():
        want = serial_m if i == 2 else serial[i]
        for g, w in zip(got, want):
            assert np.array_equal(g, w, equal_nan=True), (i, rep)


def test_result_arrays_come_from_a_pinned_pool(env):
    """Result arrays of the numpy entry points are page-locked blocks that return
    to a pool when the array dies and are handed out again: ordinary arrays to
    the caller (writeable, contiguous, correct), no fresh page faults per call."""
    import gc
    torch, RasterEngine, table = env
    import mod16_amd
    from mod16_amd import _lib
    from oracle import synth
    cls, drv = synth.drivers((900, 1000), seed=50)
    day, night = mod16_amd.evapotranspiration_raster(table, cls, *drv)
    assert day.flags.writeable and day.flags.c_contiguous and day.base is not None
    addr = day.ctypes.data
    keep = day.copy()
    del day, night
    gc.collect()
    assert _lib.pinned.cached >= 2 * 900 * 1000 * 8
    day2, night2 = mod16_amd.evapotranspiration_raster(table, cls, *drv)
    assert addr in (day2.ctypes.data, night2.ctypes.data)          # the block came back
    assert np.array_equal(day2, keep, equal_nan=True)
    day2 += 1.0                                                    # an ordinary array
    # a caller that keeps results keeps its memory: nothing is recycled under it
    a = mod16_amd.evapotranspiration_raster(table, cls, *drv)
    b = mod16_amd.evapotranspiration_raster(table, cls, *drv)
    assert len({a[0].ctypes.data, a[1].ctypes.data, b[0].ctypes.data, b[1].ctypes.data}) == 4
    assert np.array_equal(a[0], b[0], equal_nan=True)


def test_unaligned_device_pointers_take_the_scalar_path(env):
    """Tensors that start 8 bytes into an allocation are not 16-byte aligned:
    the library must fall back to the scalar-access kernel, same results."""
    torch, RasterEngine, table = env
    eng = RasterEngine(table)
    n = 100001
    cls, drv = eng.synth(n + 1, seed=4)
    want_d, want_n = eng.run(cls[1:].clone(), [d[1:].clone() for d in drv])
    day = torch.empty(n + 1, dtype=torch.float64, device='cuda')
    night = torch.empty(n + 1, dtype=torch.float64, device='cuda')
    got_d, got_n = eng.run(cls[1:], [d[1:] for d in drv], day[1:], night[1:])
    eng.check()
    assert drv[0][1:].data_ptr() % 16 == 8
    assert torch.equal(torch.nan_to_num(got_d), torch.nan_to_num(want_d))
    assert torch.equal(torch.nan_to_num(got_n), torch.nan_to_num(want_n))


@pytest.mark.parametrize('dtype', ['float64', 'float32'])
def test_slab_and_scattered_drivers_agree(env, dtype):
    """Equally spaced driver arrays (one slab) take the scalar base + pitch
    addressing of the production kernel, separately allocated ones the
    14-pointer form: same pixels, same bits, same diagnostics."""
    torch, RasterEngine, table = env
    eng = RasterEngine(table, dtype=dtype)
    n = 300000 + 36          # ragged last piece
    cls, drv, day1, night1 = eng.alloc_raster(n)
    eng.synth(n, seed=21, out=(cls, drv))
    gaps = [drv[k + 1].data_ptr() - drv[k].data_ptr() for k in range(13)]
    assert len(set(gaps)) == 1, 'alloc_raster lays the drivers out equally spaced'
    d1 = torch.zeros(8, dtype=torch.float64, device='cuda')
    eng.run(cls, drv, day1, night1, diag=d1)
    # copies whose spacing is irregular: 16-byte-aligned views at growing offsets
    scattered = []
    for k, d in enumerate(drv):
        buf = torch.empty(n + 4 * (k * k + 1), dtype=d.dtype, device='cuda')
        view = buf[4 * k * k:4 * k * k + n]
        view.copy_(d)
        scattered.append(view)
    gaps = [scattered[k + 1].data_ptr() - scattered[k].data_ptr() for k in range(13)]
    assert len(set(gaps)) > 1
    d2 = torch.zeros(8, dtype=torch.float64, device='cuda')
    day2, night2 = eng.run(cls, scattered, diag=d2)
    eng.check()
    assert torch.equal(torch.nan_to_num(day1), torch.nan_to_num(day2))
    assert torch.equal(torch.nan_to_num(night1), torch.nan_to_num(night2))
    assert torch.equal(d1, d2)


def test_tuned_allocation_returns_a_usable_slab(env):
    torch, RasterEngine, table = env
    eng = RasterEngine(table)
    n = 400000
    extras = (0, 1 << 20, 3 << 20)
    (cls, drv, day, night), report = eng.alloc_raster_tuned(n, extras=extras)
    assert sorted(report['extra_bytes_ms']) == sorted(str(e) for e in extras)
    assert report['chosen_extra_bytes'] in extras
    gaps = {drv[k + 1].data_ptr() - drv[k].data_ptr() for k in range(13)}
    assert gaps == {eng._pitch(n, report['chosen_extra_bytes'])}
    assert cls.numel() == n and len(drv) == 14 and day.numel() == n
    eng.synth(n, seed=2, out=(cls, drv))
    eng.run(cls, drv, day, night)
    want_d, want_n = eng.run(cls, [d.clone() for d in drv])
    eng.check()
    assert torch.equal(torch.nan_to_num(day), torch.nan_to_num(want_d))
    assert torch.equal(torch.nan_to_num(night), torch.nan_to_num(want_n))


@pytest.mark.parametrize('dtype', ['float64', 'float32'])
def test_host_path_over_many_tiles_equals_device_path(env, dtype):
    """numpy in / numpy out stages 2 Mi-pixel tiles through several slabs, one
    host thread each: every tile (and the ragged last one) must land where it
    belongs, bit-identical to the device-resident run."""
    torch, RasterEngine, table = env
    import mod16_amd
    eng = RasterEngine(table, dtype=dtype)
    n = 5 * (1 << 21) + 12345
    cls, drv = eng.synth(n, seed=61)
    day, night = eng.run(cls, drv)
    eng.check()
    h_cls, h_drv = cls.cpu().numpy(), to_np(drv)
    hd, hn = mod16_amd.evapotranspiration_raster(table, h_cls, *h_drv)
    assert hd.dtype == h_drv[0].dtype
    assert np.array_equal(hd, day.cpu().numpy(), equal_nan=True)
    assert np.array_equal(hn, night.cpu().numpy(), equal_nan=True)
    # components: the other pipeline form through the same staging
    (cd, sd, td), (cn, sn, tn) = mod16_amd.evapotranspiration_raster(table, h_cls, *h_drv, separate=True)
    sep = eng.empty(n, 6)
    eng.run(cls, drv, out_sep=sep)
    eng.check()
    for got, want in zip((cd, sd, td, cn, sn, tn), to_np(sep)):
        assert np.array_equal(got, want, equal_nan=True)


def test_more_than_two_to_the_31_pixels(env):
    """64-bit indexing end to end: a float32 raster of 2^31 + 12345 pixels (137 GB
    on the device); windows on both sides of 2^31 and at the end equal small runs
    on copies of the same pixels, and the diagnostics count every pixel once."""
    torch, RasterEngine, table = env
    eng = RasterEngine(table, dtype='float32')
    n = 2 ** 31 + 12345
    free, _ = torch.cuda.mem_get_info()
    if free < 150e9:
        pytest.skip('needs 150 GB of free device memory')
    cls, drv, day, night = eng.alloc_raster(n)
    eng.synth(n, seed=81, out=(cls, drv))
    diag = torch.zeros(8, dtype=torch.float64, device='cuda')
    eng.run(cls, drv, day, night, diag=diag)
    eng.check()
    d = diag.cpu().numpy()
    assert d[2] + d[4] == n and d[3] + d[5] == n            # n_valid + n_nan, day and night
    m = 4096
    for off in (0, 2 ** 31 - 2048, 2 ** 31 + 4, n - m):
        sub_d, sub_n = eng.run(cls[off:off + m].clone(), [x[off:off + m].clone() for x in drv])
        assert torch.equal(torch.nan_to_num(day[off:off + m]), torch.nan_to_num(sub_d)), off
        assert torch.equal(torch.nan_to_num(night[off:off + m]), torch.nan_to_num(sub_n)), off
    # the generator itself is keyed on the global pixel index: the tail equals a direct call
    c2, d2 = eng.synth(m, seed=81, pixel_offset=n - m)
    assert torch.equal(c2, cls[n - m:])
    assert torch.equal(torch.nan_to_num(d2[5]), torch.nan_to_num(drv[5][n - m:]))
    eng.check()
    # the same raster in the tiled layout (float32: 8192 pixels per tile, 262146 tiles; a tiled
    # raster holds whole 16-byte vectors: the first nt = n - 1 pixels), mixed precision as
    # well: the plain inputs make room first
    nt = n // 4 * 4
    keep = {off: (day[off:off + m].clone(), night[off:off + m].clone())
            for off in (0, 2 ** 31 - 2048, 2 ** 31 + 4, nt - m)}
    last = (float(day[nt:].double().nan_to_num().sum()), float(night[nt:].double().nan_to_num().sum()),
            int(torch.isnan(day[nt:]).sum()), int(torch.isnan(night[nt:]).sum()))
    del cls, drv, day, night
    torch.cuda.empty_cache()
    r = eng.synth_tiled(eng.alloc_tiled(nt), seed=81)
    d_tiled = torch.zeros(8, dtype=torch.float64, device='cuda')
    eng.run_tiled(r, diag=d_tiled)
    eng.check()
    t = d_tiled.cpu().numpy()
    assert t[2] + t[4] == nt and t[3] + t[5] == nt
    assert t[4] == d[4] - last[2] and t[5] == d[5] - last[3]
    assert np.isclose(t[0], d[0] - last[0], rtol=1e-12) and np.isclose(t[1], d[1] - last[1], rtol=1e-12)
    for off, (kd, kn) in keep.items():
        assert torch.equal(torch.nan_to_num(r.flat(r.day, off, off + m)), torch.nan_to_num(kd)), off
        assert torch.equal(torch.nan_to_num(r.flat(r.night, off, off + m)), torch.nan_to_num(kn)), off
    from mod16_amd import _lib
    mixed = RasterEngine(table, dtype='float32', math=_lib.MATH_MIXED)
    d_mixed = torch.zeros(8, dtype=torch.float64, device='cuda')
    mixed.run_tiled(r, diag=d_mixed)
    mixed.check()
    x = d_mixed.cpu().numpy()
    assert np.array_equal(x[2:6], t[2:6]) and np.allclose(x[:2], t[:2], rtol=1e-5)
    for off, (kd, kn) in keep.items():
        got = r.flat(r.day, off, off + m)
        assert torch.equal(torch.isnan(got), torch.isnan(kd)) and torch.equal(got == 0, kd == 0), off
    del r
    torch.cuda.empty_cache()


@pytest.mark.parametrize('dtype', ['float64', 'float32'])
def test_diagnostics_of_an_all_nan_raster(env, dtype):
    """Every pixel of an invalid class (NaN parameters): sums 0, no valid pixel, the
    maxima stay at their start value -inf; fused and stand-alone reductions agree."""
    torch, RasterEngine, table = env
    eng = RasterEngine(table, dtype=dtype)
    n = 300000
    cls, drv = eng.synth(n, seed=6)
    cls.fill_(11)
    fused = torch.zeros(8, dtype=torch.float64, device='cuda')
    day, night = eng.run(cls, drv, diag=fused)
    alone = eng.diagnostics(day, night)
    eng.check()
    assert bool(torch.isnan(day).all()) and bool(torch.isnan(night).all())
    want = np.array([0.0, 0.0, 0.0, 0.0, n, n, -np.inf, -np.inf])
    assert np.array_equal(fused.cpu().numpy(), want)
    assert np.array_equal(alone.cpu().numpy(), want)


def test_global_grid_float64(env):
    """BASELINE.json configs[2] at FULL size on one GPU: the 43200 x 21600
    float64 grid (120 GB resident; skipped when less than 150 GB of HBM is
    free). Checks that do not need a CPU pass over 933 M pixels: diagnostics
    consistent with the raster (n_valid + n_nan = n, NaN share of the
    generator), the production (FAST) kernel against the reference-order
    (EXACT) kernel on every pixel (masks identical, nothing above 1e-5, the
    north-star tolerance), and the numpy oracle on four 1200 x 1200 windows
    copied back (1e-8, masks identical)."""
    torch, RasterEngine, table = env
    from mod16_amd import _lib
    free, _ = torch.cuda.mem_get_info()
    if free < 150e9:
        pytest.skip('needs 150 GB of free HBM, %.0f GB available' % (free / 1e9))
    rows, cols = 21600, 43200
    n = rows * cols
    eng = RasterEngine(table)
    cls, drv, day, night = eng.alloc_raster(n, 512 << 20)
    eng.synth(n, seed=16, out=(cls, drv))
    diag = torch.zeros(8, dtype=torch.float64, device='cuda')
    eng.run(cls, drv, day, night, diag=diag)
    eng.check()
    g = diag.cpu().numpy()
    assert g[2] + g[4] == n and g[3] + g[5] == n
    assert 0.02 < g[4] / n < 0.06                     # invalid classes + NaN fills of the generator
    assert g[0] > 0 and g[1] > 0 and np.isfinite(g[6]) and np.isfinite(g[7])
    # every pixel: FAST against EXACT on the device, in slices (bounded temporaries)
    exact = RasterEngine(table, math=_lib.MATH_EXACT)
    eday, enight = exact.run(cls, drv)
    exact.check()
    worst, above = 0.0, 0
    step = 1 << 27
    for got, ref in ((day, eday), (night, enight)):
        for lo in range(0, n, step):
            a, b = got[lo:lo + step], ref[lo:lo + step]
            assert torch.equal(torch.isnan(a), torch.isnan(b))
            assert torch.equal(a == 0, b == 0)
            err = torch.nan_to_num_((a - b).abs_().div_(b.abs()), nan=0.0, posinf=0.0)
            worst = max(worst, float(err.max()))
            above += int((err > 1e-5).sum())
            del err
    assert above == 0 and worst < 1e-7, (above, worst)
    del eday, enight
    # the oracle on four windows (start, two inside, end)
    bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    m = 1200 * 1200
    for s0 in (0, (n // 3) // 4 * 4, (2 * n // 3) // 4 * 4, n - m):
        h_cls = cls[s0:s0 + m].cpu().numpy()
        h_drv = [d[s0:s0 + m].cpu().numpy() for d in drv]
        want = oracle.evapotranspiration_raster(bplut, h_cls, *h_drv)
        assert_parity(day[s0:s0 + m].cpu().numpy(), want[0], 1e-8, 'day @%d' % s0)
        assert_parity(night[s0:s0 + m].cpu().numpy(), want[1], 1e-8, 'night @%d' % s0)
    # the same grid in the engine's TILED layout (the layout bench.py times): the plain inputs
    # make room, the generator writes the identical field into the tiled raster, and every pixel
    # of its outputs must equal the plain run's bit for bit (the full-size tile / row address
    # arithmetic: 227813 tiles, offsets beyond 2^36 bytes); diagnostics to 1e-12
    del cls, drv, h_cls, h_drv
    torch.cuda.empty_cache()
    r = eng.synth_tiled(eng.alloc_tiled(n), seed=16)
    d_tiled = torch.zeros(8, dtype=torch.float64, device='cuda')
    eng.run_tiled(r, diag=d_tiled)
    eng.check()
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        assert torch.equal(torch.nan_to_num(r.flat(r.day, lo, hi), nan=-7.0), torch.nan_to_num(day[lo:hi], nan=-7.0)), lo
        assert torch.equal(torch.nan_to_num(r.flat(r.night, lo, hi), nan=-7.0), torch.nan_to_num(night[lo:hi], nan=-7.0)), lo
    t = d_tiled.cpu().numpy()
    assert np.array_equal(t[2:], g[2:]) and np.allclose(t[:2], g[:2], rtol=1e-12, atol=0)
    # a window of the tiled INPUTS against the oracle too (what the kernel read is what was written)
    s0 = (n // 2) // 4 * 4
    want = oracle.evapotranspiration_raster(
        bplut, r.flat(r.cls, s0, s0 + m).cpu().numpy(), *[r.flat(d, s0, s0 + m).cpu().numpy() for d in r.drivers])
    assert_parity(r.flat(r.day, s0, s0 + m).cpu().numpy(), want[0], 1e-8, 'tiled day')
    assert_parity(r.flat(r.night, s0, s0 + m).cpu().numpy(), want[1], 1e-8, 'tiled night')
    del r, day, night
    torch.cuda.empty_cache()


def test_pinned_pool_is_bounded(env):
    """The page-locked memory behind result arrays is bounded: beyond MAX_LIVE bytes handed out the
    results are plain numpy arrays (a caller keeping many rasters must not pin RAM without limit) and
    the pool says so once; idle blocks are kept up to the sizes of the last results handed out (one
    step of a time loop finds its blocks again) or MAX_CACHED, whichever is larger; trim() frees the
    rest. give() runs from a finalizer -- also inside a collection that take() triggers while it
    holds the (re-entrant) lock."""
    import gc
    import warnings
    from mod16_amd import _lib
    pool = _lib._PinnedPool()
    pool.MAX_LIVE, pool.MAX_CACHED = 5 << 20, 2 << 20
    held = [pool.empty((1 << 20,), np.uint8) for _ in range(5)]        # 5 MiB: all page-locked
    assert all(a.base is not None for a in held) and pool.live == 5 << 20
    with pytest.warns(RuntimeWarning, match='did not fit the page-locked pool'):
        extra = pool.empty((1 << 20,), np.uint8)                       # over the bound: ordinary, and said
    assert extra.base is None and pool.live == 5 << 20 and pool.fallbacks == 1
    with warnings.catch_warnings():
        warnings.simplefilter('error')                                 # ... once
        extra2 = pool.empty((1 << 20,), np.uint8)
    assert extra2.base is None and pool.fallbacks == 2
    del held
    gc.collect()
    assert pool.cached == 5 << 20 and pool.live == 5 << 20             # the last results' blocks stay idle
    again = pool.empty((1 << 20,), np.uint8)
    assert again.base is not None and pool.cached == 4 << 20           # an idle block came back
    pool.trim(keep=1 << 20)
    assert pool.cached == 1 << 20 and pool.live == 2 << 20
    pool.trim()
    assert pool.cached == 0 and pool.live == 1 << 20
    # with nothing recent, idle blocks beyond MAX_CACHED are freed when they come back
    more = [again] + [pool.empty((1 << 20,), np.uint8) for _ in range(3)]
    del again
    pool.recent.clear()
    del more
    gc.collect()
    assert pool.cached == 2 << 20 and pool.live == 2 << 20
    # an idle block of another size makes room when the bound is reached
    big = pool.empty((4 << 20,), np.uint8)
    big2 = pool.empty((1 << 20,), np.uint8)
    assert big.base is not None and big2.base is not None and pool.live <= 5 << 20
    # a finalizer firing inside take(): the lock is re-entrant
    with pool.lock:
        pool.give(0, 0)


def test_flat_always_copies(env):
    """TiledRaster.flat() hands back a copy whether the window lies inside one tile (where the
    strided view happens to be contiguous) or spans several."""
    torch, RasterEngine, table = env
    eng = RasterEngine(table)
    r = eng.synth_tiled(eng.alloc_tiled(4096 * 3), seed=3)
    inside = r.flat(r.drivers[5], 10, 500)
    across = r.flat(r.drivers[5], 4000, 4200)
    was = inside.clone(), across.clone()
    r.drivers[5].fill_(-1.0)
    assert torch.equal(inside, was[0]) and torch.equal(across, was[1])


def test_bench_reads_the_sensors_of_its_own_device(env):
    """The bench line's clock / power figures come from the hwmon files of the card the process
    computes on (a box shows all the cards of its host): found by PCI address, plausible while a
    kernel runs. No such files on a box (a different kernel driver layout): nothing to check."""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    torch, RasterEngine, table = env
    eng = RasterEngine(table)
    hw = bench.device_sensors(torch)
    if hw is None:
        pytest.skip('no hwmon sensors for this device on this box')
    try:
        cap = float(open(os.path.join(hw, 'power1_cap')).read()) * 1e-6
        float(open(os.path.join(hw, 'power1_input')).read())
        float(open(os.path.join(hw, 'freq1_input')).read())
    except (OSError, ValueError):
        pytest.skip('the hwmon files of this device cannot be read on this box')
    assert 100 < cap < 5000
    n = 1 << 24
    ras = eng.synth_tiled(eng.alloc_tiled(n), seed=3)
    diag = torch.zeros(8, dtype=torch.float64, device='cuda')
    step = eng.bind_tiled(ras, diag)
    res = bench.device_under_load(torch, step, 600, True)      # ~0.25 s of launches
    assert res is not None and res['samples'] >= 1
    assert 90 <= res['sclk_mhz'] <= 3000 and 0 < res['power_w'] <= 1.2 * cap
    assert res['power_cap_w'] == cap


def test_rank_order_fold_of_gathered_diagnostics(env):
    """mod16_fold_diag (the kernel behind dist.allreduce_diag's all-gather): sums and counts added
    in rank order -- the bits of a left-to-right float64 sum --, maxima maximised; 1, 2 and 8 ranks,
    in place on a vector that is also row 0 of nothing (the gather buffer is separate)."""
    torch, RasterEngine, table = env
    eng = RasterEngine(table)
    rng = np.random.default_rng(3)
    for world in (1, 2, 8):
        block = rng.normal(0, 1e6, (world, 8))
        block[:, 2:6] = rng.integers(0, 1 << 30, (world, 4))
        want = np.empty(8)
        for k in range(6):
            acc = block[0, k]
            for r in range(1, world):
                acc = acc + block[r, k]
            want[k] = acc
        want[6:] = block[:, 6:].max(axis=0)
        gathered = torch.from_numpy(block).cuda().reshape(-1)
        diag = torch.zeros(8, dtype=torch.float64, device='cuda')
        eng.fold_ranks(gathered, world, diag)
        torch.cuda.synchronize()
        assert np.array_equal(diag.cpu().numpy(), want), world
    with pytest.raises(Exception):
        eng.fold_ranks(gathered, 0, diag)


def test_graph_as_the_first_gpu_call_of_a_process(env):
    """A captured step built as the very first GPU work of a fresh process (no launch of any
    library kernel before hipStreamBeginCapture: lazy code-object loading, the scalar-driver
    staging and the workspace set-up all happen on the capture path), plain arrays with and
    without a broadcast scalar driver, and a tiled raster; replayed twice and checked against
    direct launches. A child process: this one has long initialised the device."""
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import sys
sys.path.insert(0, %r)
import numpy as np
import torch
from mod16_amd import _lib
from mod16_amd.raster import RasterEngine
from mod16_amd.utils import restore_bplut, bplut_table
from mod16_amd.models import COLLECTION61_BPLUT
table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
which = sys.argv[1]
eng = RasterEngine(table)
n = 3_000_000 if which != 'tiled_large' else 40_000_000
same = lambda a, b: torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))
diag = torch.zeros(8, dtype=torch.float64, device='cuda')
if which.startswith('tiled'):
    r = eng.alloc_tiled(n)
    r.slab.zero_()                       # torch kernels only so far
    for k, v in enumerate((-50., -30., 150., 0., .2, 293., 290., 285., 285., 1000., 500., 1e5, .5, 1.5)):
        r.drivers[k].fill_(v)
    r.cls.fill_(7)
    step = eng.bind_tiled(r, diag)       # FIRST library launch sequence: inside the capture
    step(); step()
    torch.cuda.synchronize()
    d2 = torch.zeros(8, dtype=torch.float64, device='cuda')
    day, night = r.flat(r.day), r.flat(r.night)
    eng.run_tiled(r, diag=d2)
    torch.cuda.synchronize()
    assert same(day, r.flat(r.day)) and same(night, r.flat(r.night)) and same(diag, d2)
    assert float(diag[2]) == n and 1e-7 < float(day[0]) < 1e-3
else:
    cls = torch.full((n,), 7, dtype=torch.uint8, device='cuda')
    vals = (-50., -30., 150., 0., .2, 293., 290., 285., 285., 1000., 500., 1e5, .5, 1.5)
    drv = [torch.full((n,), v, dtype=torch.float64, device='cuda') for v in vals]
    if which == 'scalar':
        drv[3] = 0.0                     # sw_rad_night as a broadcast scalar (notebook cell 17)
    day, night = eng.empty(n, 2)
    step = eng.bind(cls, drv, day, night, diag, graph=True)
    step(); step()
    torch.cuda.synchronize()
    d2 = torch.zeros(8, dtype=torch.float64, device='cuda')
    rd, rn = eng.run(cls, drv, diag=d2)
    eng.check()
    assert same(day, rd) and same(night, rn) and same(diag, d2), which
    assert float(diag[2]) == n
print('ok', which)
''' % ROOT
    for which in ('dense', 'scalar', 'tiled', 'tiled_large'):
        proc = subprocess.run([sys.executable, '-c', code, which], capture_output=True, text=True, timeout=600)
        assert proc.returncode == 0 and ('ok ' + which) in proc.stdout, (which, proc.stdout[-1000:], proc.stderr[-3000:])


@pytest.mark.parametrize('math', ['fast', 'mixed'])
def test_series_from_host_in_the_light_input_form(env, math):
    """RasterEngine.run_series_host: per step the 14 float32 raw fields and the uint8 fPAR / LAI of a
    host record copied (page-locked memory, a second stream, tile-wide rows into the slot's pitch)
    into a two-slot ring of FORM_RAW tiled rasters under the kernel of the step before; the class
    raster stays resident. Every step's outputs equal the plain raw-driver call on that record's
    arrays bit for bit, and the oracle's (the reference's pre-processing + forward run) on a step."""
    from mod16_amd import _lib
    torch, RasterEngine, table = env
    m = {'fast': _lib.MATH_FAST, 'mixed': _lib.MATH_MIXED}[math]
    eng = RasterEngine(table, dtype='float32', math=m)
    n = 8192 * 37
    ring = [eng.alloc_tiled(n, form=_lib.FORM_RAW) for _ in range(2)]
    g = torch.Generator(device='cuda').manual_seed(3)
    pin = lambda x: torch.empty(x.shape, dtype=x.dtype, pin_memory=True).copy_(x)
    recs, dev = [], []
    for k in range(3):
        cls, drv = eng.synth(n, seed=5, step=k)
        u = lambda lo, hi: torch.empty(n, dtype=torch.float32, device='cuda').uniform_(lo, hi, generator=g)
        raw = drv[:9] + [u(0.001, 0.02), u(0.001, 0.02), u(7e4, 1.0134e5), u(7e4, 1.0134e5), u(0, 3500)]
        fpar = torch.randint(0, 101, (n,), dtype=torch.uint8, device='cuda', generator=g)
        lai = torch.randint(0, 71, (n,), dtype=torch.uint8, device='cuda', generator=g)
        fpar[::97] = 255
        raw[13][5::1001] = 5e4                       # an elevation outside the domain: revisited in the reference's order
        recs.append({'wide': [pin(x) for x in raw], 'bytes': [None, pin(fpar), pin(lai)]})
        dev.append((raw, fpar, lai))
    for r in ring:
        r.put(r.bytes[0], cls)
    seen = {}

    def grab(s, slot):
        seen[s] = (slot.flat(slot.outs[0]), slot.flat(slot.outs[1]))

    last = eng.run_series_host(ring, recs, 7, on_step=grab)
    eng.check()
    assert last is ring[0] and sorted(seen) == list(range(7))
    same = lambda a, b: torch.equal(torch.nan_to_num(a, nan=-7.0, posinf=9e30, neginf=-9e30),
                                    torch.nan_to_num(b, nan=-7.0, posinf=9e30, neginf=-9e30))
    for s in range(7):
        raw, fpar, lai = dev[s % 3]
        want = eng.run_raw(cls, raw, fpar, lai)
        assert same(seen[s][0], want[0]) and same(seen[s][1], want[1]), s
    raw, fpar, lai = dev[5 % 3]
    bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    with np.errstate(all='ignore'):
        o = oracle.evapotranspiration_raw(bplut, cls.cpu().numpy(), [x.cpu().numpy().astype(np.float64) for x in raw],
                                          fpar.cpu().numpy(), lai.cpu().numpy())
    for got, w in zip(seen[5], o):
        if math == 'mixed':
            from parity import assert_mixed_parity
            with np.errstate(over='ignore'):
                assert_mixed_parity(got.cpu().numpy(), w.astype(np.float32), 'raw series, mixed')
        else:
            with np.errstate(over='ignore'):
                assert_parity(got.cpu().numpy(), w.astype(np.float32), 1e-6, 'raw series')


def test_a_launch_behind_one_on_a_stream_that_is_gone(env):
    """Launches on DIFFERENT streams share the context's workspace and are ordered by the
    library. A context that has only ever seen one stream records nothing behind its launches;
    the first launch on another stream waits for the device -- not for the earlier stream, which
    its owner may have destroyed by then (recording an event on a destroyed stream faults) --
    and from then on every launch leaves an event, recorded while its stream is certainly alive
    (ws_acquire / ws_release, mod16_capi.hip). Both phases with a destroyed stream in between."""
    import ctypes
    import torch as _t
    torch, RasterEngine, table = env
    eng = RasterEngine(table)
    n = 1200 * 1200
    cls, drv, day, night = eng.alloc_raster(n)
    eng.synth(n, seed=8, out=(cls, drv))
    # the HIP runtime torch has loaded (another copy of the library would be another runtime)
    hip = ctypes.CDLL(__import__('os').path.join(__import__('os').path.dirname(_t.__file__), 'lib', 'libamdhip64.so'))
    vecs = [torch.zeros(8, dtype=torch.float64, device='cuda') for _ in range(6)]
    torch.cuda.synchronize()
    k = 0
    for phase in range(2):
        raw = ctypes.c_void_p()
        assert hip.hipStreamCreate(ctypes.byref(raw)) == 0
        gone = torch.cuda.ExternalStream(raw.value)
        with torch.cuda.stream(gone):
            eng.run(cls, drv, day, night, diag=vecs[k])
        gone.synchronize()
        del gone
        assert hip.hipStreamDestroy(raw) == 0
        eng.run(cls, drv, day, night, diag=vecs[k + 1])      # the current stream: behind a launch on `raw`
        eng.run(cls, drv, day, night, diag=vecs[k + 2])
        torch.cuda.synchronize()
        eng.check()
        k += 3
    assert float(vecs[0][2]) > 0
    for v in vecs[1:]:
        assert torch.equal(vecs[0], v)
