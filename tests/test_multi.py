"""Several GPUs behind one call of the numpy entry points (mod16_amd/multi.py; SURVEY.md 8e for
the PCIe-bound HOST paths of reference mod16/__init__.py:675-793): the partition and the worker
plumbing on the CPU; on the GPU, devices=[0, 0] -- two contexts on the one card -- against
devices=[0] and the plain call: outputs and diagnostics bit for bit, error paths intact."""
import threading

import numpy as np
import pytest

from oracle import mod16_oracle as oracle
from oracle import synth


# ------------------------------------------------------------------ CPU: the partition
@pytest.mark.parametrize('n,parts,align', [
    (0, 1, 4), (0, 3, 1 << 21), (1, 4, 1 << 21), (5 * (1 << 21) + 12345, 2, 1 << 21),
    (5 * (1 << 21) + 12345, 8, 1 << 21), (933120000, 8, 1 << 21), (10, 3, 4), (7, 8, 1), (64, 8, 8)])
def test_shards_tile_the_range_at_aligned_boundaries(n, parts, align):
    from mod16_amd import multi
    cuts = multi.shards(n, parts, align)
    assert len(cuts) == parts
    pos = 0
    for off, m in cuts:
        assert off == pos and m >= 0           # contiguous, in order
        assert off % align == 0 or m == 0      # every boundary on a unit (empty shards sit at n)
        pos += m
    assert pos == n
    units = [-(-m // align) for _, m in cuts]
    assert max(units) - min(units) <= 1        # balanced to one unit
    # more devices than units: the trailing shards are empty, nothing is lost
    if -(-n // align) < parts:
        assert [m for _, m in cuts].count(0) == parts - (-(-n // align))


def test_shards_argument_errors():
    from mod16_amd import multi
    with pytest.raises(ValueError):
        multi.shards(-1, 2)
    with pytest.raises(ValueError):
        multi.shards(10, 0)
    assert multi.device_list(None) is None
    assert multi.device_list((1, 1, 0)) == [1, 1, 0]
    with pytest.raises(ValueError):
        multi.device_list([])
    with pytest.raises(ValueError):
        multi.device_list([0, -1])


def test_fold_is_the_rank_order_rule():
    """mod16_fold_diag_host: sums and counts added first to last, maxima maximised -- the rule of
    the device-side fold behind the all-gather (mod16_fold_diag)."""
    from mod16_amd import multi
    rng = np.random.default_rng(4)
    parts = rng.normal(size=(37, 8)) * 10.0 ** rng.integers(-8, 8, size=(37, 8))
    parts[5, 6:] = -np.inf                       # an all-NaN tile's maxima
    want = np.empty(8)
    for k in range(6):
        acc = parts[0, k]
        for r in range(1, 37):
            acc = acc + parts[r, k]
        want[k] = acc
    want[6:] = parts[:, 6:].max(axis=0)
    assert np.array_equal(multi.fold_diag(parts), want)
    assert np.array_equal(multi.fold_diag(parts[:1]), parts[0])
    with pytest.raises(ValueError):
        multi.fold_diag(np.empty((0, 8)))


def test_run_keeps_list_order_and_raises_the_first_failure(monkeypatch):
    """The worker plumbing without a GPU: one thread per list entry (a device listed twice gets
    two), results in list order, every shard waited for, the first failure in list order raised
    as it is."""
    from mod16_amd import _lib, multi
    made = []

    def fake_context(device):
        made.append((threading.current_thread().name, device))
        return ('ctx', device, threading.get_ident())
    monkeypatch.setattr(_lib, 'context', fake_context)
    got = multi.run([3, 3, 5], lambda i, ctx: (i, ctx[1], ctx[2]))
    assert [g[:2] for g in got] == [(0, 3), (1, 3), (2, 5)]
    assert len({g[2] for g in got}) == 3 and threading.get_ident() not in {g[2] for g in got}
    again = multi.run([3, 3, 5], lambda i, ctx: ctx[2])
    assert again == [g[2] for g in got]          # the same threads serve the next call
    done = []

    def fn(i, ctx):
        if i == 1:
            raise IndexError('class code')
        if i == 2:
            raise RuntimeError('later shard')
        done.append(i)
    with pytest.raises(IndexError, match='class code'):
        multi.run([3, 3, 5], fn)
    assert done == [0]
    # shutdown(): the workers end (their contexts go with them) and the next call starts new ones
    old = [t for t in threading.enumerate() if t.name.startswith('mod16-dev')]
    assert {t.ident for t in old} == set(again) and multi.shutdown() == 3
    assert not any(t.is_alive() for t in old)
    assert not [t for t in threading.enumerate() if t.name.startswith('mod16-dev')]
    assert len(set(multi.run([3, 3, 5], lambda i, ctx: ctx[2]))) == 3
    new = [t for t in threading.enumerate() if t.name.startswith('mod16-dev')]
    assert len(new) == 3 and not any(t in old for t in new)
    assert multi.shutdown() == 3 and multi.shutdown() == 0


def test_entry_points_fail_loudly_without_a_gpu():
    """devices= does not open a CPU path: without an MI355X the sharded call raises like the plain one."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    import mod16_amd
    from mod16_amd import _lib
    from mod16_amd.utils import restore_bplut, bplut_table
    from mod16_amd.models import COLLECTION61_BPLUT
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    cls, drv = synth.drivers((8, 8), seed=1)
    with pytest.raises(_lib.Mod16Error, match='no CPU fallback'):
        mod16_amd.evapotranspiration_raster(table, cls, *drv, devices=[0, 0])


# ------------------------------------------------------------------ GPU
@pytest.fixture(scope='module')
def table():
    from mod16_amd.utils import restore_bplut, bplut_table
    from mod16_amd.models import COLLECTION61_BPLUT
    return bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_two_contexts_on_one_gpu_give_the_bits_of_one(table, dtype):
    """evapotranspiration_raster over 5 staging tiles + a ragged end: devices=[0, 0] (and three
    entries, and more entries than tiles) against devices=[0] and the call without devices --
    outputs identical, diagnostics identical to the last bit, and right (numpy on the outputs)."""
    import mod16_amd
    from mod16_amd import multi
    n = 5 * multi.host_tile() + 12345
    cls, drv = synth.drivers((n,), seed=23)
    drv = [d.astype(dtype) for d in drv]
    plain = mod16_amd.evapotranspiration_raster(table, cls, *drv)
    one = mod16_amd.evapotranspiration_raster(table, cls, *drv, devices=[0], diagnostics=True)
    assert same(one[0], plain[0]) and same(one[1], plain[1]) and one[0].dtype == dtype
    for devices in ([0, 0], [0, 0, 0], [0] * 8):
        got = mod16_amd.evapotranspiration_raster(table, cls, *drv, devices=devices, diagnostics=True)
        assert same(got[0], one[0]) and same(got[1], one[1]), devices
        assert np.array_equal(got[2], one[2]), (devices, got[2], one[2])
    diag = one[2]
    for k, a in enumerate(one[:2]):
        a = a.astype(np.float64)
        assert diag[2 + k] == np.isfinite(a).sum() + np.isinf(a).sum() and diag[4 + k] == np.isnan(a).sum()
        assert diag[2 + k] + diag[4 + k] == n
        np.testing.assert_allclose(diag[k], np.nansum(a), rtol=1e-12)
        assert diag[6 + k] == np.nanmax(a)
    # the oracle on a window across the cut between the two shards
    lo = 3 * multi.host_tile() - 500
    tab = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    with np.errstate(all='ignore'):
        want = oracle.evapotranspiration_raster(tab, cls[lo:lo + 1000], *[d[lo:lo + 1000].astype(np.float64) for d in drv])
    from parity import assert_parity
    for g, w in zip(one[:2], want):
        assert_parity(g[lo:lo + 1000], w.astype(dtype), 1e-8 if dtype == np.float64 else 1e-6, 'window')


@pytest.mark.gpu
def test_page_locked_driver_arrays(table):
    """mod16_amd.pinned_empty: driver arrays in page-locked memory (pure-DMA uploads) are ordinary
    numpy arrays to the call -- same results, single device and sharded; small ones are plain arrays."""
    import mod16_amd
    from mod16_amd import multi
    n = multi.host_tile() + 999
    cls, drv = synth.drivers((n,), seed=77)
    want = mod16_amd.evapotranspiration_raster(table, cls, *drv)
    p_drv = [mod16_amd.pinned_empty(n) for _ in range(14)]
    p_cls = mod16_amd.pinned_empty((n,), np.uint8)
    assert all(a.base is not None and a.flags.c_contiguous and a.flags.writeable for a in p_drv)
    p_cls[:] = cls
    for a, b in zip(p_drv, drv):
        a[:] = b
    for devices in (None, [0, 0]):
        got = mod16_amd.evapotranspiration_raster(table, p_cls, *p_drv, devices=devices)
        assert same(got[0], want[0]) and same(got[1], want[1])
    small = mod16_amd.pinned_empty((10, 3), np.float32)
    assert small.shape == (10, 3) and small.dtype == np.float32 and small.base is None


@pytest.mark.gpu
def test_sharded_components_pet_and_per_pixel_parameters(table):
    """The other shapes of the call through the shards: separate=True, pet=True, the class
    surface with per-pixel parameter arrays (MOD16(params, devices=...)), a 2-D raster, and a
    (T, N) call with (N,) rows, which is cut at whole rows."""
    import mod16_amd
    from mod16_amd import multi
    n = 2 * multi.host_tile() + 4321
    cls, drv = synth.drivers((n,), seed=5)
    a = mod16_amd.evapotranspiration_raster(table, cls, *drv, separate=True)
    b = mod16_amd.evapotranspiration_raster(table, cls, *drv, separate=True, devices=[0, 0])
    assert all(same(x, y) for p, q in zip(a, b) for x, y in zip(p, q))
    a = mod16_amd.evapotranspiration_raster(table, cls, *drv, pet=True)
    b = mod16_amd.evapotranspiration_raster(table, cls, *drv, pet=True, devices=[0, 0, 0])
    assert len(b) == 4 and all(same(x, y) for x, y in zip(a, b))
    params = {k: table[:, j][cls] for j, k in enumerate(mod16_amd.MOD16.required_parameters)}
    a = mod16_amd.MOD16(params).evapotranspiration(*drv)
    b = mod16_amd.MOD16(params, devices=[0, 0]).evapotranspiration(*drv)
    assert same(a[0], b[0]) and same(a[1], b[1])
    c = mod16_amd.evapotranspiration_raster(table, cls, *drv)
    assert same(a[0], c[0]) and same(a[1], c[1])
    # 2-D raster: flattened row-major, cut at tile boundaries (not at rows)
    rows = 3
    cls2, drv2 = synth.drivers((rows, n // rows), seed=6)
    a = mod16_amd.evapotranspiration_raster(table, cls2, *drv2)
    b = mod16_amd.evapotranspiration_raster(table, cls2, *drv2, devices=[0, 0])
    assert b[0].shape == (rows, n // rows) and same(a[0], b[0]) and same(a[1], b[1])
    # (T, N) drivers against (N,) per-site arrays: two-level broadcasting, cut at whole rows
    T, N = 7, 1000
    cls3, drv3 = synth.drivers((T, N), seed=7)
    drv3 = list(drv3)
    drv3[7], drv3[11] = drv3[7][0], drv3[11][0]          # temp_annual, pressure per site
    site = cls3[0]
    a = mod16_amd.evapotranspiration_raster(table, site, *drv3)
    b = mod16_amd.evapotranspiration_raster(table, site, *drv3, devices=[0, 0, 0])
    assert a[0].shape == (T, N) and same(a[0], b[0]) and same(a[1], b[1])


@pytest.mark.gpu
def test_sharded_error_paths(table):
    """A class code >= 13 in the LAST shard only still raises IndexError (the numpy gather's
    error) and the other shards have run; an empty raster and more devices than pixels work."""
    import mod16_amd
    from mod16_amd import multi
    n = 2 * multi.host_tile() + 100
    cls, drv = synth.drivers((n,), seed=9)
    bad = cls.copy()
    bad[-3] = 13
    with pytest.raises(IndexError):
        mod16_amd.evapotranspiration_raster(table, bad, *drv, devices=[0, 0, 0])
    with pytest.raises(IndexError):
        mod16_amd.evapotranspiration_raster(table, bad, *drv, devices=[0, 0], diagnostics=True)
    # ... and the contexts are usable afterwards
    a = mod16_amd.evapotranspiration_raster(table, cls, *drv)
    b = mod16_amd.evapotranspiration_raster(table, cls, *drv, devices=[0, 0, 0])
    assert same(a[0], b[0]) and same(a[1], b[1])
    e = mod16_amd.evapotranspiration_raster(table, cls[:0], *[d[:0] for d in drv], devices=[0, 0], diagnostics=True)
    assert e[0].shape == (0,) and e[2][2] == 0 and e[2][6] == -np.inf
    s = mod16_amd.evapotranspiration_raster(table, cls[:3], *[d[:3] for d in drv], devices=[0] * 5, diagnostics=True)
    assert same(s[0], a[0][:3]) and s[2][2] + s[2][4] == 3
    with pytest.raises(ValueError):
        mod16_amd.evapotranspiration_raster(table, cls, *drv, devices=[])
    with pytest.raises(ValueError):
        mod16_amd.evapotranspiration_raster(table, cls, *drv, separate=True, diagnostics=True)
    with pytest.raises(mod16_amd._lib.Mod16Error):
        mod16_amd.evapotranspiration_raster(table, cls, *drv, devices=[0, 99])


@pytest.mark.gpu
def test_sharded_raw_drivers(table):
    """evapotranspiration_raw(devices=[0, 0]): the raw-driver form cut at staging tiles."""
    import mod16_amd
    from mod16_amd import multi
    rng = np.random.default_rng(12)
    n = multi.host_tile() + 7777
    cls, drv = synth.drivers((n,), seed=31, special=False)
    raw = list(drv[:9]) + [rng.uniform(0.001, 0.02, n), rng.uniform(0.001, 0.02, n),
                           rng.uniform(7e4, 1.0134e5, n), rng.uniform(7e4, 1.0134e5, n), rng.uniform(-50, 3500, n)]
    fpar = rng.integers(0, 101, n).astype(np.uint8)
    lai = rng.integers(0, 71, n).astype(np.uint8)
    fpar[::97] = 255
    hours = rng.uniform(8, 16, n)
    a = mod16_amd.evapotranspiration_raw(table, cls, *raw, fpar, lai, day_hours=hours)
    b = mod16_amd.evapotranspiration_raw(table, cls, *raw, fpar, lai, day_hours=hours, devices=[0, 0])
    assert len(b) == 3 and all(same(x, y) for x, y in zip(a, b))
    c = mod16_amd.evapotranspiration_raw(table, cls, *raw, fpar, lai, day_hours=12.0, devices=[0, 0, 0])
    d = mod16_amd.evapotranspiration_raw(table, cls, *raw, fpar, lai, day_hours=12.0)
    assert all(same(x, y) for x, y in zip(c, d))


@pytest.mark.gpu
def test_sharded_store_writes_the_same_files(tmp_path, table):
    """io.run_store(devices=[0, 0]): pipelines on both list entries take tile jobs from the one
    queue; the output files are the bytes of the single-device run."""
    from mod16_amd import io
    from test_io import fill_store
    T, N = 2, 5 * 32768 + 77
    outs = []
    for devices in (None, [0, 0]):
        store = io.RasterStore.create(str(tmp_path / ('s%d' % len(outs))), T, N, np.float32)
        fill_store(store, seed=13)
        report = io.run_store(table, store.root, tile_pixels=32768, workers=2, devices=devices)
        assert report['devices'] == ([0] if devices is None else devices)
        assert report['workers'] == 2 * len(report['devices'])
        outs.append([np.array(store.array(io.OUT_DAY)), np.array(store.array(io.OUT_NIGHT))])
    assert same(outs[0][0], outs[1][0]) and same(outs[0][1], outs[1][1])
    assert np.isfinite(outs[0][0]).any()


@pytest.mark.gpu
def test_sharded_series_from_host(table):
    """raster.ShardedSeries: the HOST ingest series with the raster's tiles dealt over two
    contexts of the one GPU equals RasterEngine.run_series_host on the whole raster, step by step."""
    import torch
    from mod16_amd import _lib
    from mod16_amd.raster import RasterEngine, ShardedSeries
    eng = RasterEngine(table, dtype='float32')
    n = 8192 * 11 + 64                       # 12 tiles, the last one ragged
    g = torch.Generator(device='cuda').manual_seed(3)
    pin = lambda x: torch.empty(x.shape, dtype=x.dtype, pin_memory=True).copy_(x)
    P = eng.TILE_BYTES // 4
    padded = -(-n // P) * P

    def pad(x):
        out = torch.zeros(padded, dtype=x.dtype, device='cuda')
        out[:n] = x
        return out
    recs = []
    for k in range(2):
        cls, drv = eng.synth(n, seed=5, step=k)
        u = lambda lo, hi: torch.empty(n, dtype=torch.float32, device='cuda').uniform_(lo, hi, generator=g)
        raw = drv[:9] + [u(0.001, 0.02), u(0.001, 0.02), u(7e4, 1.0134e5), u(7e4, 1.0134e5), u(0, 3500)]
        fpar = torch.randint(0, 101, (n,), dtype=torch.uint8, device='cuda', generator=g)
        lai = torch.randint(0, 71, (n,), dtype=torch.uint8, device='cuda', generator=g)
        recs.append({'wide': [pin(pad(x)) for x in raw], 'bytes': [pin(pad(cls)), pin(pad(fpar)), pin(pad(lai))]})
    steps = 5
    ring = [eng.alloc_tiled(n, form=_lib.FORM_RAW) for _ in range(2)]
    want = {}
    eng.run_series_host(ring, recs, steps, on_step=lambda s, slot: want.__setitem__(s, (slot.flat(slot.outs[0]), slot.flat(slot.outs[1]))))
    eng.check()
    for devices in ([0], [0, 0], [0] * 20):
        series = ShardedSeries(table, n, devices)
        assert sum(p['n'] for p in series.parts) == n and len(series.parts) == min(len(devices), 12)
        seen = {}
        lock = threading.Lock()

        def grab(part, s, slot):
            day, night = slot.flat(slot.outs[0]), slot.flat(slot.outs[1])
            with lock:
                seen[(part['offset'], s)] = (day, night)
        slots = series.run_host(recs, steps, on_step=grab)
        for s in range(steps):
            for k in range(2):
                got = torch.cat([seen[(p['offset'], s)][k] for p in series.parts])
                assert torch.equal(torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(want[s][k], nan=-7.0)), (devices, s, k)
        host = torch.empty(n, dtype=torch.float32, pin_memory=True)
        series.read(slots, 1, host)
        assert torch.equal(torch.nan_to_num(host, nan=-7.0), torch.nan_to_num(want[steps - 1][1].cpu(), nan=-7.0))
