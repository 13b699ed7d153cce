"""Rasters on disk (mod16_amd/io.py): the store's files and
positional I/O on the CPU; the streamed forward run against the oracle on the GPU."""
import numpy as np
import pytest

from oracle import mod16_oracle as oracle


def fill_store(store, seed):
    """Random raw fields in physical ranges (as tests/golden/make_golden.py F8)."""
    from mod16_amd import io
    from oracle import synth
    rng = np.random.default_rng(seed)
    T, N = store.n_steps, store.n_pixels
    cls, drv = synth.drivers((T, N), seed=seed, special=False)
    dt = store.dtype
    raw = {0: drv[0], 1: drv[1], 2: drv[2], 4: drv[4], 5: drv[5], 6: drv[6], 8: drv[8],
           9: rng.uniform(0.001, 0.02, (T, N)), 10: rng.uniform(0.001, 0.02, (T, N)),
           11: rng.uniform(70000, 101340, (T, N)), 12: rng.uniform(70000, 101340, (T, N))}
    for idx, name in io.DYNAMIC_FIELDS:
        store.array(name, 'r+')[:] = raw[idx].astype(dt)
    store.array('MERRA2/T10M_annual', 'r+')[:] = drv[7][0].astype(dt)
    store.array('state/elevation_m', 'r+')[:] = rng.uniform(-50, 3500, N).astype(dt)
    fpar = rng.integers(0, 101, (T, N)).astype(np.uint8)
    lai = rng.integers(0, 71, (T, N)).astype(np.uint8)
    fpar[0, :4] = (249, 250, 255, 0)
    lai[1, :4] = (249, 253, 255, 0)
    store.array(io.FPAR, 'r+')[:] = fpar
    store.array(io.LAI, 'r+')[:] = lai
    store.array(io.PFT, 'r+')[:] = cls[0]


def test_store_files_and_positional_io(tmp_path):
    from mod16_amd import io
    store = io.RasterStore.create(str(tmp_path / 's'), 3, 1000, np.float32)
    assert (store.n_steps, store.n_pixels, store.dtype) == (3, 1000, np.float32)
    again = io.RasterStore(str(tmp_path / 's'))
    assert again.n_pixels == 1000
    for _, name in io.DYNAMIC_FIELDS:
        assert again.array(name).shape == (3, 1000)
    assert again.array(io.PFT).shape == (1000,) and again.array(io.PFT).dtype == np.uint8
    data = np.arange(3000, dtype=np.float32).reshape(3, 1000)
    store.array('MERRA2/Tmin', 'r+')[:] = data
    f = io._Npy(store.path('MERRA2/Tmin'))
    buf = np.empty(100, np.float32)
    assert f.read_into(buf, 2, 250) == 400
    assert np.array_equal(buf, data[2, 250:350])
    f.close()
    w = io._Npy(store.path(io.OUT_DAY), 'w')
    w.write_from(np.full(10, 7.0, np.float32), 1, 990)
    w.close()
    out = store.array(io.OUT_DAY)
    assert (out[1, 990:] == 7.0).all() and (out[1, :990] == 0).all() and (out[0] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize('dtype,rtol', [(np.float64, 1e-8), (np.float32, None)])
def test_streamed_store_matches_the_oracle(tmp_path, dtype, rtol):
    """The tiled, overlapped pipeline on a small store (ragged last tile, fill codes)
    against the oracle's restatement of the reference's pre-processing + forward run."""
    from mod16_amd import io
    from mod16_amd.utils import restore_bplut, bplut_table
    from mod16_amd.models import COLLECTION61_BPLUT
    from parity import assert_parity
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    T, N = 3, 3 * 32768 + 4 * 301 + 1
    store = io.RasterStore.create(str(tmp_path / 's'), T, N, dtype)
    fill_store(store, seed=11)
    report = io.run_store(table, store.root, tile_pixels=32768, workers=3)
    assert report['pixels'] == N and report['steps'] == T and report['workers'] == 3
    for stage in ('read', 'h2d', 'kernel', 'd2h', 'write'):
        assert report['stages'][stage]['bytes'] > 0 and report['stages'][stage]['busy_s_sum_over_workers'] > 0
    esz = np.dtype(dtype).itemsize
    assert report['stages']['write']['bytes'] == 2 * T * N * esz
    assert report['stages']['read']['bytes'] == T * N * (11 * esz + 2) + N * (2 * esz + 1)
    bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    raw = [None] * 14
    for idx, name in io.DYNAMIC_FIELDS:
        raw[idx] = np.asarray(store.array(name), np.float64)
    raw[3] = np.zeros((T, N))
    raw[7] = np.broadcast_to(np.asarray(store.array('MERRA2/T10M_annual'), np.float64), (T, N))
    raw[13] = np.broadcast_to(np.asarray(store.array('state/elevation_m'), np.float64), (T, N))
    cls = np.broadcast_to(np.asarray(store.array(io.PFT)), (T, N))
    want = oracle.evapotranspiration_raw(bplut, cls, raw, np.asarray(store.array(io.FPAR)),
                                         np.asarray(store.array(io.LAI)))
    got = [np.asarray(store.array(io.OUT_DAY)), np.asarray(store.array(io.OUT_NIGHT))]
    for g, w, what in zip(got, want, ('day', 'night')):
        assert g.dtype == dtype
        if rtol is not None:
            assert_parity(g, w, rtol, what)
        else:       # float32 store: float64 arithmetic on the widened inputs, rounded once
            assert_parity(g, w.astype(np.float32), 1e-6, what)
    assert np.isnan(got[0][0, 1]) and np.isnan(got[1][1, 1])      # fill codes
