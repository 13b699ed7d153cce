"""tools/h5_to_store.py executed (reference mod16/calibration.py:50-112 layout,
:304-423 reader): the converter runs on a dict-backed stand-in for an open h5py.File
(tests/fake_h5py.py -- this image has no h5py) holding a 3-tower x 9-sub-pixel Cal-Val container;
the store it writes is checked array by array, then streamed through io.run_store on the GPU against
the oracle fed the container's own fields through the oracle's restatement of the reference's
pre-processing (calibration.py:380-423)."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT
from oracle import mod16_oracle as oracle

sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))

T_ALL, T0, N, P = 5, 1, 3, 9


def container(seed=3, homogeneous=False):
    """A Cal-Val container as the reference documents it: (T x N) MERRA-2 fields, (T x N x P) MODIS
    fields as FLOATS with NaN for missing values (what calibration.py:410-419 reads), (N x P) PFT."""
    rng = np.random.default_rng(seed)
    u = lambda lo, hi, shape=(T_ALL, N): rng.uniform(lo, hi, shape).astype(np.float32)
    t_day = u(265, 305)
    t_night = t_day - u(0, 12)
    d = {
        'MERRA2/LWGNT_daytime': u(-100, 0), 'MERRA2/LWGNT_nighttime': u(-50, 0),
        'MERRA2/SWGDN_daytime': u(0, 360), 'MERRA2/T10M_daytime': t_day, 'MERRA2/T10M_nighttime': t_night,
        'MERRA2/T10M': (0.5 * (t_day + t_night)).astype(np.float32), 'MERRA2/Tmin': t_night - u(0, 3),
        'MERRA2/QV10M_daytime': u(0.001, 0.02), 'MERRA2/QV10M_nighttime': u(0.001, 0.02),
        'MERRA2/PS_daytime': u(7e4, 1.0134e5), 'MERRA2/PS_nighttime': u(7e4, 1.0134e5),
        'MODIS/MCD43GF_black_sky_sw_albedo': u(0.1, 0.22, (T_ALL, N, P)),
        'MODIS/MOD15A2HGF_fPAR': np.rint(u(2, 89, (T_ALL, N, P))),
        'MODIS/MOD15A2HGF_LAI': np.rint(u(1, 53, (T_ALL, N, P))),
        'state/PFT': np.array(oracle.PFT_VALID, np.int16)[rng.integers(0, 11, (N, P))],
        'state/elevation_m': u(-50, 3500, (N,)),
        'FLUXNET/latent_heat': u(0, 200),
    }
    if homogeneous:
        for k in ('MODIS/MCD43GF_black_sky_sw_albedo', 'MODIS/MOD15A2HGF_fPAR', 'MODIS/MOD15A2HGF_LAI'):
            d[k][:] = d[k][:, :, :1]
        d['state/PFT'][:] = d['state/PFT'][:, :1]
    else:
        d['MODIS/MOD15A2HGF_fPAR'][2, 1, 4] = np.nan           # missing values, as the reference sees them
        d['MODIS/MOD15A2HGF_LAI'][3, 0, 0] = np.nan
        d['MODIS/MOD15A2HGF_LAI'][T0, 2, 8] = 0.0
    return d


def test_modis_codes_keep_missing_values_missing():
    """fPAR / LAI arrive as floats with NaN for missing values; a bare astype(uint8)
    turned NaN into 0 -- a VALID fPAR. Non-finite and out-of-range values become the fill code the
    kernel decodes to NaN; stored uint8 codes pass through."""
    import h5_to_store
    a = np.array([0.0, 1.4, 1.5, 2.5, 100.0, 248.4, 248.6, 249.0, 255.0, -0.4, -0.6, -3.0, np.nan, np.inf, -np.inf, 1e9])
    got = h5_to_store.modis_code(a)
    assert got.dtype == np.uint8
    assert got.tolist() == [0, 1, 2, 2, 100, 248, 255, 255, 255, 0, 255, 255, 255, 255, 255, 255]
    assert h5_to_store.modis_code(a.astype(np.float32)).tolist() == got.tolist()
    codes = np.arange(256, dtype=np.uint8)
    assert h5_to_store.modis_code(codes) is codes                  # MOD15 codes as stored: untouched (>= 249 stay fill)
    assert h5_to_store.modis_code(np.array([3, 250, -1], np.int16)).tolist() == [3, 255, 255]


@pytest.mark.parametrize('subgrid', ['flatten', 'center'])
def test_converter_writes_the_documented_store(tmp_path, subgrid):
    import fake_h5py
    import h5_to_store
    from mod16_amd import io
    src = container()
    with fake_h5py.File(src) as hdf:
        store = h5_to_store.convert(hdf, str(tmp_path / 's'), t0=T0, subgrid=subgrid)
        # step by step (the container may not fit host memory), and nothing it does not need
        assert hdf['FLUXNET/latent_heat'].reads == 0 and hdf['MERRA2']['LWGNT_daytime'].reads == T_ALL - T0
    T = T_ALL - T0
    n_pix = N * P if subgrid == 'flatten' else N
    assert (store.n_steps, store.n_pixels, store.dtype) == (T, n_pix, np.float32)
    again = io.RasterStore(str(tmp_path / 's'))
    assert (again.n_steps, again.n_pixels) == (T, n_pix)

    def tower(a):       # (T, N) or (N,) -> the store's pixel axis
        return np.repeat(a, P, axis=-1) if subgrid == 'flatten' else a

    def sub(a):         # (T, N, P) or (N, P)
        return a.reshape(a.shape[:-2] + (N * P,)) if subgrid == 'flatten' else a[..., P // 2]
    for _, name in io.DYNAMIC_FIELDS:
        want = sub(src[name][T0:]) if src[name].ndim == 3 else tower(src[name][T0:])
        assert np.array_equal(np.asarray(again.array(name)), want.astype(np.float32)), name
    # annual mean temperature: the mean over the converted steps of the 24-h mean (calibration.py:390)
    mat = src['MERRA2/T10M'][T0:].astype(np.float64).mean(axis=0).astype(np.float32)
    assert np.array_equal(np.asarray(again.array('MERRA2/T10M_annual')), tower(mat))
    assert np.array_equal(np.asarray(again.array('state/elevation_m')), tower(src['state/elevation_m']))
    assert np.array_equal(np.asarray(again.array(io.PFT)), sub(src['state/PFT']).astype(np.uint8))
    for name, key in ((io.FPAR, 'MODIS/MOD15A2HGF_fPAR'), (io.LAI, 'MODIS/MOD15A2HGF_LAI')):
        got = np.asarray(again.array(name))
        want = sub(src[key][T0:])
        assert got.dtype == np.uint8
        assert np.array_equal(got[~np.isnan(want)], want[~np.isnan(want)].astype(np.uint8))
        assert (got[np.isnan(want)] == 255).all()
    if subgrid == 'flatten':
        fp = np.asarray(again.array(io.FPAR))
        assert fp[2 - T0, 1 * P + 4] == 255 and np.asarray(again.array(io.LAI))[3 - T0, 0] == 255
    with pytest.raises(ValueError):
        h5_to_store.convert(fake_h5py.File(src), str(tmp_path / 'x'), t0=T_ALL)
    with pytest.raises(KeyError):
        h5_to_store.convert(fake_h5py.File({k: v for k, v in src.items() if 'Tmin' not in k}), str(tmp_path / 'y'))
    # another albedo dataset (a starred, i.e. configurable, name of calibration.py:50-112)
    src2 = dict(src, **{'MODIS/MCD43GF_white_sky_sw_albedo': src['MODIS/MCD43GF_black_sky_sw_albedo'] + 0.01})
    st2 = h5_to_store.convert(fake_h5py.File(src2), str(tmp_path / 'z'), t0=T0, subgrid=subgrid,
                              names={'albedo': 'MODIS/MCD43GF_white_sky_sw_albedo'})
    assert np.allclose(np.asarray(st2.array('MODIS/MCD43GF_black_sky_sw_albedo')),
                       np.asarray(again.array('MODIS/MCD43GF_black_sky_sw_albedo')) + 0.01, atol=1e-6)


def reference_inputs(src, reduce_subgrid):
    """The container's fields through the reference's pre-processing as the oracle restates it
    (oracle.evapotranspiration_raw = calibration.py:380-423 + the forward run), per sub-pixel
    (`reduce_subgrid` None) or on the np.nanmean over the sub-grid as calibration.py:412-419 does."""
    T = T_ALL - T0
    f64 = lambda k: src[k][T0:].astype(np.float64)
    if reduce_subgrid is None:
        rep = lambda a: np.repeat(a, P, axis=-1)
        flat = lambda a: a.reshape(a.shape[:-2] + (N * P,))
        fpar, lai, alb = (flat(f64(k)) for k in ('MODIS/MOD15A2HGF_fPAR', 'MODIS/MOD15A2HGF_LAI', 'MODIS/MCD43GF_black_sky_sw_albedo'))
        cls = np.broadcast_to(flat(src['state/PFT']).astype(np.uint8), (T, N * P))
    else:
        rep = lambda a: a
        with np.errstate(all='ignore'):
            fpar, lai, alb = (np.nanmean(f64(k), axis=-1) for k in ('MODIS/MOD15A2HGF_fPAR', 'MODIS/MOD15A2HGF_LAI', 'MODIS/MCD43GF_black_sky_sw_albedo'))
        cls = np.broadcast_to(src['state/PFT'][:, P // 2].astype(np.uint8), (T, N))
    mat = src['MERRA2/T10M'][T0:].astype(np.float64).mean(axis=0).astype(np.float32).astype(np.float64)
    raw = [rep(f64('MERRA2/LWGNT_daytime')), rep(f64('MERRA2/LWGNT_nighttime')), rep(f64('MERRA2/SWGDN_daytime')),
           np.zeros_like(rep(f64('MERRA2/SWGDN_daytime'))), alb, rep(f64('MERRA2/T10M_daytime')),
           rep(f64('MERRA2/T10M_nighttime')), np.broadcast_to(rep(mat), rep(f64('MERRA2/Tmin')).shape),
           rep(f64('MERRA2/Tmin')), rep(f64('MERRA2/QV10M_daytime')), rep(f64('MERRA2/QV10M_nighttime')),
           rep(f64('MERRA2/PS_daytime')), rep(f64('MERRA2/PS_nighttime')),
           np.broadcast_to(rep(src['state/elevation_m'].astype(np.float64)), rep(f64('MERRA2/Tmin')).shape)]
    return cls, raw, fpar, lai


@pytest.mark.gpu
def test_converted_container_through_run_store_matches_the_oracle(tmp_path):
    """flatten: every sub-pixel of every tower through io.run_store == the oracle on the container's
    own fields (the store never enters the expected values); then what the reference does instead --
    one forward run per tower on the sub-grid MEANS (calibration.py:412-419) -- against the mean of
    our per-sub-pixel results: equal for a homogeneous sub-grid, different otherwise (the forward run
    is not linear in fPAR / LAI / albedo)."""
    import fake_h5py
    import h5_to_store
    from mod16_amd import io
    from mod16_amd.utils import restore_bplut, bplut_table
    from mod16_amd.models import COLLECTION61_BPLUT
    from parity import assert_parity
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    T = T_ALL - T0
    gaps = {}
    for tag, homogeneous in (('mixed', False), ('homogeneous', True)):
        src = container(homogeneous=homogeneous)
        store = h5_to_store.convert(fake_h5py.File(src), str(tmp_path / tag), t0=T0, subgrid='flatten')
        report = io.run_store(table, store.root, tile_pixels=16, workers=2)
        assert report['pixels'] == N * P and report['steps'] == T
        got = [np.asarray(store.array(io.OUT_DAY)), np.asarray(store.array(io.OUT_NIGHT))]
        cls, raw, fpar, lai = reference_inputs(src, None)
        with np.errstate(all='ignore'):
            # MODIS codes: the oracle decodes the uint8 fields (NaN where the container had NaN)
            want = oracle.evapotranspiration_raw(bplut, cls, raw, h5_to_store.modis_code(fpar), h5_to_store.modis_code(lai))
        for g, w, what in zip(got, want, ('day', 'night')):
            assert_parity(g, w.astype(np.float32), 1e-6, '%s %s' % (tag, what))
        if not homogeneous:
            assert np.isnan(got[0][2 - T0, 1 * P + 4]) and np.isnan(got[1][3 - T0, 0])      # the missing fPAR / LAI
        # the reference's tower-level run on sub-grid means (float fPAR / LAI: the class surface, not the store)
        cls_t, raw_t, fpar_t, lai_t = reference_inputs(src, 'mean')
        with np.errstate(all='ignore'):
            params = oracle.gather_params(bplut, cls_t)
            vpd_d = oracle.vpd_from_humidity(raw_t[9], raw_t[11], raw_t[5])
            vpd_n = np.maximum(oracle.vpd_from_humidity(raw_t[10], raw_t[12], raw_t[6]), 0)
            tower = oracle.evapotranspiration(params, *raw_t[:9], vpd_d, vpd_n, oracle.air_pressure(raw_t[13]),
                                              fpar_t / 100, lai_t / 10)
            ours = [np.nanmean(g.astype(np.float64).reshape(T, N, P), axis=-1) for g in got]
        with np.errstate(all='ignore'):
            gaps[tag] = max(float(np.nanmax(np.where(t != 0, np.abs(o - t) / np.abs(t), 0.0))) for o, t in zip(ours, tower))
    assert gaps['homogeneous'] < 1e-5, gaps          # same inputs in every sub-pixel: the two orders agree
    assert gaps['mixed'] > 1e-3, gaps                # heterogeneous sub-grid: mean of ETs != ET of means


def test_field_map_covers_what_the_reference_reads():
    """tools/h5_to_store.py's field map against the layout AS THE REFERENCE STATES IT
    (tests/golden/calval_layout.json: the file specification of calibration.py:50-112 and every
    dataset _load_data, :304-423, opens -- tests/golden/make_calval_layout.py reads both out of the
    reference's text): every look-up key the reference's forward-run inputs come from is a key of
    the converter (same key, same day / night indices), every default path is a dataset of the
    documented layout with the rank the converter expects, nothing is read that the reference does
    not read, and the keys left out are the ones that are no forward-run input."""
    import json
    import h5_to_store
    from conftest import GOLDEN
    from mod16_amd import io
    lay = json.load(open(os.path.join(GOLDEN, 'calval_layout.json')))
    documented = {e['path']: e for e in lay['documented']}
    ref_keys = lay['load_data']['lookup_keys']
    # look-up keys: the reference's, plus class_map (a configuration key there, calibration.py:336)
    mine = dict(h5_to_store.SOURCES)
    assert 'class_map' in lay['load_data']['config_keys'] and 'class_map' in mine
    assert set(mine) - {'class_map'} == set(ref_keys) - {'VPD'}, (sorted(mine), sorted(ref_keys))
    for key, indices in ref_keys.items():
        if key == 'VPD':
            continue                                    # optional in the reference too (:391-393): --subgrid mean reads it if named
        v = mine[key]
        if indices:                                     # lookup[KEY][i]: a [daytime, nighttime] pair, same indices read
            assert isinstance(v, list) and len(v) == 2 and all(v[i] for i in indices), key
        else:
            assert isinstance(v, str), key
    # every default path is documented, with the rank the converter handles
    rank = {'class_map': ['N', 'P'], 'elevation': ['N'], 'albedo': ['T', 'N', 'P'], 'fPAR': ['T', 'N', 'P'], 'LAI': ['T', 'N', 'P']}
    for key, v in mine.items():
        for path in (v if isinstance(v, list) else [v]):
            if path is None:
                continue
            assert path in documented, path
            assert documented[path]['dims'] == rank.get(key, ['T', 'N']), (path, documented[path]['dims'])
    # names the reference lets its configuration change are the ones --name changes here (all keys);
    # the MODIS and land-cover names carry the star in the documented layout
    for key in ('albedo', 'fPAR', 'LAI', 'class_map'):
        assert documented[mine[key]]['starred'], key
    # the store's files: one per dynamic driver of mod16_amd.io, each fed by a look-up key
    assert set(h5_to_store.STORE_FROM) == {name for _, name in io.DYNAMIC_FIELDS}
    for name, (key, index) in h5_to_store.STORE_FROM.items():
        assert h5_to_store.source(mine, key, index) == name      # default names: the store mirrors the container
    # the shipped configuration's keys: the forward-run ones are all here; annual_precip is a constraint
    # of the calibration (:constraints), not a driver
    shipped = set(lay['shipped_config_datasets'])
    assert shipped - set(mine) - {'VPD'} == {'annual_precip'}
    # what _load_data opens besides the drivers is calibration bookkeeping the forward run does not need
    assert set(lay['load_data']['literal_paths']) == {'FLUXNET/site_id', 'FLUXNET/validation_mask', 'time', 'weights'}


@pytest.mark.gpu
def test_tower_protocol_subgrid_mean_matches_the_oracle(tmp_path):
    """--subgrid mean: the reference's own tower protocol (calibration.py:336-340, :380-423) -- fPAR,
    LAI and albedo averaged over the sub-grid BEFORE one forward run per tower-day on the sub-grid's
    dominant PFT. The converter writes the processed drivers (VPD and air pressure through the
    library's methods), run_processed streams them through the forward run; against the oracle fed
    the container's own fields through the oracle's restatement of that pre-processing."""
    import fake_h5py
    import h5_to_store
    from mod16_amd import io
    from mod16_amd.utils import restore_bplut, bplut_table, pft_dominant
    from mod16_amd.models import COLLECTION61_BPLUT
    from parity import assert_parity
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    T = T_ALL - T0
    src = container()
    out_dir = h5_to_store.convert(fake_h5py.File(src), str(tmp_path / 'm'), t0=T0, subgrid='mean')
    # the means are np.nanmean over the sub-grid, as floats (no MODIS codes), scaled as :422-423
    with np.errstate(all='ignore'):
        want_fpar = np.nanmean(src['MODIS/MOD15A2HGF_fPAR'][T0:].astype(np.float64), axis=-1) / 100
    got_fpar = np.load(os.path.join(out_dir, 'fpar.npy'))
    assert got_fpar.shape == (T, N) and got_fpar.dtype == np.float32
    assert np.allclose(got_fpar, want_fpar, rtol=1e-6, equal_nan=True)
    assert np.array_equal(np.load(os.path.join(out_dir, 'class.npy'))[0], pft_dominant(src['state/PFT']).astype(np.uint8))
    assert not np.load(os.path.join(out_dir, 'sw_rad_night.npy')).any()
    day, night = h5_to_store.run_processed(table, out_dir)
    # the oracle on the same protocol
    cls_t, raw_t, fpar_t, lai_t = reference_inputs(src, 'mean')
    cls_dom = np.broadcast_to(pft_dominant(src['state/PFT']).astype(np.uint8), (T, N))
    with np.errstate(all='ignore'):
        params = oracle.gather_params(bplut, cls_dom)
        vpd_d = oracle.vpd_from_humidity(raw_t[9], raw_t[11], raw_t[5])
        vpd_n = np.maximum(oracle.vpd_from_humidity(raw_t[10], raw_t[12], raw_t[6]), 0)
        want = oracle.evapotranspiration(params, *raw_t[:9], vpd_d, vpd_n, oracle.air_pressure(raw_t[13]),
                                         fpar_t / 100, lai_t / 10)
    for g, w, what in zip((day, night), want, ('day', 'night')):
        assert g.shape == (T, N) and g.dtype == np.float32
        # float32 drivers (VPD is a difference of like numbers in humid air): 2e-4
        assert_parity(np.asarray(g), np.asarray(w, np.float64).astype(np.float32), 2e-4, what)
