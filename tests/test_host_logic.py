"""Host-side logic of the drop-in surface (no GPU): BPLUT parsing against the
reference's own parse (golden), model construction, marshalling rules."""
import io
import os
import sys

import numpy as np
import pytest

import mod16_amd
from mod16_amd import models, utils


def test_restore_bplut_matches_reference_parse(golden):
    tables = golden('bplut_tables')
    assert len(tables.files) == 5
    for fn in tables.files:
        got = utils.restore_bplut(os.path.join(utils.DATA_DIR, fn))
        assert list(got) == list(utils.BPLUT_FIELD_LOOKUP.values())
        table = np.stack([got[k] for k in mod16_amd.MOD16.required_parameters], 1)
        assert np.array_equal(table, tables[fn], equal_nan=True), fn
        assert np.isnan(got['beta']).all()              # no beta row in any file
        assert np.isnan(got['gl_sh'][[0, 11]]).all()    # codes that are not PFTs
    # buffers work as paths do
    path = os.path.join(utils.DATA_DIR, tables.files[0])
    a = utils.restore_bplut(io.StringIO(open(path).read()))
    assert np.array_equal(a['csl'], utils.restore_bplut(path)['csl'], equal_nan=True)


def test_write_bplut_round_trip(tmp_path):
    src = utils.restore_bplut(models.COLLECTION61_BPLUT)
    src['beta'] = np.where(np.isnan(src['csl']), np.nan, 250.0)
    out = tmp_path / 'bplut.csv'
    utils.write_bplut(src, str(out))
    back = utils.restore_bplut(str(out))
    for k in src:
        assert np.array_equal(src[k], back[k], equal_nan=True), k


def test_collection61_parameters(golden):
    want = golden('collection61_params')['table']
    for pft in mod16_amd.PFT_VALID:
        m = models.MOD16Collection61(pft)
        got = [getattr(m, k) for k in mod16_amd.MOD16.required_parameters]
        assert np.array_equal(got, want[pft]), pft
        assert m.beta == 250 and m.params['beta'] == 250
    with pytest.raises(AssertionError):
        models.MOD16Collection61(11)
    assert models.PFT_ALL['Croplands (CRO)'] == 12 and len(models.PFT_ALL) == 11


def test_constructor_contract():
    p = dict.fromkeys(mod16_amd.MOD16.required_parameters, 2.0)
    m = mod16_amd.MOD16(p)
    assert m.params is p and m.csl == 2.0
    del p['beta']
    with pytest.raises(KeyError):
        mod16_amd.MOD16(p)
    assert mod16_amd.MOD16.required_parameters == [
        'tmin_close', 'tmin_open', 'vpd_open', 'vpd_close', 'gl_sh', 'gl_wv',
        'g_cuticular', 'csl', 'rbl_min', 'rbl_max', 'beta']
    assert mod16_amd.PFT_VALID == (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12)
    assert mod16_amd.SPECIFIC_HEAT_CAPACITY_AIR == 1013
    assert mod16_amd.STEFAN_BOLTZMANN == 5.67e-8


def test_bplut_table_layout():
    src = utils.restore_bplut(models.COLLECTION61_BPLUT)
    t = utils.bplut_table(src, beta=250)
    assert t.shape == (13, 11) and t.flags.c_contiguous
    assert t[7, 3] == 4400 and t[12, 7] == 0.0055          # vpd_close OSH, csl CRO
    assert np.isnan(t[0]).all() and np.isnan(t[11]).all()  # beta only where a PFT exists
    assert (t[[1, 5, 12], 10] == 250).all()
    assert np.isnan(utils.bplut_table(src)[:, 10]).all()


def test_pft_dominant():
    pft_map = np.array([[1, 1, 2, 0], [11, 0, 0, 0], [4, 5, 5, 5], [7, 7, 7, 7]])
    got = utils.pft_dominant(pft_map)
    assert got.dtype == np.float32 and list(got) == [1, 0, 5, 7]
    got = utils.pft_dominant(pft_map, site_list=['a', 'US-A10', 'CA-SF2', 'US-NGC'])
    assert list(got) == [1, 0, 3, 3]


def test_result_dtype_follows_numpy_rules():
    f32, f64 = np.float32, np.float64
    rd = mod16_amd._result_dtype
    assert rd([np.ones(3, f32), 1.0, 2]) == f32           # Python scalars are weak
    assert rd([np.ones(3, f32), np.float64(1.0)]) == f64  # numpy float64 scalar is not
    assert rd([np.ones(3, f32), np.ones(3, f64)]) == f64
    assert rd([np.ones(3, np.int64), np.ones(3, f32)]) == f64
    assert rd([1.0, 2.0]) == f64


def test_cheap_marshalling_helpers_agree_with_numpy():
    """The call on a flux-tower site's scalars spends its time in the marshalling, so the dtype
    rule, the shapes and the broadcast are computed without numpy's general machinery where the
    answer is obvious -- and must be numpy's answer everywhere: every combination of up to three
    of these inputs."""
    import itertools
    cands = [1.0, 2, True, np.float32(1), np.float64(1), np.ones(3, np.float32), np.ones(3),
             np.ones(3, np.uint8), np.ones(3, np.int16), np.ones(3, np.float16), np.int64(3),
             [1.0, 2.0, 3.0], np.ones((2, 3), np.float32), np.ones((2, 1)), np.ones(()), np.ma.masked_array([1.0, 2.0, 3.0])]

    def numpy_rule(values):
        strong = [np.asarray(v).dtype for v in values if isinstance(v, (np.ndarray, np.generic))]
        if strong and np.result_type(*strong) == np.float32:
            return np.dtype(np.float32)
        return np.dtype(np.float64)
    for r in (1, 2, 3):
        for combo in itertools.product(cands, repeat=r):
            assert mod16_amd._result_dtype(combo) == numpy_rule(combo), [type(c) for c in combo]
            shapes = [mod16_amd._shape(v) for v in combo]
            assert shapes == [np.shape(v) for v in combo]
            shape, n = mod16_amd._broadcast(shapes)
            assert shape == np.broadcast_shapes(*shapes) and n == int(np.prod(shape, dtype=np.int64))
    assert mod16_amd._broadcast([(0, 3), (1, 3)]) == ((0, 3), 0)
    with pytest.raises(ValueError):
        mod16_amd._broadcast([(2,), (3,)])
    # addresses: the buffer protocol where it applies, the array interface otherwise -- the same number
    for a in (np.ones(5), np.ones((2, 3), np.float32), np.ones(0), np.broadcast_to(np.ones(3), (2, 3)).copy()):
        assert mod16_amd._address(a) == a.ctypes.data
    ro = np.ones(4)
    ro.flags.writeable = False
    assert mod16_amd._address(ro) == ro.ctypes.data


def test_marshal_strides_and_broadcast():
    shape = (4, 5)
    vals = [3.0, np.arange(5.0), np.ones(shape), np.float32(2)]
    keep, ptrs, strides = mod16_amd._marshal(vals, shape, np.float64)
    assert strides == [0, 1, 1, 0]
    # the size-1 inputs share one small array (slot i of it for value i), the others follow
    scal, row, ones = keep
    assert scal[0] == 3.0 and scal[3] == 2.0
    assert ptrs[0] == scal.ctypes.data and ptrs[3] == scal.ctypes.data + 3 * 8
    assert row.shape == shape and row.flags.c_contiguous
    assert np.array_equal(row[2], np.arange(5.0))
    assert all(k.dtype == np.float64 for k in keep)
    assert ptrs[1] == row.ctypes.data and ptrs[2] == ones.ctypes.data
    # float32 results: the scalars are stored as float32
    keep, ptrs, strides = mod16_amd._marshal([0.1, np.ones(3, np.float32)], (3,), np.float32)
    assert keep[0].dtype == np.float32 and keep[0][0] == np.float32(0.1) and strides == [0, 1]
    # a read-only dense input (a broadcast view made contiguous, a memory map opened 'r')
    ro = np.ones(shape)
    ro.flags.writeable = False
    keep, ptrs, strides = mod16_amd._marshal([ro], shape, np.float64)
    assert ptrs[0] == ro.ctypes.data and keep[0] is ro


def test_result_arrays_fall_back_to_numpy_without_a_gpu():
    """The pool of page-locked result blocks (mod16_amd/_lib.py) hands out plain
    numpy arrays when no page-locked memory can be had (no GPU here) or the array
    is small."""
    import numpy as np
    from mod16_amd import _lib
    a = _lib.pinned.empty((1200, 1200), np.float64)
    assert a.shape == (1200, 1200) and a.dtype == np.float64
    assert a.flags.writeable and a.flags.c_contiguous
    a[:] = 2.0
    assert a.sum() == 2.0 * 1200 * 1200
    small = _lib.pinned.empty((10,), np.float32)
    assert small.base is None and small.dtype == np.float32


def test_broadcast_kinds_are_not_made_dense():
    """(N,) rows and (T, 1) columns against (T, N) drivers (reference
    mod16/__init__.py:180-181, notebook cell 17) go to the C ABI as they are:
    the marshalling hands over N (or T) elements and a broadcast kind, no (T, N)
    temporary."""
    import numpy as np
    import mod16_amd
    from mod16_amd import _lib
    T, N = 7, 50
    kind = mod16_amd._broadcast_kind
    assert kind((), 1, (T, N)) == _lib.BC_SCALAR
    assert kind((1, 1), 1, (T, N)) == _lib.BC_SCALAR
    assert kind((T, N), T * N, (T, N)) == _lib.BC_DENSE
    assert kind((N,), N, (T, N)) == _lib.BC_ROW
    assert kind((1, N), N, (T, N)) == _lib.BC_ROW
    assert kind((T, 1), T, (T, N)) == _lib.BC_COL
    assert kind((N,), N, (3, T, N)) == _lib.BC_ROW
    assert kind((3, T, 1), 3 * T, (3, T, N)) == _lib.BC_COL
    assert kind((T, 1), T, (3, T, N)) is None            # another pattern: made dense
    assert kind((N,), N, (N,)) == _lib.BC_DENSE
    values = [np.ones((T, N)), np.arange(N, dtype=float), np.arange(T, dtype=float)[:, None], 3.0,
              np.ones((1, N), np.float32)]
    keep, ptrs, kinds = mod16_amd._marshal(values, (T, N), np.float64, kinds=True)
    assert kinds == [_lib.BC_DENSE, _lib.BC_ROW, _lib.BC_COL, _lib.BC_SCALAR, _lib.BC_ROW]
    assert [a.size for a in keep] == [T * N, N, T, len(values), N]      # (the size-1 inputs share one array)
    assert all(a.flags.c_contiguous and a.dtype == np.float64 for a in keep)
    # without the switch everything but scalars is dense, as before
    keep, ptrs, strides = mod16_amd._marshal(values, (T, N), np.float64)
    assert strides == [1, 1, 1, 0, 1] and [a.size for a in keep] == [T * N, T * N, T * N, len(values), T * N]


def test_pmc_traffic_belongs_to_the_build(tmp_path):
    """bench.py reports roofline.traffic only from a PMC record measured on the very build of the
    library that is loaded (mod16_build_id): a record of another build -- a kernel change without
    fresh tools/run_profiles.sh passes -- gives None and says why."""
    import json
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    from mod16_amd import _lib
    mine = _lib.build_id()
    assert len(mine) == 16 and mine != 'unknown'
    assert _lib.load_experiments().mod16_build_id().decode() != mine      # another set of flags
    rec = {'pixels_per_launch': 1000, 'dtype': 'float64', 'layout': 'tiled',
           'traffic_bytes_per_launch': 129050.0, 'build_id': 'feedfacefeedface', 'git_commit': 'abc'}
    (tmp_path / 'r09_pmc_hbm_traffic.json').write_text(json.dumps(rec))
    got = bench.pmc_traffic(1000, 'float64', 'tiled', mine, profiles_dir=str(tmp_path))
    assert got[0] is None and got[1] is None and 'feedfacefeedface' in got[2] and mine in got[2]
    rec['build_id'] = mine
    (tmp_path / 'r10_pmc_hbm_traffic.json').write_text(json.dumps(rec))
    got = bench.pmc_traffic(1000, 'float64', 'tiled', mine, profiles_dir=str(tmp_path))
    assert got[0] == 129050.0 and got[1].endswith('r10_pmc_hbm_traffic.json')
    # another shape: nothing applies
    assert bench.pmc_traffic(2000, 'float64', 'tiled', mine, profiles_dir=str(tmp_path))[0] is None
    assert bench.pmc_traffic(1000, 'float32', 'tiled', mine, profiles_dir=str(tmp_path))[0] is None
    # the committed records (older rounds carry no build id): never reported for this build unless measured on it
    t, src, note = bench.pmc_traffic(933120000, 'float64', 'tiled', mine)
    assert (t is None) == (src is None) and note


def test_summary_is_the_tail_of_the_line():
    """The driver's record keeps the last 8 KB of stdout and only the scalars of `roofline`: the
    line ends in a flat `summary` object of scalars (every configuration's headline number, the
    clock, the power and cycles_per_step = kernel_ms x shader clock) that the last 6000 bytes of
    a FULL line still carry, and `roofline` repeats plain_frac / sclk_mhz / power_w /
    cycles_per_step as scalars. Input: last round's complete line (profiles/r04b_bench_line.json),
    put through this round's build_summary()."""
    import json
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    line = json.load(open(os.path.join(ROOT, 'profiles', 'r04b_bench_line.json')))
    line.pop('summary', None)
    line['numpy_in_numpy_out'] = {'pixels': 33554432, 'n_devices': 2, 'pixels_per_s': 1.0e9,
                                  'one_device_pixels_per_s': 0.5e9, 'bits_equal_one_device': True}
    line['roofline'].update(bench.roofline_scalars(line))
    line['summary'] = bench.build_summary(line)
    text = json.dumps(line)
    assert len(text) > 12000                              # a full line: well beyond what the record keeps
    tail = text[-6000:]
    at = tail.index('"summary": ')
    summary = json.loads(tail[at + len('"summary": '):-1])
    assert text.endswith(json.dumps(summary) + '}')        # the LAST key
    for v in summary.values():
        assert v is None or isinstance(v, (bool, int, float, str)), v      # scalars only
    want = ('c2_us', 'c2_us_no_diag', 'c4_ms_per_step', 'c4_gpx_s', 'c5_mixed_ms', 'c5_fast_ms', 'plain_ms',
            'plain_frac', 'raw_f64_frac', 'raw_mixed_frac', 'ingest_gpx_s', 'n2_resident_frac', 'parity_max_rel',
            'parity_masks_equal', 'sclk_mhz', 'power_w', 'cycles_per_step', 'kernel_ms', 'frac', 'host_call_gpx_s')
    for k in want:
        assert summary.get(k) is not None, k
    roof = line['roofline']
    assert summary['cycles_per_step'] == roof['kernel_ms'] * roof['device_under_load']['sclk_mhz'] * 1e3
    assert 30e6 < summary['cycles_per_step'] < 40e6
    for k in ('plain_frac', 'sclk_mhz', 'power_w', 'cycles_per_step'):
        assert isinstance(roof[k], float) and roof[k] == summary[k]
    assert summary['c2_us'] == line['configs']['c2_1200x1200_float64']['tile_us_per_launch']
    assert summary['host_call_gpx_s'] == 1.0 and summary['host_call_devices'] == 2
    # a line without the optional legs (N > 1, --no-configs ...) still has a summary of nulls
    bare = {'value': 1e9, 'roofline': {'kernel_ms': 2.5, 'frac': 0.7}, 'configs': None, 'parity': None}
    bare['roofline'].update(bench.roofline_scalars(bare))
    s = bench.build_summary(bare)
    assert s['c2_us'] is None and s['cycles_per_step'] is None and s['kernel_ms'] == 2.5


def test_pinned_pool_bound_per_rank_and_deferred_frees(monkeypatch):
    """(i) the default bound on page-locked result memory is per PROCESS -- with N ranks
    on the node (LOCAL_WORLD_SIZE / WORLD_SIZE) each takes 1/N of it, not N times a quarter of the RAM;
    (ii) a give() that runs from a finalizer while its own thread is inside take() / trim() must not
    call hipHostFree (which synchronises the device) under the lock: the block is deferred to the
    lock's holder, who frees it after releasing."""
    from mod16_amd import _lib
    monkeypatch.delenv('LOCAL_WORLD_SIZE', raising=False)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    whole = _lib._default_pinned_live()
    assert (8 << 30) <= whole <= (64 << 30)
    monkeypatch.setenv('WORLD_SIZE', '4')
    assert _lib._default_pinned_live() == max(1 << 30, whole // 4)
    monkeypatch.setenv('LOCAL_WORLD_SIZE', '8')          # the ranks of THIS node count
    assert _lib._default_pinned_live() == max(1 << 30, whole // 8)
    monkeypatch.setenv('LOCAL_WORLD_SIZE', 'x')
    assert _lib._default_pinned_live() == whole
    pool = _lib._PinnedPool()
    pool.MAX_LIVE, pool.MAX_CACHED = 4 << 20, 0
    freed = []

    def free_blocks(doomed):
        assert getattr(pool._inside, 'n', 0) == 0        # never under the lock
        freed.extend(doomed)
    monkeypatch.setattr(pool, '_free_blocks', free_blocks)
    pool.live = 3 << 20
    pool.give(111, 1 << 20)                              # nothing recent, MAX_CACHED = 0: freed at once, lock released
    assert freed == [111] and pool.live == 2 << 20
    # the re-entrant case: a finalizer fires while this thread is inside the lock
    doomed = []
    pool._enter()
    pool.give(222, 1 << 20)
    assert freed == [111] and pool.deferred == [222]     # not freed under the lock
    pool._leave(doomed)
    assert doomed == [222] and pool.deferred == [] and pool.live == 1 << 20
    # ... and trim() / take() hand the deferred blocks to _free_blocks on their way out
    pool.deferred.append(333)
    pool.trim()
    assert freed == [111, 333]


def test_build_is_judged_by_the_digest_in_the_library(tmp_path):
    """mod16_amd/csrc/build.py rebuilds a library whose embedded build id is not the digest of the
    sources beside it -- whatever the files' dates say (round 5 compared dates: a snapshot that
    already held a library never met the compiler): the in-tree library is current, a copy of it
    with another id is not, and neither is a file without the marker."""
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, 'mod16_amd', 'csrc'))
    import build
    assert build.built_id(build.OUT) == build.build_id() and build.up_to_date(build.OUT)
    assert build.built_id(build.OUT_EXP) == build.build_id(['-DMOD16_EXPERIMENTS'])
    assert build.up_to_date(build.OUT_EXP, ['-DMOD16_EXPERIMENTS']) and not build.up_to_date(build.OUT_EXP)
    blob = open(build.OUT, 'rb').read()
    mine = build.build_id().encode()
    other = tmp_path / 'other.so'
    other.write_bytes(blob.replace(b'mod16-build-id=' + mine, b'mod16-build-id=' + b'0123456789abcdef'))
    os.utime(other, None)                                  # newer than every source
    assert build.built_id(str(other)) == '0123456789abcdef' and not build.up_to_date(str(other))
    assert build.built_id(str(tmp_path / 'absent.so')) is None
