'''
Rasters on disk.

Two entry points:

``evapotranspiration_npy``
    ``evapotranspiration_raster`` on ``.npy`` files opened as memory maps (the
    processed drivers of ``MOD16.evapotranspiration``), through the library's
    HOST mode.

``RasterStore`` / ``run_store``
    A raster time series on disk in the field set of the reference's Cal-Val
    dataset (mod16/calibration.py:50-112) -- the RAW reanalysis and MODIS
    fields, one file per dataset under the same group / dataset names -- run
    through a tiled, overlapped pipeline::

        file -> page-locked buffer -> H2D -> fused raw-driver kernel -> D2H -> file

    several workers deep, so that disk reads, PCIe in both directions, the kernel
    and disk writes of different tiles proceed at the same time; the report gives
    every stage's rate and the sustained pixels/s. The pre-processing the
    reference does after reading those fields (calibration.py:380-423:
    ``MOD16.vpd``, the night-time VPD clamp, ``MOD16.air_pressure``, fPAR / 100,
    LAI / 10) is the kernel's (``mod16_et_raw_*``), so the raw bytes are what
    crosses PCIe: 46 B/pixel in float32 against 57 for the processed drivers.

HDF5 itself cannot be served in this image (no h5py); the store is the same
datasets as ``.npy`` files in directories named like the HDF5 groups.
'''
import os
import threading
import time

import numpy as np

from . import _lib, evapotranspiration_raster

DRIVER_NAMES = (
    'lw_net_day', 'lw_net_night', 'sw_rad_day', 'sw_rad_night', 'sw_albedo',
    'temp_day', 'temp_night', 'temp_annual', 'tmin', 'vpd_day', 'vpd_night',
    'pressure', 'fpar', 'lai')


def evapotranspiration_npy(bplut, cls_path, driver_paths, out_day_path, out_night_path,
                           beta=None, math=_lib.MATH_FAST, device=0, devices=None):
    '''
    ``evapotranspiration_raster`` on ``.npy`` files.

    Parameters
    ----------
    bplut : dict or numpy.ndarray
        As for ``evapotranspiration_raster``
    cls_path : str
        ``.npy`` file of the land-cover class raster (uint8)
    driver_paths : dict or sequence
        The 14 driver files, by name (``DRIVER_NAMES``) or in that order; all
        of the class raster's shape, all float64 or all float32
    out_day_path, out_night_path : str
        ``.npy`` files to create (same shape and dtype as the drivers)
    devices : sequence of int
        Several GPUs behind the call, as for ``evapotranspiration_raster``

    Returns
    -------
    tuple
        The two output memory maps (flushed)
    '''
    if isinstance(driver_paths, dict):
        missing = [k for k in DRIVER_NAMES if k not in driver_paths]
        if missing:
            raise KeyError('missing driver files: %s' % ', '.join(missing))
        driver_paths = [driver_paths[k] for k in DRIVER_NAMES]
    if len(driver_paths) != len(DRIVER_NAMES):
        raise ValueError('expected %d driver files' % len(DRIVER_NAMES))
    cls = np.load(cls_path, mmap_mode='r')
    drivers = [np.load(p, mmap_mode='r') for p in driver_paths]
    dtype = drivers[0].dtype
    if dtype not in (np.float64, np.float32):
        raise TypeError('drivers must be float64 or float32')
    for name, d in zip(DRIVER_NAMES, drivers):
        if d.shape != cls.shape or d.dtype != dtype or not d.flags.c_contiguous:
            raise ValueError('%s: expected a C-ordered %s array of shape %s' % (name, dtype, cls.shape))
    outs = [np.lib.format.open_memmap(p, mode='w+', dtype=dtype, shape=cls.shape)
            for p in (out_day_path, out_night_path)]
    evapotranspiration_raster(bplut, cls, *drivers, beta=beta, math=math, device=device, out=outs,
                              devices=devices)
    for o in outs:
        o.flush()
    return tuple(outs)


# ---------------------------------------------------------------- the store
#: (T x N) float fields in ``mod16_raw_driver`` order without the static ones:
#: (raw driver index, dataset path) -- names as in the reference's Cal-Val
#: layout (mod16/calibration.py:50-112; *_daytime / *_nighttime pairs are what
#: its ``_load_data`` reads, :380-401)
DYNAMIC_FIELDS = (
    (0, 'MERRA2/LWGNT_daytime'), (1, 'MERRA2/LWGNT_nighttime'),
    (2, 'MERRA2/SWGDN_daytime'), (4, 'MODIS/MCD43GF_black_sky_sw_albedo'),
    (5, 'MERRA2/T10M_daytime'), (6, 'MERRA2/T10M_nighttime'), (8, 'MERRA2/Tmin'),
    (9, 'MERRA2/QV10M_daytime'), (10, 'MERRA2/QV10M_nighttime'),
    (11, 'MERRA2/PS_daytime'), (12, 'MERRA2/PS_nighttime'))
#: (N,) float fields: annual mean temperature (the reference's ``MAT`` look-up,
#: :388) and elevation (:405-408)
STATIC_FIELDS = ((7, 'MERRA2/T10M_annual'), (13, 'state/elevation_m'))
#: uint8 fields: MODIS fPAR [%] and LAI [x 10] (T x N), land cover (N)
FPAR, LAI, PFT = 'MODIS/MOD15A2HGF_fPAR', 'MODIS/MOD15A2HGF_LAI', 'state/PFT'
OUT_DAY, OUT_NIGHT = 'out/ET_daytime', 'out/ET_nighttime'


class _Npy(object):
    '''One ``.npy`` file of the store, read and written with positional I/O
    straight from / into caller buffers (no memory map, no intermediate copy).'''

    def __init__(self, path, mode='r'):
        with open(path, 'rb') as f:
            version = np.lib.format.read_magic(f)
            read = np.lib.format.read_array_header_1_0 if version == (1, 0) \
                else np.lib.format.read_array_header_2_0
            self.shape, fortran, self.dtype = read(f)
            self.offset = f.tell()
        if fortran:
            raise ValueError('%s: Fortran-ordered array' % path)
        self.fd = os.open(path, os.O_RDONLY if mode == 'r' else os.O_RDWR)
        self.row = int(np.prod(self.shape[1:], dtype=np.int64)) if len(self.shape) > 1 else 0

    def _pos(self, t, p0):
        return self.offset + ((t * self.row if len(self.shape) > 1 else 0) + p0) * self.dtype.itemsize

    def read_into(self, buf, t, p0):
        '''pixels [p0, p0 + len(buf)) of step t -> the numpy array ``buf``'''
        view = memoryview(buf).cast('B')
        done, pos = 0, self._pos(t, p0)
        while done < len(view):
            got = os.preadv(self.fd, [view[done:]], pos + done)
            if got <= 0:
                raise IOError('short read')
            done += got
        return len(view)

    def write_from(self, buf, t, p0):
        view = memoryview(buf).cast('B')
        done, pos = 0, self._pos(t, p0)
        while done < len(view):
            done += os.pwrite(self.fd, view[done:], pos + done)
        return len(view)

    def close(self):
        if self.fd is not None:
            os.close(self.fd)
            self.fd = None


class RasterStore(object):
    '''
    A raster time series on disk: ``n_steps`` time steps of ``n_pixels`` pixels,
    one ``.npy`` file per dataset of the Cal-Val layout under ``root``
    (``root/MERRA2/LWGNT_daytime.npy`` ...). ``create`` makes an empty store,
    ``RasterStore(root)`` opens one.
    '''

    def __init__(self, root):
        self.root = root
        probe = _Npy(self.path(DYNAMIC_FIELDS[0][1]))
        self.n_steps, self.n_pixels = probe.shape
        self.dtype = probe.dtype
        probe.close()

    def path(self, name):
        return os.path.join(self.root, name + '.npy')

    @classmethod
    def create(cls, root, n_steps, n_pixels, dtype=np.float32):
        '''Create the files of an empty store (sparse until written).'''
        dtype = np.dtype(dtype)
        specs = [(name, (n_steps, n_pixels), dtype) for _, name in DYNAMIC_FIELDS] + \
                [(name, (n_pixels,), dtype) for _, name in STATIC_FIELDS] + \
                [(FPAR, (n_steps, n_pixels), np.uint8), (LAI, (n_steps, n_pixels), np.uint8),
                 (PFT, (n_pixels,), np.uint8),
                 (OUT_DAY, (n_steps, n_pixels), dtype), (OUT_NIGHT, (n_steps, n_pixels), dtype)]
        for name, shape, dt in specs:
            path = os.path.join(root, name + '.npy')
            os.makedirs(os.path.dirname(path), exist_ok=True)
            m = np.lib.format.open_memmap(path, mode='w+', dtype=dt, shape=shape)
            del m
        return cls(root)

    def array(self, name, mode='r'):
        '''Memory map of one dataset (tests, small stores).'''
        return np.load(self.path(name), mmap_mode=mode)


def run_store(bplut, root, tile_pixels=1 << 22, workers=4, beta=None, math=_lib.MATH_FAST,
              device=0, readers=3, devices=None):
    '''
    Forward run over every step and pixel of the store at ``root``: reads the raw
    fields, writes ``out/ET_daytime`` and ``out/ET_nighttime`` [kg m-2 s-1], tile
    by tile through ``workers`` concurrent pipelines (each: page-locked buffers,
    device buffers, a stream and a context of its own; ``readers`` threads per
    pipeline share the positional reads and writes of a step -- one thread copies
    13-17 GB/s out of the page cache, which was the bound of the single-reader
    pipeline of round 2). Short-wave radiation at night is zero, as in the
    reference (calibration.py:383). ``devices``: several GPUs (``mod16_amd.multi``) --
    ``workers`` pipelines on EACH listed device, all taking tile jobs from the one
    queue; a tile's results do not depend on which pipeline ran it, so the files are the
    same bytes whatever the list.

    Returns a report: per stage (``read``, ``h2d``, ``kernel``, ``d2h``,
    ``write``) the bytes moved, the busy seconds summed over the workers and the
    rate while busy, plus the workers' set-up time (buffers, streams, contexts),
    the wall time of the run itself, sustained pixels/s and file bytes/s.
    '''
    import torch
    from . import multi
    from .raster import RasterEngine
    from .utils import bplut_table
    devs = multi.device_list(devices) or [int(device)]
    table = bplut_table(bplut, beta=beta) if isinstance(bplut, dict) else np.array(bplut, np.float64)
    store = RasterStore(root)
    T, N, dt = store.n_steps, store.n_pixels, store.dtype
    if dt not in (np.float32, np.float64):
        raise TypeError('store fields must be float32 or float64')
    tdt = torch.float32 if dt == np.float32 else torch.float64
    tile = int(min(tile_pixels, N))
    tile -= tile % 4 if tile >= 4 else 0
    jobs = [(p0, min(tile, N - p0)) for p0 in range(0, N, tile)]
    lock = threading.Lock()
    nxt = [0]
    stats = {k: [0, 0.0] for k in ('read', 'h2d', 'kernel', 'd2h', 'write')}
    errors = []

    nthreads = max(1, min(max(1, int(workers)) * len(devs), len(jobs)))
    ready = threading.Barrier(nthreads + 1)
    esz = dt.itemsize

    def worker(device):
        try:
            torch.cuda.set_device(device)
            eng = RasterEngine(table, device=device, dtype=dt.name, math=math)
            files = {name: _Npy(store.path(name)) for _, name in DYNAMIC_FIELDS + STATIC_FIELDS}
            files.update({name: _Npy(store.path(name)) for name in (FPAR, LAI, PFT)})
            outs = {name: _Npy(store.path(name), 'w') for name in (OUT_DAY, OUT_NIGHT)}
            stream = torch.cuda.Stream(device=device)
            pin = lambda shape, d: torch.empty(shape, dtype=d, pin_memory=True)
            dev = lambda shape, d: torch.empty(shape, dtype=d, device='cuda:%d' % device)
            nd = len(DYNAMIC_FIELDS)
            # two sets of buffers: the next step is read from the files while the
            # copies and the kernel of the current one are in flight
            sets = [dict(h_dyn=pin((nd, tile), tdt), d_dyn=dev((nd, tile), tdt),
                         h_u8=pin((2, tile), torch.uint8), d_u8=dev((2, tile), torch.uint8),
                         h_out=pin((2, tile), tdt), d_out=dev((2, tile), tdt),
                         ev=[torch.cuda.Event(enable_timing=True) for _ in range(4)])
                    for _ in range(2)]
            h_sta, d_sta = pin((len(STATIC_FIELDS), tile), tdt), dev((len(STATIC_FIELDS), tile), tdt)
            h_pft, d_pft = pin((tile,), torch.uint8), dev((tile,), torch.uint8)
            zero = torch.zeros(1, dtype=tdt, device='cuda:%d' % device)
            local = {k: [0, 0.0] for k in stats}
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max(1, int(readers)))
            ready.wait()

            def read_step(b, t, p0, m):
                t0 = time.perf_counter()
                todo = [(files[name], b['h_dyn'][k, :m].numpy()) for k, (_, name) in enumerate(DYNAMIC_FIELDS)]
                todo += [(files[FPAR], b['h_u8'][0, :m].numpy()), (files[LAI], b['h_u8'][1, :m].numpy())]
                # (preadv releases the GIL: the fields of a step are read side by side)
                nb = sum(pool.map(lambda fa: fa[0].read_into(fa[1], t, p0), todo))
                local['read'][0] += nb
                local['read'][1] += time.perf_counter() - t0
                b['nbytes'] = nb

            def enqueue(b, m):
                ev = b['ev']
                with torch.cuda.stream(stream):
                    ev[0].record(stream)
                    b['d_dyn'][:, :m].copy_(b['h_dyn'][:, :m], non_blocking=True)
                    b['d_u8'][:, :m].copy_(b['h_u8'][:, :m], non_blocking=True)
                    ev[1].record(stream)
                    raw = [None] * 14
                    for k, (idx, _) in enumerate(DYNAMIC_FIELDS):
                        raw[idx] = b['d_dyn'][k, :m]
                    for k, (idx, _) in enumerate(STATIC_FIELDS):
                        raw[idx] = d_sta[k, :m]
                    raw[3] = zero                                   # sw_rad_night
                    eng.run_raw(d_pft[:m], raw, b['d_u8'][0, :m], b['d_u8'][1, :m],
                                out_day=b['d_out'][0, :m], out_night=b['d_out'][1, :m])
                    ev[2].record(stream)
                    b['h_out'][:, :m].copy_(b['d_out'][:, :m], non_blocking=True)
                    ev[3].record(stream)

            def finish(b, t, p0, m):
                ev = b['ev']
                ev[3].synchronize()
                for key, i, j, bytes_ in (('h2d', 0, 1, b['nbytes']),
                                          ('kernel', 1, 2, b['nbytes'] + 2 * m * esz),
                                          ('d2h', 2, 3, 2 * m * esz)):
                    local[key][0] += bytes_
                    local[key][1] += ev[i].elapsed_time(ev[j]) * 1e-3
                t0 = time.perf_counter()
                nb = sum(pool.map(lambda fa: fa[0].write_from(fa[1], t, p0),
                                  [(outs[OUT_DAY], b['h_out'][0, :m].numpy()),
                                   (outs[OUT_NIGHT], b['h_out'][1, :m].numpy())]))
                local['write'][0] += nb
                local['write'][1] += time.perf_counter() - t0

            while True:
                with lock:
                    j = nxt[0]
                    nxt[0] += 1
                if j >= len(jobs) or errors:
                    break
                p0, m = jobs[j]
                # static fields of this tile: once, reused by every step
                t0 = time.perf_counter()
                nb = 0
                for k, (_, name) in enumerate(STATIC_FIELDS):
                    nb += files[name].read_into(h_sta[k, :m].numpy(), 0, p0)
                nb += files[PFT].read_into(h_pft[:m].numpy(), 0, p0)
                local['read'][0] += nb
                local['read'][1] += time.perf_counter() - t0
                with torch.cuda.stream(stream):
                    d_sta[:, :m].copy_(h_sta[:, :m], non_blocking=True)
                    d_pft[:m].copy_(h_pft[:m], non_blocking=True)
                read_step(sets[0], 0, p0, m)
                for t in range(T):
                    cur, nxt_set = sets[t % 2], sets[(t + 1) % 2]
                    enqueue(cur, m)
                    if t + 1 < T:
                        read_step(nxt_set, t + 1, p0, m)    # while step t is on the GPU
                    finish(cur, t, p0, m)
                eng.check()
            pool.shutdown()
            for f in list(files.values()) + list(outs.values()):
                f.close()
            with lock:
                for k in stats:
                    stats[k][0] += local[k][0]
                    stats[k][1] += local[k][1]
        except Exception as exc:          # surfaced by the caller
            errors.append(exc)
            try:
                ready.abort()
            except Exception:
                pass

    t_setup = time.perf_counter()
    # pipeline k runs on devs[k % len(devs)]: with fewer jobs than pipelines every device still
    # gets its share
    threads = [threading.Thread(target=worker, args=(devs[k % len(devs)],)) for k in range(nthreads)]
    for th in threads:
        th.start()
    try:
        ready.wait()            # every worker has its buffers, streams and context
    except threading.BrokenBarrierError:
        pass
    t_wall = time.perf_counter()
    setup = t_wall - t_setup
    for th in threads:
        th.join()
    wall = time.perf_counter() - t_wall
    if errors:
        raise errors[0]
    total_bytes = stats['read'][0] + stats['write'][0]
    report = {'steps': T, 'pixels': N, 'dtype': dt.name, 'tile_pixels': tile,
              'workers': len(threads), 'devices': devs, 'readers_per_worker': max(1, int(readers)), 'setup_s': setup, 'wall_s': wall,
              'pixels_per_s': T * N / wall, 'file_GBps': total_bytes / wall / 1e9,
              'stages': {k: {'bytes': v[0], 'busy_s_sum_over_workers': v[1],
                             'GBps_while_busy': (v[0] / v[1] / 1e9) if v[1] else None}
                         for k, v in stats.items()}}
    return report
