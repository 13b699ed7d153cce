'''
Several GPUs behind ONE call of the numpy entry points (in-process sharding).

``north_star``: "tiles of the global grid shard embarrassingly across the 8 GPUs
of one node". For device-resident rasters that is one process per GPU
(``mod16_amd.dist``, ``bench.py --gpus N``). The numpy entry points --
``MOD16.evapotranspiration`` (reference mod16/__init__.py:675-793),
``evapotranspiration_raster``, ``evapotranspiration_raw``, ``io.run_store``,
``raster.ShardedSeries`` -- move every byte over PCIe and are bound by ONE link
(0.5-0.95 Gpixel/s) on a box that has eight, so they take ``devices=[...]``: the
flattened pixel range is cut at the boundaries of the HOST mode's staging tiles
(``mod16_host_tile_pixels()``) and dealt over the listed devices in order, one
host thread + library context + staging slabs per list entry, every shard
writing its part of the one caller-visible result. No collective, no
``torch.distributed``: the pixels are independent and the host arrays are
shared memory already.

What does not depend on the device list: the outputs (a pixel's arithmetic does
not depend on where its tile ran) and the diagnostics (one vector per staging
tile, folded in tile order: ``mod16_et_hdiag_*`` + ``mod16_fold_diag_host``) --
``devices=[0]``, ``[0, 0]`` and ``range(8)`` return the same bits. A device may
be listed more than once (two contexts on one GPU: how the tests rehearse it).
'''
import queue
import threading

from . import _lib, dist


def device_list(devices):
    '''``devices`` argument -> list of device indices, or None for the
    single-device path (``devices`` not given).'''
    if devices is None:
        return None
    devs = [int(d) for d in devices]
    if not devs:
        raise ValueError('devices must name at least one GPU')
    if min(devs) < 0:
        raise ValueError('negative device index')
    return devs


def host_tile():
    '''Pixels per staging tile of the library's HOST mode.'''
    return int(_lib.load().mod16_host_tile_pixels())


def shards(n, parts, align=1):
    '''Cuts pixels [0, n) into ``parts`` contiguous ranges ``(offset, count)``,
    in order, every boundary a multiple of ``align`` (the last range takes the
    ragged end). Ranges differ by at most ``align`` pixels plus the ragged
    end; with fewer than ``parts`` units of ``align`` the trailing ranges are
    empty ``(n, 0)``. The band rule of ``dist.band`` applied to units.'''
    n, parts, align = int(n), int(parts), max(1, int(align))
    if n < 0 or parts < 1:
        raise ValueError('n >= 0 and parts >= 1 are required')
    units = -(-n // align)
    out = []
    for r in range(parts):
        u0, u1 = dist.band(units, r, parts)
        lo, hi = min(n, u0 * align), min(n, u1 * align)
        out.append((lo, hi - lo))
    return out


class _Worker(threading.Thread):
    '''A host thread bound to one entry of a device list. It lives as long as
    the process: its library context (``_lib.context`` is per thread) keeps the
    staging slabs, streams and BPLUT copy between calls.'''

    def __init__(self, slot, device):
        super().__init__(name='mod16-dev%d-slot%d' % (device, slot), daemon=True)
        self.device = device
        self.jobs = queue.Queue()
        self.start()

    def run(self):
        while True:
            fn, box, done = self.jobs.get()
            if fn is None:                      # shutdown(): the thread's context goes with it
                table = getattr(_lib._local, 'contexts', None) or {}
                for ctx in table.values():
                    ctx.close()
                table.clear()
                done.set()
                return
            try:
                box.append((True, fn(_lib.context(self.device))))
            except BaseException as exc:        # handed to the caller's thread
                box.append((False, exc))
            finally:
                done.set()


_workers = {}
_workers_lock = threading.Lock()
#: one sharded call at a time per process: the workers (and their contexts) are shared
_call_lock = threading.Lock()


def _worker(slot, device):
    with _workers_lock:
        w = _workers.get((slot, device))
        if w is None:
            w = _workers[(slot, device)] = _Worker(slot, device)
        return w


def run(devices, fn):
    '''Calls ``fn(index, ctx)`` for every entry of ``devices`` concurrently, entry
    ``index`` on its own host thread with that thread's library context ``ctx`` on
    ``devices[index]``. Returns the results in device-list order. If shards fail,
    every shard is still waited for and the FIRST failure in list order is
    raised as it is (``IndexError`` for a class code >= 13, ``Mod16Error`` ...).'''
    with _call_lock:
        pending = []
        for i, d in enumerate(devices):
            box, done = [], threading.Event()
            _worker(i, d).jobs.put((lambda ctx, i=i: fn(i, ctx), box, done))
            pending.append((box, done))
        results = []
        for box, done in pending:
            done.wait()
            results.append(box[0])
    for ok, value in results:
        if not ok:
            raise value
    return [value for _, value in results]


def shutdown():
    '''Ends the worker threads of the sharded calls and frees what they hold -- per (list position,
    device) a library context with up to ~4.7 GB of staging slabs in HBM and its page-locked
    buffers, which otherwise live as long as the process. Waits for a sharded call
    in flight; the next ``devices=[...]`` call starts new workers.'''
    with _call_lock:
        with _workers_lock:
            workers = list(_workers.values())
            _workers.clear()
        waits = []
        for w in workers:
            done = threading.Event()
            w.jobs.put((None, None, done))
            waits.append((w, done))
        for w, done in waits:
            done.wait()
            w.join(timeout=10)
    return len(workers)


def fold_diag(parts):
    '''The fixed-order fold of an ``(m, 8)`` float64 array of diagnostics
    vectors (``raster.DIAG_FIELDS``): sums and counts added first to last, maxima
    maximised (``mod16_fold_diag_host``).'''
    import numpy as np
    parts = np.ascontiguousarray(parts, np.float64).reshape(-1, 8)
    out = np.empty(8, np.float64)
    rc = _lib.load().mod16_fold_diag_host(parts.ctypes.data, parts.shape[0], out.ctypes.data)
    if rc != _lib.OK:
        raise ValueError('fold_diag needs at least one diagnostics vector')
    return out
