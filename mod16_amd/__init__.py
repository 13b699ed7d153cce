r'''
MI355X-native MOD16 forward run.

Drop-in for the forward-run surface of the reference package ``mod16``
(arthur-e/MOD16 v1.2.0): the ``MOD16`` class (reference mod16/__init__.py:124),
``MOD16.evapotranspiration()`` (:675-793), the BPLUT look-up
(``mod16_amd.utils.restore_bplut``, ``mod16_amd.models.MOD16Collection61``)
and the module constants keep their names, argument order, units and error
behaviour. The per-pixel Penman-Monteith stack itself runs as one fused HIP
kernel on gfx950 through ``libmod16hip.so`` (C ABI: ``include/mod16_hip.h``).
There is no CPU fallback: without the library or without an MI355X the
forward run raises.

The sub-methods of the class (``evaporation_soil``, ``transpiration``,
``radiation_soil`` ...) and the module functions (``svp``, ``svp_slope`` ...) run
on the GPU too (``mod16_method_*``, reference operation order).

Two ways in:

- ``MOD16(params).evapotranspiration(*drivers)`` -- numpy in, numpy out, as
  the reference; ``params`` values may be scalars (one land-cover type) or
  arrays broadcastable against the drivers (``params_dict[key][pft_map]``).
- ``evapotranspiration_raster(bplut, cls, *drivers)`` -- multi-class rasters:
  the BPLUT is held in LDS and indexed by the uint8 class raster in-kernel,
  which replaces the reference idiom of gathering 11 per-pixel parameter
  arrays first. ``mod16_amd.raster.RasterEngine`` is the device-resident
  (zero-copy, asynchronous) form of the same call for large grids.
'''

__version__ = 'v1.2.0+mi355x.r1'

import ctypes as _ctypes

import numpy as np
import threading as _threading

from . import _lib

# Module constants, same names and values as reference mod16/__init__.py:106-118
PFT_VALID = (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12)
STEFAN_BOLTZMANN = 5.67e-8  # W m-2 K-4
SPECIFIC_HEAT_CAPACITY_AIR = 1013  # J kg-1 K-1
MOL_WEIGHT_WET_DRY_RATIO_AIR = 0.622
TEMP_LAPSE_RATE = 0.0065  # -(deg K) m-1
GRAV_ACCEL = 9.80665  # m s-2
GAS_LAW_CONST = 8.3143  # m3 Pa (mol)-1 K-1
AIR_MOL_WEIGHT = 28.9644e-3  # kg (mol)-1
STD_TEMP_K = 288.15
STD_PRESSURE_PASCALS = 101325.0
AIR_PRESSURE_RATE = GRAV_ACCEL / (
    TEMP_LAPSE_RATE * (GAS_LAW_CONST / AIR_MOL_WEIGHT))

DRIVER_NAMES = (
    'lw_net_day', 'lw_net_night', 'sw_rad_day', 'sw_rad_night', 'sw_albedo',
    'temp_day', 'temp_night', 'temp_annual', 'tmin', 'vpd_day', 'vpd_night',
    'pressure', 'fpar', 'lai')


def pinned_empty(shape, dtype=np.float64):
    '''(Extension.) ``numpy.empty(shape, dtype)`` in page-locked host memory -- for DRIVER arrays
    that are filled once and handed to the numpy entry points again and again (a time loop, a
    raster read from disk into it). A host-to-device copy from ordinary (pageable) numpy memory
    goes through the HIP runtime's own staging buffers on the calling threads: a CPU copy per
    byte, 13-17 GB/s per thread, which the HOST mode hides behind eight threads per GPU -- but
    with ``devices=range(8)`` that is 64 threads copying at once and the host's memory, not the
    eight PCIe links, becomes the bound. From page-locked memory the copy is pure DMA. The
    arrays come from the same bounded pool as the results (``_lib.pinned``; beyond its bound, or
    without a GPU runtime, an ordinary array comes back) and are ordinary numpy arrays to
    everything else.'''
    return _lib.pinned.empty(tuple(np.atleast_1d(shape)) if not isinstance(shape, tuple) else shape, np.dtype(dtype))


def _result_dtype(values):
    '''float32 only if every array-like input is float32 (numpy's own rule
    for the reference code); Python scalars are weak.'''
    seen32, others = False, None
    for v in values:
        t = type(v)
        if t is float or t is int:
            continue
        if isinstance(v, (np.ndarray, np.generic)):
            if v.dtype == _F32:
                seen32 = True
            elif others is None:
                others = [v.dtype]
            else:
                others.append(v.dtype)
    if others is None:
        return _F32 if seen32 else _F64
    if seen32:
        others.append(_F32)
    return _F32 if np.result_type(*others) == _F32 else _F64


_F32, _F64 = np.dtype(np.float32), np.dtype(np.float64)


def _address(a):
    '''Address of a C-contiguous array's first element (the buffer protocol is a third of
    the cost of ``a.ctypes.data``, but wants a writeable, non-empty array).'''
    if a.flags.writeable and a.size:
        return _ctypes.addressof(_ctypes.c_char.from_buffer(a))
    return a.__array_interface__['data'][0]


def _broadcast(shapes):
    '''-> (broadcast shape, number of elements); equal shapes -- the usual call -- without numpy.'''
    shape = shapes[0]
    for sh in shapes:
        if sh != shape:
            shape = np.broadcast_shapes(*shapes)
            break
    n = 1
    for extent in shape:
        n *= int(extent)
    return shape, n


def _shape(v):
    '''``np.shape`` without its detour through ``asarray`` for the two common cases.'''
    t = type(v)
    if t is float or t is int:
        return ()
    if t is np.ndarray:
        return v.shape
    return np.shape(v)


def _broadcast_kind(a_shape, size, shape):
    '''How an input of shape ``a_shape`` broadcasts against the full ``shape``,
    as one of the kinds the C ABI takes without making it dense (``mod16_et2_*``):
    scalar, dense, an (N,) row against (..., N), or a (..., 1) column. ``None``:
    another pattern (made dense by the caller).'''
    if size == 1:
        return _lib.BC_SCALAR
    padded = (1,) * (len(shape) - len(a_shape)) + tuple(a_shape)
    if padded == tuple(shape):
        return _lib.BC_DENSE
    if len(shape) >= 2:
        if padded[-1] == shape[-1] and all(x == 1 for x in padded[:-1]):
            return _lib.BC_ROW
        if padded[-1] == 1 and padded[:-1] == tuple(shape[:-1]):
            return _lib.BC_COL
    return None


def _marshal(values, shape, dtype, kinds=False):
    '''-> (keepalive arrays, addresses, element strides) for the C ABI: a
    size-1 input is passed as a broadcast scalar (stride 0), anything else is
    made dense over ``shape`` (stride 1). With ``kinds`` the third list holds
    broadcast kinds (``_lib.BC_*``) and (N,) rows / (..., 1) columns stay as
    small as they are. (The size-1 inputs share ONE small array: a call on the
    scalars of a flux-tower site costs a handful of numpy operations, not 25
    times three.)'''
    keep, ptrs, strides = [], [], []
    scal = base = None
    dtype = np.dtype(dtype)
    esz = dtype.itemsize
    for i, v in enumerate(values):
        t = type(v)
        if t is float or t is int:
            a = None
        else:
            a = v if (t is np.ndarray and v.dtype == dtype) else np.asarray(v, dtype=dtype)
            if a.size == 1:
                v = a.reshape(-1)[0]
                a = None
        if a is None:
            if scal is None:
                scal = np.empty(len(values), dtype)
                base = _address(scal)
                keep.append(scal)
            scal[i] = v
            ptrs.append(base + i * esz)
            strides.append(0)
            continue
        kind = _broadcast_kind(a.shape, a.size, shape) if kinds else None
        if kind in (_lib.BC_ROW, _lib.BC_COL):
            a = np.ascontiguousarray(a.reshape(-1))
            strides.append(kind)
        else:
            if a.shape != shape:
                a = np.broadcast_to(a, shape)
            if not a.flags.c_contiguous:
                a = np.ascontiguousarray(a)
            strides.append(1)
        keep.append(a)
        ptrs.append(_address(a))
    return keep, ptrs, strides


def _shift(ptrs, kinds, off, inner, itemsize):
    '''The addresses of pixel ``off`` onwards: dense arrays move by ``off``
    elements, (T, 1) columns by ``off // inner`` (``off`` is then a multiple of
    ``inner``), scalars and (N,) rows stay.'''
    if ptrs is None or off == 0:
        return ptrs
    step = {_lib.BC_SCALAR: 0, _lib.BC_DENSE: off, _lib.BC_ROW: 0, _lib.BC_COL: off // max(inner, 1)}
    return [p + step[k] * itemsize if p is not None else None for p, k in zip(ptrs, kinds)]


def _forward(cls, drivers, params, separate, flags, device, pet=False, out=None,
             devices=None, table=None, diagnostics=False):
    '''Shared host path of MOD16.evapotranspiration and
    evapotranspiration_raster: marshal numpy inputs, run mod16_et_* in HOST
    mode, shape the outputs as the reference does (mod16/__init__.py:789-793).
    ``devices``: the pixel range dealt over several GPUs (``mod16_amd.multi``);
    ``table``: the BPLUT every context that takes part is given first;
    ``diagnostics``: also the vector of ``raster.DIAG_FIELDS`` (per staging
    tile on the device, folded in tile order).
    '''
    from . import multi
    devs = multi.device_list(devices)
    values = list(drivers) + (list(params) if params is not None else [])
    dtype = _result_dtype(values)
    shapes = [_shape(v) for v in values]
    if cls is not None:
        cls = np.asarray(cls)
        if cls.dtype != np.uint8:
            if cls.size and (cls.min() < 0 or cls.max() > 255):
                raise IndexError('class code outside [0, 255]')
            cls = cls.astype(np.uint8)
        shapes.append(cls.shape)
    shape, n = _broadcast(shapes)
    if diagnostics and (pet or separate):
        raise ValueError('diagnostics come with the (day, night) totals only')
    # (N,) rows and (..., 1) columns against (..., N) drivers are not made dense
    # (reference mod16/__init__.py:180-181): the C ABI takes them as they are
    two_level = not pet and not diagnostics and len(shape) >= 2 and n > 0
    keep_d, dptr, dstride = _marshal(drivers, shape, dtype, kinds=two_level)
    pptr = pstride = cptr = None
    ckind = _lib.BC_DENSE
    if cls is not None:
        k = _broadcast_kind(cls.shape, cls.size, shape) if two_level else None
        if k in (_lib.BC_ROW, _lib.BC_COL):
            cls, ckind = np.ascontiguousarray(cls.reshape(-1)), k
        else:
            cls = np.ascontiguousarray(np.broadcast_to(cls, shape))
        cptr = cls.ctypes.data
    else:
        keep_p, pptr, pstride = _marshal(params, shape, dtype, kinds=two_level)
    two_level = two_level and (ckind != _lib.BC_DENSE or max(dstride) > 1 or
                               (pstride is not None and max(pstride) > 1))
    inner = shape[-1] if two_level else 1
    nout = 4 if pet else (6 if separate else 2)
    if out is not None:      # caller's arrays (e.g. memory-mapped files), written in place
        outs = list(out)
        if len(outs) != nout:
            raise ValueError('out must hold %d arrays' % nout)
        for o in outs:
            if not (isinstance(o, np.ndarray) and o.shape == shape and o.dtype == dtype
                    and o.flags.c_contiguous and o.flags.writeable):
                raise ValueError('out arrays must be writeable C-contiguous %s arrays of shape %s'
                                 % (dtype, shape))
    else:
        outs = [_lib.pinned.empty(shape, dtype) for _ in range(nout)]
    optr = [o.ctypes.data for o in outs]
    esz = dtype.itemsize
    tile = multi.host_tile()
    tile_diag = np.empty((max(1, -(-n // tile)), 8), np.float64) if diagnostics else None

    def part(ctx, off, m):
        '''pixels [off, off + m) on the calling thread's context'''
        if table is not None:
            ctx.set_bplut(table)
        if m <= 0:
            return
        d = _shift(dptr, dstride, off, inner, esz)
        p = _shift(pptr, pstride, off, inner, esz)
        c = _shift([cptr], [ckind], off, inner, 1)[0]
        o = [a + off * esz for a in optr]
        if pet:
            fn = ctx.lib.mod16_et_pet_f32 if dtype == np.float32 else ctx.lib.mod16_et_pet_f64
            ctx.check(fn(
                ctx.handle, c, _lib.ptr_array(d), _lib.i64_array(dstride),
                _lib.ptr_array(p) if p is not None else None,
                _lib.i64_array(pstride) if pstride is not None else None, m,
                o[0], o[1], o[2], o[3], int(flags), _lib.HOST, None))
            return
        if diagnostics:
            fn = ctx.lib.mod16_et_hdiag_f32 if dtype == np.float32 else ctx.lib.mod16_et_hdiag_f64
            ctx.check(fn(
                ctx.handle, c, _lib.ptr_array(d), _lib.i64_array(dstride),
                _lib.ptr_array(p) if p is not None else None,
                _lib.i64_array(pstride) if pstride is not None else None, m,
                o[0], o[1], int(flags), tile_diag[off // tile:].ctypes.data))
            return
        day, night, sep = (None, None, o) if separate else (o[0], o[1], None)
        if two_level:
            ctx.et2(dtype, c, ckind, d, dstride, p, pstride, inner, m, day, night,
                    sep, flags=flags, where=_lib.HOST)
        else:
            ctx.et(dtype, c, d, dstride, p, pstride, m, day, night, sep,
                   flags=flags, where=_lib.HOST)

    if devs is None:
        part(_lib.context(device), 0, n)
    else:
        # whole staging tiles per device (two-level shapes: whole rows), so every tile is the
        # tile of the undivided call
        cuts = multi.shards(n, len(devs), inner if two_level else tile)
        multi.run(devs, lambda i, ctx: part(ctx, *cuts[i]))
    if not shape:
        # all-scalar input, as the reference returns it: the totals (sums, :792) and the soil component
        # (e / lhv, :864) are numpy scalars, the canopy and transpiration components -- np.where
        # results, :961 and :1258 -- stay 0-d arrays
        outs = [o if separate and k % 3 != 1 else o[()] for k, o in enumerate(outs)]
    if pet:
        return tuple(outs)
    if separate:
        return (tuple(outs[0:3]), tuple(outs[3:6]))
    if diagnostics:
        return (outs[0], outs[1], multi.fold_diag(tile_diag) if n else
                np.array([0, 0, 0, 0, 0, 0, -np.inf, -np.inf]))
    return (outs[0], outs[1])


def _is_device_tensor(v):
    return getattr(v, 'is_cuda', False) and hasattr(v, 'data_ptr')


_SPREAD_SCALARS_FROM = 1 << 20
# What the device-tensor path keeps between calls. Launches there are ASYNCHRONOUS on the caller's
# stream, so nothing a launch reads may change or be freed under it:
#  - per GPU one class raster of ones (uint8, one byte per pixel of the largest raster seen), shared
#    by the threads of the process and only ever read; it is replaced -- under the lock, behind a
#    wait for the device -- when a larger raster arrives;
#  - per thread, GPU and PARAMETER SET one library context whose parameter table is written once,
#    when the context is made, and never again: the numpy entry points (which set the table of the
#    thread's own context, _lib.context, with a blocking copy) never touch a table an asynchronous
#    launch reads, and a loop over plant functional types -- the reference's usual pattern -- pays
#    no synchronisation and no copy per call. At most _TENSOR_CONTEXTS per thread; the least
#    recently used one is closed behind a wait for its device.
_ONES = {}
_ONES_LOCK = _threading.Lock()
_TENSOR_CONTEXTS = 16
_tensor_local = _threading.local()


def release_device_cache():
    '''Frees what the device-tensor path keeps between calls: the class rasters of ones (all
    threads) and the calling thread's per-parameter-set contexts. Waits for the devices first.'''
    import torch
    with _ONES_LOCK:
        for index in list(_ONES):
            torch.cuda.synchronize(index)
        _ONES.clear()
    table = getattr(_tensor_local, 'contexts', None)
    if table:
        for (index, _), ctx in list(table.items()):
            torch.cuda.synchronize(index)
            ctx.close()
        table.clear()


def _class_of_ones(torch, dev, index, n):
    with _ONES_LOCK:
        t = _ONES.get(index)
        if t is None or t.numel() < n:
            if t is not None:
                torch.cuda.synchronize(dev)      # launches in flight (any stream) still read the old one
            t = _ONES[index] = torch.ones(n, dtype=torch.uint8, device=dev)
            torch.cuda.current_stream(dev).synchronize()     # filled before any other stream reads it
        return t


def _tensor_context(torch, dev, index, table):
    '''The calling thread's context on GPU ``index`` whose parameter table is ``table``.'''
    import collections
    cache = getattr(_tensor_local, 'contexts', None)
    if cache is None:
        cache = _tensor_local.contexts = collections.OrderedDict()
    key = (index, table.tobytes())
    ctx = cache.get(key)
    if ctx is not None:
        cache.move_to_end(key)
        return ctx
    while len(cache) >= _TENSOR_CONTEXTS:
        (old_index, _), old = cache.popitem(last=False)
        torch.cuda.synchronize(old_index)            # its launches may still read its table
        old.close()
    ctx = cache[key] = _lib.Context(index)
    ctx.set_bplut(table)                             # once: nothing has been launched on it yet
    return ctx


_GATHER_SAMPLE = 1 << 16        # pixels looked at for the candidate rows of a gathered parameter set


def _gathered_classes(torch, dev, index, values, shape, n, dtype, np_dtype, cache):
    '''The reference's multi-class idiom on device tensors -- ``MOD16({k: bplut[k][pft_map]})``,
    notebook cell 32: eleven parameter rasters that hold, pixel for pixel, one of at most 13 rows of
    a table. Returns ``(table, cls)`` -- the (13, 11) float64 table of the distinct rows and a uint8
    class raster on the device -- if ``values[14:]`` are such rasters (tensors of the drivers' shape,
    or single values), else None. The rows come from a strided sample of the rasters (their bit
    patterns: a NaN row, an invalid class, is a row like any other); ``mod16_classify_*`` then
    compares EVERY pixel with them, and a pixel that matches none hands in its row for another
    pass -- a raster with more than 13 distinct rows is not a gather and takes the plain kernel.
    The answer is kept in ``cache`` (the model instance's) for as long as the parameter tensors are
    the same objects, unchanged (``Tensor._version``): a model called once per time step looks at its
    parameters once.'''
    pars = values[14:]
    if not any(_is_device_tensor(v) and v.numel() > 1 for v in pars):
        return None
    key = tuple((id(v), v.data_ptr(), v._version, tuple(v.shape), str(v.dtype)) if _is_device_tensor(v)
                else ('scalar', float(np.asarray(v).reshape(()))) for v in pars) + (shape, str(dtype), index)
    if cache is not None and cache.get('key') == key:
        return cache.get('answer')
    answer = None
    flat = []
    for v in pars:
        if _is_device_tensor(v):
            t = v if v.dtype == dtype else v.to(dtype)
            if t.numel() > 1:
                if tuple(t.shape) != shape:
                    t = t.expand(shape)
                t = t.contiguous().reshape(-1)
            else:
                t = t.reshape(1)
        else:
            t = torch.tensor([float(np.asarray(v).reshape(()))], dtype=dtype, device=dev)
        flat.append(t)
    ibits = torch.int64 if dtype == torch.float64 else torch.int32
    step = max(1, n // _GATHER_SAMPLE)
    cols = [(t[::step] if t.numel() > 1 else t.expand((n + step - 1) // step)).view(ibits) for t in flat]
    rows = torch.unique(torch.stack(cols, dim=1), dim=0)
    if rows.shape[0] <= _lib.N_CLASSES:
        rows = rows.view(dtype).cpu().numpy()                       # (r, 11), the sample's distinct rows
        cls = torch.empty(n, dtype=torch.uint8, device=dev)
        ctx = _lib.context(index)
        fn = ctx.lib.mod16_classify_f32 if np_dtype == _F32 else ctx.lib.mod16_classify_f64
        stream = _ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        ptrs = _lib.ptr_array([t.data_ptr() for t in flat])
        strides = _lib.i64_array([1 if t.numel() > 1 else 0 for t in flat])
        missing = _ctypes.c_int64(-1)
        while True:
            host_rows = np.ascontiguousarray(rows, np_dtype)
            ctx.check(fn(ctx.handle, ptrs, strides, n, host_rows.ctypes.data, int(host_rows.shape[0]),
                         cls.data_ptr(), _ctypes.byref(missing), stream))
            if missing.value < 0:
                table = np.full((_lib.N_CLASSES, _lib.N_PARAMS), np.nan)
                table[:host_rows.shape[0]] = host_rows.astype(np.float64)
                answer = (table, cls)
                break
            if rows.shape[0] >= _lib.N_CLASSES:
                break                                               # a 14th row: not a gather of a table
            i = int(missing.value)
            extra = np.array([[float(t[i if t.numel() > 1 else 0]) for t in flat]], np_dtype)
            rows = np.concatenate([rows, extra], axis=0)
    if cache is not None:
        cache.clear()
        cache.update(key=key, answer=answer, keep=pars)     # (the tensors stay alive: ids are not reused under the key)
    return answer


def _forward_device(drivers, params, separate, flags, pet=False, cache=None):
    '''``MOD16.evapotranspiration`` on ``torch`` tensors that live on the GPU (extension): the
    same entry points in DEVICE mode -- zero-copy, asynchronous on the current stream of the
    tensors' device -- and ``torch`` tensors back, shaped as the reference shapes its arrays.
    Inputs are device tensors (broadcast against each other as numpy would) and plain numbers
    (Python / numpy scalars, weak as in the dtype rule: float32 only if every tensor is); host
    arrays cannot be mixed in. Nothing is checked on the host afterwards: the result is valid when
    the stream has run (``torch.cuda.synchronize()`` or any stream-ordered use).'''
    import torch
    values = list(drivers) + list(params)
    tens = [v for v in values if _is_device_tensor(v)]
    dev = tens[0].device
    for v in values:
        if _is_device_tensor(v):
            if v.device != dev:
                raise ValueError('device tensors on different GPUs: %s and %s' % (dev, v.device))
        elif np.size(v) != 1:
            raise TypeError('device tensors and host arrays cannot be mixed in one call '
                            '(move the arrays to %s, or the tensors to the host)' % (dev,))
    f32 = all(t.dtype == torch.float32 for t in tens)
    dtype, np_dtype = (torch.float32, _F32) if f32 else (torch.float64, _F64)
    esz = np_dtype.itemsize
    shape = tuple(torch.broadcast_shapes(*[tuple(t.shape) for t in tens]))
    n = 1
    for extent in shape:
        n *= int(extent)
    host_scalars = [float(np.asarray(v).reshape(())) if not _is_device_tensor(v) else 0.0 for v in values]
    with torch.cuda.device(dev):
        scal = torch.tensor(host_scalars, dtype=dtype).to(dev, non_blocking=False)
        keep, ptrs, strides = [scal], [], []
        # a large raster with one set of parameters runs the production pipeline (below), which wants
        # fourteen dense drivers: a scalar among them (a constant pressure, say) is written out -- 8
        # bytes per pixel more to read, still well ahead of the plain kernel
        spread = n >= _SPREAD_SCALARS_FROM and not any(_is_device_tensor(v) for v in params)
        for i, v in enumerate(values):
            if not _is_device_tensor(v):
                if spread and i < 14:
                    t = scal[i].expand(shape).contiguous()
                    keep.append(t)
                    ptrs.append(t.data_ptr())
                    strides.append(1)
                    continue
                ptrs.append(scal.data_ptr() + i * esz)
                strides.append(0)
                continue
            t = v if v.dtype == dtype else v.to(dtype)
            if t.numel() == 1 and not (spread and i < 14):
                strides.append(0)
            else:
                if tuple(t.shape) != shape:
                    t = t.expand(shape)
                t = t.contiguous()
                strides.append(1)
            keep.append(t)
            ptrs.append(t.data_ptr())
        nout = 4 if pet else (6 if separate else 2)
        outs = [torch.empty(shape, dtype=dtype, device=dev) for _ in range(nout)]
        if n:
            index = dev.index if dev.index is not None else torch.cuda.current_device()
            ctx = _lib.context(index)
            stream = _ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            optr = [o.data_ptr() for o in outs]
            dptr, dstr, pptr, pstr = ptrs[:14], strides[:14], ptrs[14:], strides[14:]
            cptr = None
            gathered = None
            if min(dstr) == 1 and n >= _SPREAD_SCALARS_FROM and any(_is_device_tensor(v) for v in params):
                gathered = _gathered_classes(torch, dev, index, values, shape, n, dtype, np_dtype, cache)
            if gathered is not None:
                # per-pixel parameter tensors that are a gather of <= 13 rows (the reference's multi-class
                # idiom): the production pipeline with the rows as its table and the class raster the
                # library made of the eleven rasters -- 14 drivers + 1 byte per pixel instead of 25 arrays
                table, cls_t = gathered
                ctx = _tensor_context(torch, dev, index, table)
                cptr = cls_t.data_ptr()
                pptr = pstr = None
            elif min(dstr) == 1 and not any(_is_device_tensor(v) for v in params):
                # Dense drivers, one set of parameters (a MOD16Collection61 of one plant functional
                # type: the reference's usual call): the production pipeline instead of the plain
                # kernel (77 instead of 55 % of the HBM peak in float64) -- the parameters as row 1
                # of a parameter table, a class raster of ones. Same arithmetic, same bits.
                table = np.full((_lib.N_CLASSES, _lib.N_PARAMS), np.nan)
                # (float32 rasters take their scalar parameters as float32, as the numpy call does)
                table[1] = np.asarray(host_scalars[14:], np_dtype)
                # a context of its own per parameter set: its table never changes under a launch
                ctx = _tensor_context(torch, dev, index, table)
                cptr = _class_of_ones(torch, dev, index, n).data_ptr()
                pptr = pstr = None
            if pet:
                fn = ctx.lib.mod16_et_pet_f32 if f32 else ctx.lib.mod16_et_pet_f64
                ctx.check(fn(ctx.handle, cptr, _lib.ptr_array(dptr), _lib.i64_array(dstr),
                             _lib.ptr_array(pptr) if pptr is not None else None,
                             _lib.i64_array(pstr) if pstr is not None else None,
                             n, optr[0], optr[1], optr[2], optr[3], int(flags), _lib.DEVICE, stream))
            else:
                day, night, sep = (None, None, optr) if separate else (optr[0], optr[1], None)
                ctx.et(np_dtype, cptr, dptr, dstr, pptr, pstr, n, day, night, sep, flags=flags, where=_lib.DEVICE,
                       stream=stream)
    if pet:
        return tuple(outs)
    if separate:
        return (tuple(outs[0:3]), tuple(outs[3:6]))
    return (outs[0], outs[1])


def _call_method(method, inputs, params=None, nout=1, alpha=1.26, device=0, tiny=1e-7):
    '''Runs one sub-method of the class surface on the GPU (``mod16_method_*``,
    reference operation order). ``inputs`` follows the reference signature,
    ``None`` = optional argument not given; ``params`` maps parameter name ->
    value for the parameters the method uses.'''
    ctx = _lib.context(device)
    present = [v for v in inputs if v is not None]
    params = dict((k, v) for k, v in (params or {}).items() if v is not None)
    pvals = list(params.values())
    dtype = _result_dtype(present + pvals)
    shape, n = _broadcast([_shape(v) for v in present + pvals])
    keep, ptrs, strides = _marshal(present, shape, dtype)
    it = iter(zip(ptrs, strides))
    iptr, istr = [], []
    for v in list(inputs) + [None] * (_lib.METHOD_MAX_IN - len(inputs)):
        ptr, st = next(it) if v is not None else (None, 0)
        iptr.append(ptr)
        istr.append(st)
    pptr = pstr = None
    if params:
        keep_p, pp, ps = _marshal(pvals, shape, dtype)
        by_name = dict(zip(params.keys(), zip(pp, ps)))
        pptr = [by_name.get(k, (None, 0))[0] for k in MOD16.required_parameters]
        pstr = [by_name.get(k, (None, 0))[1] for k in MOD16.required_parameters]
    outs = [np.empty(shape, dtype) for _ in range(nout)]
    if n:
        ctx.method(dtype, method, iptr, istr, pptr, pstr, n,
                   [_address(o) for o in outs] + [None] * (2 - nout), alpha=alpha,
                   tiny=tiny)
    if not shape:
        outs = [o[()] for o in outs]
    return outs[0] if nout == 1 else tuple(outs)


class MOD16(object):
    r'''
    The MODIS MxD16 Evapotranspiration model on MI355X. Same construction as
    the reference class (mod16/__init__.py:124-160); the required model
    parameters are:

    - `tmin_close`: Temperature at which stomata are almost completely
        closed due to (minimum) temperature stress (deg C)
    - `tmin_open`: Temperature at which stomata are completely open (deg C)
    - `vpd_open`: The VPD at which stomata are completely open (Pa)
    - `vpd_close`: The VPD at which stomata are almost completely closed (Pa)
    - `gl_sh`: Leaf conductance to sensible heat per unit LAI (m s-1 LAI-1)
    - `gl_wv`: Leaf conductance to evaporated water per unit LAI
        (m s-1 LAI-1)
    - `g_cuticular`: Leaf cuticular conductance (m s-1)
    - `csl`: Mean potential stomatal conductance per unit leaf area (m s-1)
    - `rbl_min`: Minimum atmospheric boundary layer resistance (s m-1)
    - `rbl_max`: Maximum atmospheric boundary layer resistance (s m-1)
    - `beta`: Factor in soil moisture constraint on potential soil
        evaporation, i.e., (VPD / beta)

    Parameters
    ----------
    params : dict
        Dictionary of model parameters: scalars, or arrays broadcastable
        against the driver arrays (per-pixel parameters)
    device : int
        (Extension) index of the GPU to run on (Default: 0)
    devices : sequence of int
        (Extension) several GPUs for ``evapotranspiration()`` on large arrays: the
        pixels are dealt over them at staging-tile boundaries, one host thread and
        PCIe link each (``mod16_amd.multi``); the results are the same bits as on
        one device. Default: None, the one ``device``.
    '''
    required_parameters = [
        'tmin_close', 'tmin_open', 'vpd_open', 'vpd_close', 'gl_sh', 'gl_wv',
        'g_cuticular', 'csl', 'rbl_min', 'rbl_max', 'beta'
    ]

    #: arithmetic of the fused kernel: _lib.MATH_FAST (default), _lib.MATH_EXACT (reference operation
    #: order, IEEE divide and pow) or, for float32 inputs, _lib.MATH_MIXED (the mixed-precision form:
    #: same masks, no value further than 2e-4 of itself from the float64 arithmetic, 1.3-1.45x faster;
    #: float64 inputs ignore it)
    math = _lib.MATH_FAST

    def __init__(self, params, device=0, devices=None):
        self.params = params
        self.device = device
        self.devices = devices
        for key in self.required_parameters:   # KeyError if one is missing
            setattr(self, key, params[key])

    def _param_values(self):
        return [getattr(self, key) for key in self.required_parameters]

    def evapotranspiration(
            self, lw_net_day, lw_net_night, sw_rad_day, sw_rad_night,
            sw_albedo, temp_day, temp_night, temp_annual, tmin, vpd_day,
            vpd_night, pressure, fpar, lai, f_wet=None, separate=False):
        r'''
        Instantaneous evapotranspiration (ET) [kg m-2 s-1] for day and night,
        the sum of wet-canopy evaporation, bare-soil evaporation and
        transpiration: same arguments, semantics and return value as the
        reference method (mod16/__init__.py:675-793), computed by the fused
        gfx950 kernel.

        Parameters
        ----------
        lw_net_day, lw_net_night : float or numpy.ndarray
            Net downward long-wave radiation integrated over daylight /
            night-time hours (J m-2 s-1)
        sw_rad_day, sw_rad_night : float or numpy.ndarray
            Down-welling short-wave radiation, day / night (J m-2 s-1)
        sw_albedo : float or numpy.ndarray
            Down-welling short-wave albedo
        temp_day, temp_night : float or numpy.ndarray
            Average temperature during daylight / night-time hours (deg K)
        temp_annual : float or numpy.ndarray
            Annual average daily temperature (deg K)
        tmin : float or numpy.ndarray
            Minimum daily temperature (deg K)
        vpd_day, vpd_night : float or numpy.ndarray
            Daytime / night-time mean vapor pressure deficit (Pa)
        pressure : float or numpy.ndarray
            Air pressure (Pa)
        fpar : float or numpy.ndarray
            Fraction of photosynthetically active radiation absorbed [0, 1]
        lai : float or numpy.ndarray
            Leaf area index
        f_wet : float or numpy.ndarray
            Accepted for interface compatibility and ignored, as in the
            reference (it recomputes the wet fraction from VPD, :764)
        separate : bool
            True to return the components (canopy evaporation, soil
            evaporation, transpiration) separately (Default: False)

        Returns
        -------
        tuple
            ``(day, night)``; with ``separate = True`` each of the two is a
            3-tuple ``(canopy, soil, transpiration)``
        '''
        drivers = (
            lw_net_day, lw_net_night, sw_rad_day, sw_rad_night, sw_albedo,
            temp_day, temp_night, temp_annual, tmin, vpd_day, vpd_night,
            pressure, fpar, lai)
        if any(_is_device_tensor(v) for v in drivers):
            return _forward_device(drivers, self._param_values(), separate, self.math,
                                   cache=self.__dict__.setdefault('_gather_cache', {}))
        return _forward(
            None, drivers, self._param_values(), separate, self.math,
            self.device, devices=self.devices)


    def evapotranspiration_and_pet(
            self, lw_net_day, lw_net_night, sw_rad_day, sw_rad_night,
            sw_albedo, temp_day, temp_night, temp_annual, tmin, vpd_day,
            vpd_night, pressure, fpar, lai):
        r'''
        (Extension) ET and potential ET in one
        pass: returns ``(day, night, pet_day, pet_night)`` [kg m-2 s-1].
        Potential ET is the sum the reference's README defines (lines
        404-424): wet-canopy evaporation + saturated-soil evaporation +
        unsaturated-soil evaporation without the soil-moisture constraint +
        ``MOD16.potential_transpiration`` (alpha = 1.26), i.e. with the
        reference's own methods::

            sat, unsat = MOD16.potential_soil_evaporation(...)
            pet = evaporation_wet_canopy(...) + (max(sat, 0) + max(unsat, 0)) / lhv \
                + MOD16.potential_transpiration(...) / lhv
        '''
        drivers = (
            lw_net_day, lw_net_night, sw_rad_day, sw_rad_night, sw_albedo,
            temp_day, temp_night, temp_annual, tmin, vpd_day, vpd_night,
            pressure, fpar, lai)
        if any(_is_device_tensor(v) for v in drivers):
            return _forward_device(drivers, self._param_values(), False, self.math, pet=True,
                                   cache=self.__dict__.setdefault('_gather_cache', {}))
        return _forward(None, drivers, self._param_values(), False, self.math,
                        self.device, pet=True, devices=self.devices)

    # ---- the rest of the reference's class surface, on the GPU as well
    #      (mod16_method_*: reference operation order, IEEE divide / pow)
    def _p(self, *names):
        return dict((k, getattr(self, k)) for k in names)

    @staticmethod
    def _evapotranspiration(
            params, lw_net_day, lw_net_night, sw_rad_day, sw_rad_night,
            sw_albedo, temp_day, temp_night, temp_annual, tmin, vpd_day,
            vpd_night, pressure, fpar, lai, f_wet=None, tiny=1e-7,
            r_corr_list=None):
        '''
        The vectorised calibration interface (reference
        mod16/__init__.py:195-382): ``params`` is the list of the 11
        parameters in ``MOD16.required_parameters`` order (scalars or (1 x N)
        arrays), the drivers are (T x N) arrays. Returns ``[day, night]``
        latent heat flux [W m-2]. Note that this is, in the reference too, a
        numerically different algorithm from ``MOD16.evapotranspiration()``;
        ``f_wet`` is ignored as in the reference (:282).
        '''
        drivers = [lw_net_day, lw_net_night, sw_rad_day, sw_rad_night,
                   sw_albedo, temp_day, temp_night, temp_annual, tmin, vpd_day,
                   vpd_night, pressure, fpar, lai]
        params = list(params)
        if len(params) != 11:
            raise IndexError('params must hold the 11 parameters in '
                             'MOD16.required_parameters order')
        rc = list(r_corr_list) if r_corr_list is not None else []
        values = drivers + params + rc
        dtype = _result_dtype(values)
        shape, n = _broadcast([_shape(v) for v in values])
        keep_d, dptr, dstr = _marshal(drivers, shape, dtype)
        keep_p, pptr, pstr = _marshal(params, shape, dtype)
        keep_r, rptr, rstr = _marshal(rc, shape, dtype) if rc else (None, None, None)
        day, night = np.empty(shape, dtype), np.empty(shape, dtype)
        ctx = _lib.context(0)
        if n:
            fn = ctx.lib.mod16_et_static_f32 if dtype == np.float32 \
                else ctx.lib.mod16_et_static_f64
            ctx.check(fn(
                ctx.handle, _lib.ptr_array(dptr), _lib.i64_array(dstr),
                _lib.ptr_array(pptr), _lib.i64_array(pstr),
                _lib.ptr_array(rptr) if rc else None,
                _lib.i64_array(rstr) if rc else None, n, _address(day),
                _address(night), float(tiny), _lib.HOST, None))
        if not shape:
            return [day[()], night[()]]
        return [day, night]

    @staticmethod
    def _et(params, lw_net_day, lw_net_night, sw_rad_day, sw_rad_night,
            sw_albedo, temp_day, temp_night, temp_annual, tmin, vpd_day,
            vpd_night, pressure, fpar, lai, f_wet=None, tiny=1e-7,
            r_corr_list=None):
        '''Total (day + night) latent heat flux [W m-2], for calibration
        (reference mod16/__init__.py:162-193). As in the reference, ``f_wet``
        and ``tiny`` are accepted and NOT passed on (:190-192 calls
        ``_evapotranspiration`` with ``tiny = 1e-7``).'''
        day, night = MOD16._evapotranspiration(
            params, lw_net_day, lw_net_night, sw_rad_day, sw_rad_night,
            sw_albedo, temp_day, temp_night, temp_annual, tmin, vpd_day,
            vpd_night, pressure, fpar, lai, r_corr_list=r_corr_list)
        return np.add(day, night)

    @staticmethod
    def _et_batch(params, lw_net_day, lw_net_night, sw_rad_day, sw_rad_night,
                  sw_albedo, temp_day, temp_night, temp_annual, tmin, vpd_day,
                  vpd_night, pressure, fpar, lai, observed=None, weights=None,
                  separate=False, math=_lib.MATH_EXACT):
        '''
        ``MOD16._et`` for many parameter vectors in one launch (extension; the
        reference evaluates ``_et`` draw by draw from its MCMC sampler,
        calibration.py:907, and Sobol analysis, sensitivity.py:95).
        ``params`` is a (D x 11) array in ``MOD16.required_parameters`` order;
        the drivers broadcast against each other as in ``_et``. Returns the
        (D x shape) array of day + night latent heat flux [W m-2] -- row d
        equals ``MOD16._et(params[d], *drivers)`` bit for bit -- or, with
        ``separate=True``, ``[day, night]``. ``math=_lib.MATH_FAST`` trades the
        bit-identity for speed: the strength-reduced float64 arithmetic of the
        forward run, within 1e-9 of the default with the same NaN / zero masks (pixels
        outside that arithmetic's domain are computed in the reference's order, as in the
        forward run). A loop over many calls on the same drivers should bind them once:
        ``MOD16._et_bind``. With ``observed`` (and optional
        ``weights``, both of the drivers' shape) nothing of that size comes
        back: the result is ``(sse, count)``, two float64 arrays (D,) with
        ``sse[d] = sum((weights * (_et_d - observed))**2)`` over the non-NaN
        pairs and their number, e.g. ``rmsd = np.sqrt(sse / count)``.
        '''
        drivers = [lw_net_day, lw_net_night, sw_rad_day, sw_rad_night,
                   sw_albedo, temp_day, temp_night, temp_annual, tmin, vpd_day,
                   vpd_night, pressure, fpar, lai]
        params = np.asarray(params)
        if params.ndim != 2 or params.shape[1] != 11:
            raise IndexError('params must be (D x 11), columns in '
                             'MOD16.required_parameters order')
        extra = [v for v in (observed, weights) if v is not None]
        dtype = _result_dtype(drivers + [params] + extra)
        shape = np.broadcast_shapes(*[np.shape(v) for v in drivers + extra])
        n = int(np.prod(shape, dtype=np.int64))
        ndraw = params.shape[0]
        keep_d, dptr, dstr = _marshal(drivers, shape, dtype)
        par = np.ascontiguousarray(params, dtype)
        full = lambda v: np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype), shape))
        obs = full(observed) if observed is not None else None
        wts = full(weights) if weights is not None else None
        if weights is not None and observed is None:
            raise ValueError('weights need observed')
        ctx = _lib.context(0)
        fn = ctx.lib.mod16_et_static_batch_f32 if dtype == np.float32 \
            else ctx.lib.mod16_et_static_batch_f64
        adr = lambda a: a.ctypes.data if a is not None else None
        if obs is not None:
            sse, count = np.zeros(ndraw), np.zeros(ndraw)
            if n and ndraw:
                ctx.check(fn(ctx.handle, _lib.ptr_array(dptr), _lib.i64_array(dstr), n,
                             adr(par), ndraw, None, None, None, adr(obs), adr(wts),
                             adr(sse), adr(count), int(math), _lib.HOST, None))
            return sse, count
        outs = [np.empty((ndraw,) + shape, dtype) for _ in range(2 if separate else 1)]
        if n and ndraw:
            ctx.check(fn(ctx.handle, _lib.ptr_array(dptr), _lib.i64_array(dstr), n,
                         adr(par), ndraw,
                         adr(outs[0]) if separate else None, adr(outs[1]) if separate else None,
                         None if separate else adr(outs[0]), None, None, None, None,
                         int(math), _lib.HOST, None))
        return outs if separate else outs[0]

    @staticmethod
    def _et_bind(lw_net_day, lw_net_night, sw_rad_day, sw_rad_night, sw_albedo,
                 temp_day, temp_night, temp_annual, tmin, vpd_day, vpd_night,
                 pressure, fpar, lai, observed=None, weights=None, max_draws=4096,
                 math=_lib.MATH_FAST, device=0):
        '''
        (Extension.) The calibration problem made RESIDENT on the GPU: what the
        reference's sampler (calibration.py:907-909) and Sobol analysis
        (sensitivity.py:94-96) do is call ``MOD16._et(params, *drivers)`` thousands
        of times on the same drivers. Returns a ``BoundCalibration`` holding the
        drivers (and ``observed`` / ``weights``) on the device; each
        ``problem.objective(params)`` then moves ``D x 11`` parameters up and
        ``(sse, count)`` down around one graph launch -- nothing of the drivers'
        size crosses PCIe again, nothing of size ``D x n`` exists anywhere.
        ``problem.rows(params)`` gives the ``(D x shape)`` array
        ``MOD16._et_batch(params, *drivers, math=math)`` gives, bit for bit.
        ``math``: ``MATH_FAST`` (default here: the strength-reduced float64
        arithmetic, within 1e-9 of the reference order; pixels outside its domain
        are computed in the reference's order, as in the forward run) or
        ``MATH_EXACT``.
        '''
        drivers = [lw_net_day, lw_net_night, sw_rad_day, sw_rad_night,
                   sw_albedo, temp_day, temp_night, temp_annual, tmin, vpd_day,
                   vpd_night, pressure, fpar, lai]
        return BoundCalibration(drivers, observed, weights, max_draws, math, device)

    @staticmethod
    def air_density(temp_k, pressure, rhumidity):
        'Air density [kg m-3], reference mod16/__init__.py:384-412.'
        return _call_method(_lib.M_AIR_DENSITY, [temp_k, pressure, rhumidity])

    @staticmethod
    def air_pressure(elevation_m):
        'Air pressure [Pa] from elevation [m], reference :414-447.'
        return _call_method(_lib.M_AIR_PRESSURE, [elevation_m])

    @staticmethod
    def potential_soil_evaporation(
            pressure, temp_k, vpd, fpar, rad_soil, r_corr=None, lhv=None,
            rhumidity=None, f_wet=None, vpd_open=None, vpd_close=None,
            rbl_min=None, rbl_max=None):
        '''(evaporation from the saturated fraction, potential evaporation
        from the unsaturated fraction) [W m-2], reference :449-544.'''
        return _call_method(
            _lib.M_POT_SOIL_EVAP,
            [pressure, temp_k, vpd, fpar, rad_soil, r_corr, lhv, rhumidity, f_wet],
            dict(vpd_open=vpd_open, vpd_close=vpd_close, rbl_min=rbl_min,
                 rbl_max=rbl_max), nout=2)

    @staticmethod
    def potential_transpiration(
            lw_net, sw_rad, sw_albedo, pressure, temp_k, vpd, fpar,
            rhumidity=None, f_wet=None, alpha=1.26):
        'Priestley-Taylor potential transpiration [W m-2], reference :546-602.'
        return _call_method(
            _lib.M_POT_TRANSPIRATION,
            [lw_net, sw_rad, sw_albedo, pressure, temp_k, vpd, fpar, rhumidity,
             f_wet], alpha=alpha)

    @staticmethod
    def vpd(qv10m, pressure, tmean):
        'Vapor pressure deficit [Pa] from specific humidity, reference :604-644.'
        return _call_method(_lib.M_VPD, [qv10m, pressure, tmean])

    @staticmethod
    def rhumidity(temp_k, vpd):
        'Relative humidity on [0, 1] from VPD, reference :646-673.'
        return _call_method(_lib.M_RHUMIDITY, [temp_k, vpd])

    def evaporation_soil(
            self, pressure, temp_k, vpd, fpar, rad_soil, r_corr=None, lhv=None,
            rhumidity=None, f_wet=None):
        'Bare-soil evaporation [kg m-2 s-1], reference :795-864.'
        return _call_method(
            _lib.M_EVAP_SOIL,
            [pressure, temp_k, vpd, fpar, rad_soil, r_corr, lhv, rhumidity, f_wet],
            self._p('vpd_open', 'vpd_close', 'rbl_min', 'rbl_max', 'beta'),
            device=self.device)

    def evaporation_wet_canopy(
            self, pressure, temp_k, vpd, lai, fpar, rad_canopy, lhv=None,
            rhumidity=None, f_wet=None, tiny=1e-7):
        'Wet-canopy evaporation [kg m-2 s-1], reference :866-961.'
        return _call_method(
            _lib.M_EVAP_WET_CANOPY,
            [pressure, temp_k, vpd, lai, fpar, rad_canopy, lhv, rhumidity, f_wet],
            self._p('gl_sh', 'gl_wv'), device=self.device, tiny=tiny)

    def radiation_soil(
            self, lw_net_day, lw_net_night, sw_rad_day, sw_rad_night, sw_albedo,
            temp_day, temp_night, temp_annual, fpar):
        'Net radiation received by the soil (day, night) [W m-2], reference :963-1053.'
        return _call_method(
            _lib.M_RADIATION_SOIL,
            [lw_net_day, lw_net_night, sw_rad_day, sw_rad_night, sw_albedo,
             temp_day, temp_night, temp_annual, fpar],
            self._p('tmin_close'), nout=2, device=self.device)

    def soil_heat_flux(
            self, rad_net_day, rad_net_night, temp_day, temp_night, temp_annual):
        'Soil heat flux [day, night] [W m-2], reference :1055-1119.'
        return list(_call_method(
            _lib.M_SOIL_HEAT_FLUX,
            [rad_net_day, rad_net_night, temp_day, temp_night, temp_annual],
            self._p('tmin_close'), nout=2, device=self.device))

    def surface_conductance(self, tmin, vpd_day):
        'Surface conductance [m s-1], reference :1121-1150.'
        return _call_method(
            _lib.M_SURFACE_CONDUCTANCE, [tmin, vpd_day],
            self._p('tmin_close', 'tmin_open', 'vpd_open', 'vpd_close', 'csl'),
            device=self.device)

    def transpiration(
            self, pressure, temp_k, vpd, lai, fpar, rad_canopy, tmin,
            r_corr=None, lhv=None, rhumidity=None, f_wet=None, daytime=True,
            tiny=1e-7):
        'Plant transpiration [kg m-2 s-1], reference :1152-1258.'
        return _call_method(
            _lib.M_TRANSPIRATION_DAY if daytime else _lib.M_TRANSPIRATION_NIGHT,
            [pressure, temp_k, vpd, lai, fpar, rad_canopy, tmin, r_corr, lhv,
             rhumidity, f_wet],
            self._p('tmin_close', 'tmin_open', 'vpd_open', 'vpd_close', 'gl_sh',
                    'g_cuticular', 'csl'), device=self.device, tiny=tiny)


class BoundCalibration(object):
    '''A calibration problem resident on the GPU (``MOD16._et_bind``;
    ``mod16_static_batch_bind_*`` of the C ABI).'''

    def __init__(self, drivers, observed, weights, max_draws, math, device):
        import ctypes as C
        if weights is not None and observed is None:
            raise ValueError('weights need observed')
        extra = [v for v in (observed, weights) if v is not None]
        self.dtype = _result_dtype(list(drivers) + extra)
        self.shape = np.broadcast_shapes(*[np.shape(v) for v in list(drivers) + extra])
        self.n = int(np.prod(self.shape, dtype=np.int64))
        if self.n == 0:
            raise ValueError('a calibration problem needs at least one pixel')
        self.max_draws = int(max_draws)
        self.math = int(math)
        self._ctx = _lib.context(device)
        keep, dptr, dstr = _marshal(drivers, self.shape, self.dtype)
        full = lambda v: np.ascontiguousarray(np.broadcast_to(np.asarray(v, self.dtype), self.shape))
        obs = full(observed) if observed is not None else None
        wts = full(weights) if weights is not None else None
        self.has_observed = obs is not None
        fn = self._ctx.lib.mod16_static_batch_bind_f32 if self.dtype == np.float32 \
            else self._ctx.lib.mod16_static_batch_bind_f64
        self._handle = C.c_void_p()
        self._ctx.check(fn(self._ctx.handle, _lib.ptr_array(dptr), _lib.i64_array(dstr), self.n,
                           obs.ctypes.data if obs is not None else None,
                           wts.ctypes.data if wts is not None else None,
                           self.max_draws, self.math, _lib.HOST, C.byref(self._handle)))
        n_out = C.c_int64(0)
        self._ctx.check(self._ctx.lib.mod16_static_batch_info(self._handle, None, None, C.byref(n_out)))
        #: pixels outside the domain of the FAST arithmetic (computed in the reference's order)
        self.n_outside_domain = n_out.value

    def _params(self, params):
        par = np.ascontiguousarray(params, self.dtype)
        if par.ndim != 2 or par.shape[1] != 11:
            raise IndexError('params must be (D x 11), columns in MOD16.required_parameters order')
        if par.shape[0] > self.max_draws:
            raise ValueError('%d parameter vectors, the problem was bound for max_draws = %d'
                             % (par.shape[0], self.max_draws))
        return par

    def objective(self, params):
        '''``(sse, count)``: two float64 arrays (D,) with ``sse[d] = sum((weights *
        (_et_d - observed))**2)`` over the non-NaN pairs and their number, as
        ``MOD16._et_batch(params, *drivers, observed=..., weights=...)``; e.g.
        ``rmsd = np.sqrt(sse / count)``.'''
        if not self.has_observed:
            raise ValueError('the problem was bound without observed')
        par = self._params(params)
        sse, count = np.zeros(par.shape[0]), np.zeros(par.shape[0])
        if par.shape[0]:
            self._ctx.check(self._ctx.lib.mod16_static_batch_objective(
                self._handle, par.ctypes.data, par.shape[0], sse.ctypes.data, count.ctypes.data))
        return sse, count

    def rows(self, params, separate=False):
        '''The (D x shape) array of ``MOD16._et`` per parameter vector (W m-2), or
        with ``separate`` ``[day, night]`` -- what ``MOD16._et_batch`` returns.'''
        par = self._params(params)
        outs = [np.empty((par.shape[0],) + self.shape, self.dtype) for _ in range(2 if separate else 1)]
        if par.shape[0]:
            adr = lambda a: a.ctypes.data
            self._ctx.check(self._ctx.lib.mod16_static_batch_rows(
                self._handle, par.ctypes.data, par.shape[0],
                adr(outs[0]) if separate else None, adr(outs[1]) if separate else None,
                None if separate else adr(outs[0])))
        return outs if separate else outs[0]

    def gpu_milliseconds(self, launches=10):
        '''Mean GPU time of one objective evaluation of the last shape (HIP events
        around graph replays); FAST problems only, after a first ``objective``.'''
        import ctypes as C
        ms = C.c_float(0)
        self._ctx.check(self._ctx.lib.mod16_static_batch_time(self._handle, int(launches), C.byref(ms)))
        return ms.value

    def close(self):
        if getattr(self, '_handle', None) is not None and self._handle.value:
            self._ctx.lib.mod16_static_batch_destroy(self._handle)
            self._handle.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def latent_heat_vaporization(temp_k):
    'Latent heat of vaporization [J kg-1], reference mod16/__init__.py:121.'
    return _call_method(_lib.M_LHV, [temp_k])


def psychrometric_constant(pressure, temp_k):
    'Psychrometric constant [Pa K-1], reference :1261-1290.'
    return _call_method(_lib.M_PSYCHROMETRIC, [pressure, temp_k])


def radiation_net(sw_rad, sw_albedo, temp_k):
    'DEPRECATED in the reference (:1293-1337); net radiation [W m-2].'
    return _call_method(_lib.M_RADIATION_NET, [sw_rad, sw_albedo, temp_k])


def svp(temp_k):
    'Saturation vapor pressure [Pa], reference :1340-1367.'
    return _call_method(_lib.M_SVP, [temp_k])


def svp_slope(temp_k, s=None):
    'Slope of the saturation vapor pressure curve [Pa K-1], reference :1370-1397.'
    return _call_method(_lib.M_SVP_SLOPE, [temp_k, s])


def evapotranspiration_raster(
        bplut, cls, lw_net_day, lw_net_night, sw_rad_day, sw_rad_night,
        sw_albedo, temp_day, temp_night, temp_annual, tmin, vpd_day,
        vpd_night, pressure, fpar, lai, separate=False, beta=None,
        math=_lib.MATH_FAST, device=0, pet=False, out=None, devices=None,
        diagnostics=False):
    r'''
    Forward run over a multi-class raster. Equivalent to the reference idiom
    (forward-run notebook, cell 32)::

        params = {k: bplut[k][cls] for k in MOD16.required_parameters}
        MOD16(params).evapotranspiration(*drivers)

    but the 11 per-pixel parameter arrays are never materialised: the BPLUT
    sits in LDS and is indexed by the class raster inside the kernel.

    Parameters
    ----------
    bplut : dict or numpy.ndarray
        What ``restore_bplut`` returns (11 arrays of 13), or a (13, 11) table
        in ``MOD16.required_parameters`` column order
    cls : numpy.ndarray
        Land-cover class (PFT code) raster, integers in [0, 12]. Codes that
        are not PFTs (0, 11) give NaN, as NaN parameters do in the reference;
        a code >= 13 raises IndexError, as the numpy gather would.
    beta : float
        (Optional) value for the ``beta`` column where the table has none
    separate : bool
        As in ``MOD16.evapotranspiration``
    pet : bool
        (Extension) True to return ``(day, night, pet_day, pet_night)``, see
        ``MOD16.evapotranspiration_and_pet``
    out : sequence of numpy.ndarray
        (Extension) the 2 (or, with ``separate``, 6) output arrays to write
        into instead of allocating them, e.g. memory-mapped files
        (``mod16_amd.io``)
    devices : sequence of int
        (Extension) several GPUs behind this one call: the flattened raster is
        cut at the boundaries of the HOST mode's staging tiles and dealt over
        the listed devices in order, one host thread, context and PCIe link per
        entry (``mod16_amd.multi``; SURVEY.md 8e without a collective). Outputs
        and diagnostics are bit-identical to ``devices=[0]`` whatever the list;
        a class code >= 13 anywhere still raises IndexError.
    diagnostics : bool
        (Extension) True to return ``(day, night, diag)``: ``diag`` is the
        float64 vector of ``mod16_amd.raster.DIAG_FIELDS`` (sums, counts of
        finite / NaN pixels, maxima), reduced on the GPU tile by tile while the
        results are there and folded in tile order

    Returns
    -------
    tuple
        As ``MOD16.evapotranspiration``
    '''
    from .utils import bplut_table
    if isinstance(bplut, dict):
        table = bplut_table(bplut, beta=beta)
    else:
        table = np.array(bplut, np.float64)
        if beta is not None:
            fill = np.isnan(table[:, 10]) & ~np.isnan(table[:, 0])
            table[fill, 10] = beta
    drivers = (
        lw_net_day, lw_net_night, sw_rad_day, sw_rad_night, sw_albedo,
        temp_day, temp_night, temp_annual, tmin, vpd_day, vpd_night,
        pressure, fpar, lai)
    return _forward(cls, drivers, None, separate, math, device, pet=pet, out=out,
                    devices=devices, table=table, diagnostics=diagnostics)


def evapotranspiration_raw(
        bplut, cls, lw_net_day, lw_net_night, sw_rad_day, sw_rad_night,
        sw_albedo, temp_day, temp_night, temp_annual, tmin, qv10m_day,
        qv10m_night, ps_day, ps_night, elevation, fpar_pct, lai_x10,
        day_hours=None, beta=None, math=_lib.MATH_FAST, device=0, devices=None):
    r'''
    (Extension) Forward run on raw drivers: the
    pre-processing the reference does in front of ``evapotranspiration()``
    (mod16/calibration.py:380-423) is folded into the kernel --

    - ``vpd_day = MOD16.vpd(qv10m_day, ps_day, temp_day)``,
      ``vpd_night = max(MOD16.vpd(qv10m_night, ps_night, temp_night), 0)``;
    - ``pressure = MOD16.air_pressure(elevation)``;
    - ``fpar = fpar_pct / 100``, ``lai = lai_x10 / 10`` with ``fpar_pct`` and
      ``lai_x10`` as uint8 rasters in the MODIS encodings (codes >= 249 are
      fill values and give NaN).

    Returns ``(day, night)`` [kg m-2 s-1] or, with ``day_hours`` (hours of
    daylight), ``(day, night, total8)`` where ``total8 = (day h + night (24 -
    h)) * 8 * 3600`` [kg m-2 (8 d)-1], the MOD16A2 unit
    (tests/verification/verify2.py:113-115). ``devices``: several GPUs behind the
    one call, as for ``evapotranspiration_raster``.
    '''
    from . import multi
    from .utils import bplut_table
    table = bplut_table(bplut, beta=beta) if isinstance(bplut, dict) else np.array(bplut, np.float64)
    if not isinstance(bplut, dict) and beta is not None:
        fill = np.isnan(table[:, 10]) & ~np.isnan(table[:, 0])
        table[fill, 10] = beta
    devs = multi.device_list(devices)
    raw = [lw_net_day, lw_net_night, sw_rad_day, sw_rad_night, sw_albedo,
           temp_day, temp_night, temp_annual, tmin, qv10m_day, qv10m_night,
           ps_day, ps_night, elevation]
    hours = [day_hours] if day_hours is not None else []
    dtype = _result_dtype(raw + hours)
    u8 = [np.asarray(v) for v in (cls, fpar_pct, lai_x10)]
    shape = np.broadcast_shapes(*[np.shape(v) for v in raw + hours + u8])
    n = int(np.prod(shape, dtype=np.int64))
    keep, rptr, rstr = _marshal(raw, shape, dtype)
    hkeep, hptr, hstr = _marshal(hours, shape, dtype) if hours else (None, [None], [0])
    bytes_ = []
    for a in u8:
        if a.dtype != np.uint8:
            if a.size and (a.min() < 0 or a.max() > 255):
                raise IndexError('uint8 raster value outside [0, 255]')
            a = a.astype(np.uint8)
        bytes_.append(np.ascontiguousarray(np.broadcast_to(a, shape)))
    outs = [_lib.pinned.empty(shape, dtype) for _ in range(3 if hours else 2)]
    esz = dtype.itemsize

    def part(ctx, off, m):
        ctx.set_bplut(table)
        if m <= 0:
            return
        fn = ctx.lib.mod16_et_raw_f32 if dtype == np.float32 else ctx.lib.mod16_et_raw_f64
        ctx.check(fn(
            ctx.handle, bytes_[0].ctypes.data + off, _lib.ptr_array(_shift(rptr, rstr, off, 1, esz)),
            _lib.i64_array(rstr), bytes_[1].ctypes.data + off, bytes_[2].ctypes.data + off,
            _shift(hptr, hstr, off, 1, esz)[0], int(hstr[0]), m,
            outs[0].ctypes.data + off * esz, outs[1].ctypes.data + off * esz,
            outs[2].ctypes.data + off * esz if hours else None, int(math), _lib.HOST, None))

    if devs is None:
        part(_lib.context(device), 0, n)
    else:
        cuts = multi.shards(n, len(devs), multi.host_tile())
        multi.run(devs, lambda i, ctx: part(ctx, *cuts[i]))
    if not shape:
        outs = [o[()] for o in outs]
    return tuple(outs)
