'''
Device-resident forward run for large rasters.

``RasterEngine`` is the zero-copy, asynchronous form of
``mod16_amd.evapotranspiration_raster``: drivers, class raster and outputs are
``torch`` tensors already in HBM, the kernel is enqueued on the current HIP
stream through the C ABI (``mod16_et_*`` with ``where = MOD16_DEVICE``) and
nothing is copied. PyTorch is used for device memory and streams only.
'''
import ctypes as C

import numpy as np

from . import _lib

#: algorithmic HBM bytes per pixel: 14 driver loads +
#: 1 class byte + 2 stores
BYTES_PER_PIXEL = {'float64': 14 * 8 + 1 + 2 * 8, 'float32': 14 * 4 + 1 + 2 * 4}
DIAG_FIELDS = ('sum_day', 'sum_night', 'n_valid_day', 'n_valid_night',
               'n_nan_day', 'n_nan_night', 'max_day', 'max_night')


def _torch():
    import torch
    return torch


class _GraphHandle(object):
    '''Owns a ``mod16_graph`` (destroyed with the bound launch that uses it).'''

    def __init__(self, lib, handle, ctx=None):
        self.lib, self.handle = lib, handle
        self.ctx = ctx          # keeps the context alive as long as its graph

    def __del__(self):
        try:
            if self.handle and self.handle.value:
                self.lib.mod16_graph_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class TiledRaster(object):
    '''
    A raster resident on the device in the engine's own layout (``mod16_layout``
    in include/mod16_hip.h): the pixels are cut into tiles of ``tile`` pixels and
    the 14 driver arrays interleaved tile by tile -- ``[tile][driver][tile
    pixels]`` -- the two outputs likewise in a block of their own, the class
    raster as plain bytes. ``form`` (``_lib.FORM_*``, ``enum mod16_form``) selects
    another form of the forward run -- potential ET, components, raw drivers -- with
    its own counts of arrays: ``wide`` (the drivers), ``bytes`` (class raster first,
    then the uint8 fPAR / LAI of the raw forms) and ``outs``, in the order of
    include/mod16_hip.h. Every array is still an array: ``drivers[k]``, ``day``,
    ``night`` and ``cls`` are 2-D strided ``torch`` views of shape ``(ntiles,
    tile)`` into one allocation, so ``drivers[5].copy_(temp.view(-1, tile))``,
    slicing and reductions work as on any tensor. Why: streamed side by side, 16
    separate 7 GB arrays lie GiB apart in HBM and the kernel's 14-read + 2-write
    mix reaches 5.7 TB/s; with the fields of a tile inside one ~1 MiB block the
    same bytes move at 6.5 TB/s (``tools/probe_layout.hip``).
    The storage is padded to whole tiles; ``n`` is the number of real pixels.
    '''

    def __init__(self, engine, n, tile=None, form=_lib.FORM_TOTALS):
        torch = _torch()
        esz = engine.np_dtype.itemsize
        self.n = int(n)
        self.form = int(form)
        nw, nb, no = _lib.FORM_SHAPE[self.form]
        self.tile = int(tile) if tile else engine.TILE_BYTES // esz
        if self.tile & (self.tile - 1) or self.tile * esz < 8192:
            raise ValueError('tile must be a power of two of at least 8 KiB per field')
        vec = 16 // esz
        if self.n % vec:
            raise ValueError('a tiled raster holds a multiple of %d pixels' % vec)
        self.ntiles = max(1, -(-self.n // self.tile))
        P, nt = self.tile, self.ntiles
        in_bytes, out_bytes, byte_bytes = nt * nw * P * esz, nt * no * P * esz, nt * nb * P
        self.slab = torch.empty(in_bytes + out_bytes + byte_bytes + 4096, dtype=torch.uint8,
                                device=engine._dev())
        wide = self.slab[:in_bytes].view(engine.dtype).view(nt, nw, P)
        self.wide = [wide[:, k, :] for k in range(nw)]
        outs = self.slab[in_bytes:in_bytes + out_bytes].view(engine.dtype).view(nt, no, P)
        self.outs = [outs[:, k, :] for k in range(no)]
        rasters = self.slab[in_bytes + out_bytes:in_bytes + out_bytes + byte_bytes].view(nt, nb, P)
        self.bytes = [rasters[:, k, :] for k in range(nb)]
        # the names of the totals form (the production step)
        self.drivers = self.wide[:14]
        self.cls = self.bytes[0]
        totals = self.form != _lib.FORM_COMPONENTS
        self.day, self.night = (self.outs[0], self.outs[1]) if totals else (None, None)
        self.layout = _lib.Layout(P, nw * P, no * P, nb * P)
        self.dtype = engine.dtype

    def flat(self, field, lo=0, hi=None):
        '''Pixels [lo, hi) of one array of the raster as a contiguous 1-D
        tensor -- always a COPY (only the tiles it touches are read): inside one
        tile the strided view is contiguous and reshape would hand back an alias
        of live raster memory, which the next launch overwrites.'''
        hi = self.n if hi is None else hi
        P = self.tile
        t0, t1 = lo // P, -(-hi // P)
        out = field[t0:t1].reshape(-1)[lo - t0 * P:hi - t0 * P]
        if out.data_ptr() == field[t0:t1].data_ptr() + (lo - t0 * P) * field.element_size():
            out = out.clone()
        return out

    def put(self, field, src, lo=0):
        '''Write the 1-D tensor ``src`` into pixels [lo, lo + len(src)) of one
        array of the raster.'''
        P = self.tile
        pos, end = lo, lo + src.numel()
        while pos < end:                       # head, whole tiles at once, tail
            t, w = divmod(pos, P)
            if w == 0 and end - pos >= P:
                k = (end - pos) // P
                field[t:t + k].copy_(src[pos - lo:pos - lo + k * P].view(k, P))
                pos += k * P
            else:
                m = min(P - w, end - pos)
                field[t, w:w + m].copy_(src[pos - lo:pos - lo + m])
                pos += m


class BoundStep(object):
    '''A pre-marshalled step (see ``RasterEngine.bind`` / ``bind_tiled``):
    calling it enqueues the step on the current stream.'''

    def __init__(self, launch, outputs, graph=None, device=0):
        self._launch, self.outputs, self._graph, self._device = launch, outputs, graph, device

    def __call__(self):
        self._launch()
        return self.outputs

    def time(self, launches=10):
        '''Mean milliseconds per replay of the captured step, HIP events on
        the current stream (``mod16_time_graph``).'''
        if self._graph is None:
            raise RuntimeError('only a step bound as a HIP graph can be timed this way')
        torch = _torch()
        ms = C.c_float(0)
        rc = self._graph.lib.mod16_time_graph(
            self._graph.handle, int(launches),
            C.c_void_p(torch.cuda.current_stream(self._device).cuda_stream), C.byref(ms))
        self._graph.ctx.check(rc)
        return ms.value


class RasterEngine(object):
    '''
    Parameters
    ----------
    table : numpy.ndarray
        (13, 11) BPLUT table (``mod16_amd.utils.bplut_table``)
    device : int
        GPU index (Default: the current torch device)
    dtype : str
        'float64' (default) or 'float32'
    math : int
        ``_lib.MATH_FAST`` (default), ``_lib.MATH_EXACT`` or, for float32
        rasters, ``_lib.MATH_MIXED`` (see ``include/mod16_hip.h``)
    trusted : bool
        ``MOD16_DOMAIN_TRUSTED``: the caller vouches that every driver lies inside
        the domain of the production arithmetic (quality-controlled or NaN-masked
        rasters; the bounds are in ``include/mod16_hip.h``) -- the totals form then
        runs without the domain test and without the dispatch that revisits flagged
        pixels (-1.6 % kernel time on the float64 global grid, -2.3 % MIXED: tools/guardcost.py, round 5).
        Default: False, the
        arithmetic that returns the reference's result on every input.
    '''

    def __init__(self, table, device=None, dtype='float64', math=_lib.MATH_FAST, trusted=False,
                 experiments=False):
        torch = _torch()
        if not torch.cuda.is_available():
            raise _lib.Mod16Error(
                _lib.ERR_NO_DEVICE, 'RasterEngine needs an MI355X; there is no CPU fallback')
        self.device = torch.cuda.current_device() if device is None else int(device)
        # a context of its own: the BPLUT set here is what this engine's launches
        # (and the graphs bound from it) read at run time, whatever table other
        # engines or the numpy entry points of the process use meanwhile
        # (experiments: a context of libmod16hip_exp.so, which reads the launch-geometry overrides
        # from the environment -- tests and tools only)
        self.ctx = _lib.Context(self.device, experiments=experiments)
        self.ctx.set_bplut(table)
        self.np_dtype = np.dtype(dtype)
        self.dtype = {'float64': torch.float64, 'float32': torch.float32}[self.np_dtype.name]
        self.math = int(math) | (_lib.DOMAIN_TRUSTED if trusted else 0)
        self.bytes_per_pixel = BYTES_PER_PIXEL[self.np_dtype.name]

    def _gate(self):
        '''A zero-work dispatch on the compute stream in front of a step of the series runners.
        Measured (round 4, 46-step series, generator on the second stream): with the persistent
        pipeline kernel enqueued directly behind its cross-stream event wait the step took 45 ms --
        the two kernels got in each other's way -- with one tiny dispatch between the wait and the
        kernel 38.5 ms, the sum of the two kernels alone (19.4 + 19.9 ms: both are bound by the same
        HBM). Round 3's launch sequence had such a dispatch by accident (the memset of the ticket
        counter, gone since the kernel resets it itself). Round 5, wall clock and kernel trace of both
        (``tools/gate_probe.py``, ``profiles/r05_gate_probe.json``): 39.4 ms per step with the gate = the sum
        of the kernels alone (39.2), 46.3 without; ``bench.py`` reports the ratio of every run
        (``configs.c4_series_float64.step_over_sum_of_kernels``, 1.004).'''
        torch = _torch()
        if getattr(self, '_gate_word', None) is None:
            self._gate_word = torch.zeros(64, dtype=torch.float32, device=self._dev())
        self._gate_word.zero_()

    #: bytes of one field per tile of a ``TiledRaster`` (16-64 KiB measured: 32 KiB best)
    TILE_BYTES = 32 * 1024

    # ---------------------------------------------------------- helpers
    def _dev(self):
        return _torch().device('cuda', self.device)

    def _stream(self):
        return C.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    def _check_tensor(self, t, dtype, n, what):
        torch = _torch()
        if not isinstance(t, torch.Tensor) or not t.is_cuda or t.device.index != self.device:
            raise TypeError('%s must be a tensor on cuda:%d' % (what, self.device))
        if t.dtype != dtype or not t.is_contiguous():
            raise TypeError('%s must be contiguous %s' % (what, dtype))
        if t.numel() != n:
            raise ValueError('%s has %d elements, expected %d' % (what, t.numel(), n))
        return t.data_ptr()

    #: extra bytes between successive arrays of ``alloc_raster``: with arrays
    #: spaced by (a multiple of 4 KiB) + 0 or + 4 KiB the 16 concurrent streams
    #: collide in HBM (+6-9 % kernel time, measured); any offset >= 8 KiB
    #: is fine, 33 KiB measured best
    STAGGER_BYTES = 33 * 1024

    def _carve(self, slab, n, per):
        esz = self.np_dtype.itemsize
        views = [slab[k * per:k * per + n * esz].view(self.dtype) for k in range(16)]
        cls = slab[16 * per:16 * per + n]
        return cls, views[:14], views[14], views[15]

    def _pitch(self, n, extra):
        return (n * self.np_dtype.itemsize + 4095) // 4096 * 4096 + self.STAGGER_BYTES + int(extra)

    def alloc_raster(self, n, extra=0):
        '''Device arrays for one raster of n pixels in the layout the fused
        kernel streams best: class raster, 14 drivers and the two outputs are
        carved out of ONE allocation, successive arrays ``STAGGER_BYTES`` +
        ``extra`` bytes apart on top of their size (see ``alloc_raster_tuned``
        for ``extra``), so their relative placement in HBM does not depend on
        the allocator. Returns ``(cls, drivers, day, night)``.'''
        torch = _torch()
        per = self._pitch(n, extra)
        slab = torch.empty(16 * per + n + 4096, dtype=torch.uint8, device=self._dev())
        return self._carve(slab, n, per)

    #: candidate extra spacings between the arrays of a raster [bytes] that
    #: ``alloc_raster_tuned`` looks at
    TUNE_EXTRA = tuple(int(g * 2 ** 30) for g in (0, 0.125, 0.25, 0.375, 0.5, 0.625, 0.75))

    def alloc_raster_tuned(self, n, extras=None, launches=3, seed=16):
        '''``alloc_raster`` with the distance between the arrays chosen by
        measurement. The 16 streams of the fused kernel walk their arrays in
        lock step, and how far apart the arrays lie -- at the scale of
        hundreds of MiB -- changes the DRAM-side read latency and with it the
        kernel time by 5-9 % (43200 x 21600 float64: 21.9-22.5 ms with the
        arrays back to back, 21.0-21.2 ms with 0.5 GiB between them;
        DESIGN.md section 6, ``tools/layout_sweep.py``). Which spacing is best
        depends on the raster size and somewhat on where the allocation
        landed, so for a raster that stays resident this allocates one slab
        large enough for every candidate in ``extras`` (bytes, default
        ``TUNE_EXTRA``), times the production kernel on synthetic drivers
        for each and keeps the fastest -- a few launches per candidate,
        once, at set-up.

        Returns ``((cls, drivers, day, night), report)``; ``report`` maps each
        extra spacing to its milliseconds per launch and names the choice.
        The arrays hold the synthetic fields of ``seed``, to be overwritten.'''
        torch = _torch()
        extras = [int(e) for e in (self.TUNE_EXTRA if extras is None else extras)]
        per_max = self._pitch(n, max(extras))
        slab = torch.empty(16 * per_max + n + 4096, dtype=torch.uint8, device=self._dev())
        times = {}
        for e in extras:
            cand = self._carve(slab, n, self._pitch(n, e))
            self.synth(n, seed=seed, out=(cand[0], cand[1]))
            self.time_kernel(cand[0], cand[1], cand[2], cand[3], launches=1)
            times[e] = self.time_kernel(cand[0], cand[1], cand[2], cand[3], launches=launches)
        best = min(times, key=times.get)
        report = {'extra_bytes_ms': {str(e): round(ms, 4) for e, ms in times.items()},
                  'chosen_extra_bytes': best}
        return self._carve(slab, n, self._pitch(n, best)), report

    def empty(self, n, count=1):
        torch = _torch()
        return [torch.empty(n, dtype=self.dtype, device=self._dev()) for _ in range(count)]

    def _marshal_drivers(self, drivers, n):
        torch = _torch()
        if len(drivers) != _lib.N_DRIVERS:
            raise ValueError('expected 14 drivers')
        keep, ptrs, strides = [], [], []
        for k, d in enumerate(drivers):
            if isinstance(d, torch.Tensor) and d.numel() != 1:
                ptrs.append(self._check_tensor(d, self.dtype, n, 'driver %d' % k))
                strides.append(1)
                keep.append(d)
            else:       # broadcast scalar
                s = torch.as_tensor(d, dtype=self.dtype).reshape(1).to(self._dev())
                keep.append(s)
                ptrs.append(s.data_ptr())
                strides.append(0)
        return keep, ptrs, strides

    # -------------------------------------------------------------- API
    def synth(self, n, seed=16, step=0, pixel_offset=0, out=None):
        '''Fill (or allocate) the class raster and the 14 driver tensors for
        global pixels [pixel_offset, pixel_offset + n) on the device
        (``mod16_synth_*``). Returns ``(cls, drivers)``.'''
        torch = _torch()
        if out is None:
            cls = torch.empty(n, dtype=torch.uint8, device=self._dev())
            drivers = self.empty(n, _lib.N_DRIVERS)
        else:
            cls, drivers = out
        fn = self.ctx.lib.mod16_synth_f32 if self.np_dtype == np.float32 \
            else self.ctx.lib.mod16_synth_f64
        ptrs = [self._check_tensor(d, self.dtype, n, 'driver') for d in drivers]
        self.ctx.check(fn(
            self.ctx.handle, int(seed), int(step), int(pixel_offset), int(n),
            self._check_tensor(cls, torch.uint8, n, 'cls'),
            _lib.ptr_array(ptrs), self._stream()))
        return cls, drivers

    def run(self, cls, drivers, out_day=None, out_night=None, out_sep=None, diag=None):
        '''Enqueue the fused ET kernel on the current stream; returns
        ``(day, night)`` tensors (allocated unless given). With ``diag`` (a
        float64 tensor of 8 on the device) the diagnostics vector
        (``DIAG_FIELDS``) is produced in the same pass (``mod16_et_diag_*``).
        Asynchronous: call ``check()`` (or synchronise the stream) before
        trusting the data.'''
        torch = _torch()
        n = cls.numel()
        cptr = self._check_tensor(cls, torch.uint8, n, 'cls')
        keep, dptr, dstride = self._marshal_drivers(drivers, n)
        if out_day is None and out_sep is None:
            out_day, out_night = self.empty(n, 2)
        pd = self._check_tensor(out_day, self.dtype, n, 'out_day') if out_day is not None else None
        pn = self._check_tensor(out_night, self.dtype, n, 'out_night') if out_night is not None else None
        if diag is not None:
            if out_sep is not None:
                raise ValueError('diag and out_sep cannot be combined')
            fn = self.ctx.lib.mod16_et_diag_f32 if self.np_dtype == np.float32 \
                else self.ctx.lib.mod16_et_diag_f64
            self.ctx.check(fn(
                self.ctx.handle, cptr, _lib.ptr_array(dptr), _lib.i64_array(dstride), n,
                pd, pn, int(self.math),
                self._check_tensor(diag, torch.float64, 8, 'diag'), self._stream()))
            return out_day, out_night
        sep = None
        if out_sep is not None:
            sep = [self._check_tensor(t, self.dtype, n, 'out_sep') if t is not None else None
                   for t in out_sep]
        self.ctx.et(self.np_dtype, cptr, dptr, dstride, None, None, n, pd, pn, sep,
                    flags=self.math, where=_lib.DEVICE, stream=self._stream())
        return out_day, out_night

    def run_pet(self, cls, drivers, out=None):
        '''ET and potential ET from one pass over device-resident drivers
        (``mod16_et_pet_*``, see ``mod16_amd.evapotranspiration_raster(...,
        pet=True)``). Returns ``(day, night, pet_day, pet_night)``; ``out`` may
        give the four output tensors.'''
        torch = _torch()
        n = cls.numel()
        cptr = self._check_tensor(cls, torch.uint8, n, 'cls')
        keep, dptr, dstride = self._marshal_drivers(drivers, n)
        out = tuple(out) if out is not None else tuple(self.empty(n, 4))
        optr = [self._check_tensor(t, self.dtype, n, 'out') for t in out]
        fn = self.ctx.lib.mod16_et_pet_f32 if self.np_dtype == np.float32 \
            else self.ctx.lib.mod16_et_pet_f64
        self.ctx.check(fn(
            self.ctx.handle, cptr, _lib.ptr_array(dptr), _lib.i64_array(dstride), None, None,
            n, optr[0], optr[1], optr[2], optr[3], int(self.math), _lib.DEVICE, self._stream()))
        return out

    def alloc_series(self, n, extra=0):
        '''Device buffers of ``run_series``: class raster, the two-slot driver
        ring and two output pairs (``extra``: see ``alloc_raster``).'''
        torch = _torch()
        a, b = self.alloc_raster(n, extra), self.alloc_raster(n, extra)
        return {'cls': a[0], 'ring': [a[1], b[1]], 'outs': [[a[2], a[3]], [b[2], b[3]]]}

    def run_series(self, n, steps, seed=16, pixel_offset=0, on_step=None, buffers=None):
        '''A time series streamed through HBM (BASELINE.json configs[3]): the
        drivers of step s + 1 are produced on a second HIP stream into the
        other half of a two-slot ring while the fused kernel works on step s.
        Here the producer is the on-device generator (``mod16_synth_*``), the
        stand-in for an ingest stage; the ring, the two streams and the event
        hand-off are what a real ingest would use. Returns ``(diag, day,
        night)``: the [steps, 8] diagnostics series and the outputs of the
        last step. ``on_step(s, day, night)`` is called (compute stream
        current) after step s has been enqueued, e.g. to accumulate; what it
        enqueues on the compute stream may read the outputs and the ring slot
        of step s (``buffers['ring'][s % 2]``), which is handed back to the
        ingest stream only behind it.'''
        torch = _torch()
        dev = self._dev()
        compute = torch.cuda.current_stream(self.device)
        ingest = torch.cuda.Stream(device=dev)
        buffers = buffers or self.alloc_series(n)
        cls, ring, outs = buffers['cls'], buffers['ring'], buffers['outs']
        diag = torch.zeros(steps, 8, dtype=torch.float64, device=dev)
        filled = [torch.cuda.Event() for _ in range(steps)]
        consumed = [torch.cuda.Event() for _ in range(steps)]
        ingest.wait_stream(compute)

        def produce(s):
            with torch.cuda.stream(ingest):
                if s >= 2:
                    ingest.wait_event(consumed[s - 2])      # slot is free again
                self.synth(n, seed=seed, step=s, pixel_offset=pixel_offset,
                           out=(cls, ring[s % 2]))
                filled[s].record(ingest)

        for s in range(min(2, steps)):
            produce(s)
        day = night = None
        for s in range(steps):
            compute.wait_event(filled[s])
            self._gate()
            day, night = outs[s % 2]
            self.run(cls, ring[s % 2], day, night, diag=diag[s])
            if on_step is not None:
                on_step(s, day, night)
            consumed[s].record(compute)     # behind on_step: it may still read the slot
            if s + 2 < steps:
                produce(s + 2)
        compute.wait_stream(ingest)
        return diag, day, night

    def run_raw(self, cls, raw, fpar_pct, lai_x10, day_hours=None, out_day=None,
                out_night=None, out_total8=None):
        '''Forward run on raw drivers held on the device (``mod16_et_raw_*``,
        see ``mod16_amd.evapotranspiration_raw``): ``raw`` = 14 tensors in
        ``mod16_raw_driver`` order, ``fpar_pct`` / ``lai_x10`` uint8 tensors.
        Returns ``(day, night)`` or, with ``day_hours``, ``(day, night,
        total8)``; outputs are allocated unless given.'''
        torch = _torch()
        n = cls.numel()
        keep, rptr, rstr = self._marshal_drivers(raw, n)
        hptr, hstr = None, 0
        if day_hours is not None:
            if isinstance(day_hours, torch.Tensor) and day_hours.numel() != 1:
                hptr, hstr = self._check_tensor(day_hours, self.dtype, n, 'day_hours'), 1
            else:
                hkeep = torch.as_tensor(day_hours, dtype=self.dtype).reshape(1).to(self._dev())
                keep.append(hkeep)
                hptr = hkeep.data_ptr()
        if out_day is None and out_night is None and out_total8 is None:
            out_day, out_night = self.empty(n, 2)
            if day_hours is not None:
                out_total8 = self.empty(n, 1)[0]
        ptr = lambda t, what: self._check_tensor(t, self.dtype, n, what) if t is not None else None
        fn = self.ctx.lib.mod16_et_raw_f32 if self.np_dtype == np.float32 \
            else self.ctx.lib.mod16_et_raw_f64
        self.ctx.check(fn(
            self.ctx.handle, self._check_tensor(cls, torch.uint8, n, 'cls'),
            _lib.ptr_array(rptr), _lib.i64_array(rstr),
            self._check_tensor(fpar_pct, torch.uint8, n, 'fpar_pct'),
            self._check_tensor(lai_x10, torch.uint8, n, 'lai_x10'), hptr, hstr, n,
            ptr(out_day, 'out_day'), ptr(out_night, 'out_night'), ptr(out_total8, 'out_total8'),
            int(self.math), _lib.DEVICE, self._stream()))
        if day_hours is not None:
            return out_day, out_night, out_total8
        return out_day, out_night

    def bind(self, cls, drivers, out_day, out_night, diag, graph=True):
        '''Pre-marshal one ``run(..., diag=diag)`` call and return a function
        that enqueues it on the then-current stream with a single library call
        (the per-step host cost of a time loop: no tensor checks, no ctypes
        array construction). With ``graph`` the launch sequence (counter
        reset, pipeline kernel, staged sum of the diagnostics) is captured
        into a HIP graph once (``mod16_graph_et_diag_*``) and a step is one
        ``hipGraphLaunch``. The tensors must stay alive and in place; their
        contents may change between steps.'''
        torch = _torch()
        n = cls.numel()
        cptr = self._check_tensor(cls, torch.uint8, n, 'cls')
        keep, dptr, dstride = self._marshal_drivers(drivers, n)
        args = (self.ctx.handle, cptr, _lib.ptr_array(dptr), _lib.i64_array(dstride), n,
                self._check_tensor(out_day, self.dtype, n, 'out_day'),
                self._check_tensor(out_night, self.dtype, n, 'out_night'), int(self.math),
                self._check_tensor(diag, torch.float64, 8, 'diag'))
        f32 = self.np_dtype == np.float32
        check, device, lib = self.ctx.check, self.device, self.ctx.lib
        keepalive = (keep, cls, out_day, out_night, diag)
        if graph:
            handle = C.c_void_p()
            make = lib.mod16_graph_et_diag_f32 if f32 else lib.mod16_graph_et_diag_f64
            rc = make(*args, C.byref(handle))
            if rc == _lib.OK:
                owner = _GraphHandle(lib, handle, self.ctx)

                def launch():
                    check(lib.mod16_graph_launch(owner.handle, torch.cuda.current_stream(device).cuda_stream))
                step = BoundStep(launch, (keepalive[2], keepalive[3]), owner, device)
                step._keep = keepalive
                return step
            if rc != _lib.ERR_HIP:      # argument errors are the caller's; a refused capture is not
                check(rc)
            import warnings
            warnings.warn('HIP graph capture failed (%s); the bound launch enqueues its kernels one by one'
                          % self.ctx.lib.mod16_last_error(self.ctx.handle).decode())
        fn = lib.mod16_et_diag_f32 if f32 else lib.mod16_et_diag_f64

        def launch():
            check(fn(*args, torch.cuda.current_stream(device).cuda_stream))
        step = BoundStep(launch, (keepalive[2], keepalive[3]), None, device)
        step._keep = keepalive
        return step

    # ------------------------------------------------------ tiled rasters
    def alloc_tiled(self, n, tile=None, form=_lib.FORM_TOTALS):
        '''A ``TiledRaster`` for n pixels (``tile`` pixels per tile, default
        ``TILE_BYTES`` per field) of one form of the forward run.'''
        return TiledRaster(self, n, tile, form)

    def run_form_tiled(self, r, day_hours=None):
        '''The forward run of the raster's form over a tiled raster
        (``mod16_et_form_tiled_*``): what ``run_pet``, ``run(out_sep=...)`` and
        ``run_raw`` compute on plain arrays. ``day_hours``: the hours of daylight of
        ``FORM_RAW_TOTAL8`` (one value; ``FORM_RAW_TOTAL8_HOURS`` reads them per pixel
        from ``r.wide[14]``). Asynchronous on the current stream; returns ``r.outs``.'''
        if r.form == _lib.FORM_RAW_TOTAL8 and day_hours is None:
            raise ValueError('FORM_RAW_TOTAL8 needs day_hours')
        fn = self.ctx.lib.mod16_et_form_tiled_f32 if self.np_dtype == np.float32 \
            else self.ctx.lib.mod16_et_form_tiled_f64
        self.ctx.check(fn(
            self.ctx.handle, C.byref(r.layout), r.form,
            _lib.ptr_array([b.data_ptr() for b in r.bytes]),
            _lib.ptr_array([w.data_ptr() for w in r.wide]),
            _lib.ptr_array([o.data_ptr() for o in r.outs]),
            float(day_hours) if day_hours is not None else 0.0, r.n, int(self.math),
            self._stream()))
        return r.outs

    def _tiled_args(self, r):
        return (C.byref(r.layout), r.cls.data_ptr(),
                _lib.ptr_array([d.data_ptr() for d in r.drivers]), r.n,
                r.day.data_ptr(), r.night.data_ptr(), int(self.math))

    def synth_tiled(self, r, seed=16, step=0, pixel_offset=0):
        '''The synthetic drivers of ``synth`` written straight into the tiled
        raster ``r`` (``mod16_synth_tiled_*``): same field, pixel for pixel.'''
        fn = self.ctx.lib.mod16_synth_tiled_f32 if self.np_dtype == np.float32 \
            else self.ctx.lib.mod16_synth_tiled_f64
        self.ctx.check(fn(
            self.ctx.handle, C.byref(r.layout), int(seed), int(step), int(pixel_offset), r.n,
            r.cls.data_ptr(), _lib.ptr_array([d.data_ptr() for d in r.drivers]), self._stream()))
        return r

    def to_tiled(self, cls, drivers, r=None):
        '''Copy plain device arrays (class raster + 14 drivers, 1-D tensors)
        into a tiled raster (one strided copy per array).'''
        n = cls.numel()
        r = r or self.alloc_tiled(n)
        r.put(r.cls, cls)
        for k in range(14):
            r.put(r.drivers[k], drivers[k])
        return r

    def run_tiled(self, r, diag=None):
        '''The fused ET kernel over a tiled raster (``mod16_et_tiled_*``),
        asynchronous on the current stream; outputs in ``r.day`` / ``r.night``,
        with ``diag`` (float64 tensor of 8 on the device) the diagnostics too.'''
        torch = _torch()
        fn = self.ctx.lib.mod16_et_tiled_f32 if self.np_dtype == np.float32 \
            else self.ctx.lib.mod16_et_tiled_f64
        dptr = self._check_tensor(diag, torch.float64, 8, 'diag') if diag is not None else None
        self.ctx.check(fn(self.ctx.handle, *self._tiled_args(r), dptr, self._stream()))
        return r.day, r.night

    def time_tiled(self, r, launches=10, diag=None):
        '''Mean milliseconds per direct launch of ``run_tiled`` (HIP events on the
        current stream, ``mod16_time_et_tiled``).'''
        torch = _torch()
        ms = C.c_float(0)
        lay, cls, drv, n, day, night, math = self._tiled_args(r)
        self.ctx.check(self.ctx.lib.mod16_time_et_tiled(
            self.ctx.handle, int(self.np_dtype == np.float32), lay, cls, drv, n, day, night, math,
            self._check_tensor(diag, torch.float64, 8, 'diag') if diag is not None else None,
            int(launches), self._stream(), C.byref(ms)))
        return ms.value

    def bind_tiled(self, r, diag):
        '''``run_tiled(r, diag)`` captured once into a HIP graph
        (``mod16_graph_et_tiled_*``); the returned ``BoundStep`` replays it with
        one library call and can time itself.'''
        torch = _torch()
        handle = C.c_void_p()
        lib, check, device = self.ctx.lib, self.ctx.check, self.device
        make = lib.mod16_graph_et_tiled_f32 if self.np_dtype == np.float32 \
            else lib.mod16_graph_et_tiled_f64
        check(make(self.ctx.handle, *self._tiled_args(r),
                   self._check_tensor(diag, torch.float64, 8, 'diag'), C.byref(handle)))
        owner = _GraphHandle(lib, handle, self.ctx)
        keep = (r, diag)

        def launch():
            check(lib.mod16_graph_launch(owner.handle, torch.cuda.current_stream(device).cuda_stream))
        step = BoundStep(launch, (r.day, r.night), owner, device)
        step._keep = keep
        return step

    def run_series_tiled(self, n, steps, seed=16, pixel_offset=0, on_step=None, ring=None):
        '''``run_series`` on tiled rasters: a ring of two ``TiledRaster`` slots,
        the drivers of step s + 1 produced on a second stream while the kernel
        works on step s. Returns ``(diag, raster of the last step)``.'''
        torch = _torch()
        dev = self._dev()
        compute = torch.cuda.current_stream(self.device)
        ingest = torch.cuda.Stream(device=dev)
        ring = ring or [self.alloc_tiled(n), self.alloc_tiled(n)]
        diag = torch.zeros(steps, 8, dtype=torch.float64, device=dev)
        filled = [torch.cuda.Event() for _ in range(steps)]
        consumed = [torch.cuda.Event() for _ in range(steps)]
        ingest.wait_stream(compute)

        def produce(s):
            with torch.cuda.stream(ingest):
                if s >= 2:
                    ingest.wait_event(consumed[s - 2])
                self.synth_tiled(ring[s % 2], seed=seed, step=s, pixel_offset=pixel_offset)
                filled[s].record(ingest)

        for s in range(min(2, steps)):
            produce(s)
        last = None
        for s in range(steps):
            compute.wait_event(filled[s])
            self._gate()
            last = ring[s % 2]
            self.run_tiled(last, diag=diag[s])
            if on_step is not None:
                on_step(s, last)
            consumed[s].record(compute)
            if s + 2 < steps:
                produce(s + 2)
        compute.wait_stream(ingest)
        return diag, last

    def run_series_host(self, ring, host_steps, steps, day_hours=None, on_step=None):
        '''A time series whose drivers arrive from HOST memory (a real ingest: SURVEY.md 8f
        N1 / N4, reference calibration.py:380-423): ``ring`` is two ``TiledRaster`` slots of one
        form -- typically ``FORM_RAW`` on a float32 engine, the light input form: 14 float32 raw
        fields + uint8 fPAR / LAI = 58 bytes per pixel and step over PCIe instead of the 113 of
        float64 plain drivers. ``host_steps`` is a list of K step records ``{'wide': [...],
        'bytes': [...]}`` of page-locked CPU tensors with ``ntiles * tile`` elements each (``None``
        = the field stays as it is in the slots: the class raster, static fields); step ``s``
        takes record ``s % K``. The copies of step s + 1 run on a second stream -- tile-wide rows
        straight into the slot's pitch, no repacking pass -- under the kernel of step s; the
        kernel is ``run_form_tiled`` (the reference's pre-processing is inside it).
        ``on_step(s, slot)`` is called with the compute stream current behind step s's kernel.
        Returns the slot of the last step.'''
        torch = _torch()
        compute = torch.cuda.current_stream(self.device)
        ingest = torch.cuda.Stream(device=self._dev())
        filled = [torch.cuda.Event() for _ in range(steps)]
        consumed = [torch.cuda.Event() for _ in range(steps)]
        ingest.wait_stream(compute)

        def produce(s):
            rec, slot = host_steps[s % len(host_steps)], ring[s % 2]
            with torch.cuda.stream(ingest):
                if s >= 2:
                    ingest.wait_event(consumed[s - 2])      # the slot is free again
                for dst, src in list(zip(slot.wide, rec.get('wide', []))) + list(zip(slot.bytes, rec.get('bytes', []))):
                    if src is not None:
                        dst.copy_(src.view(slot.ntiles, slot.tile), non_blocking=True)
                filled[s].record(ingest)

        for s in range(min(2, steps)):
            produce(s)
        last = None
        for s in range(steps):
            compute.wait_event(filled[s])
            self._gate()
            last = ring[s % 2]
            self.run_form_tiled(last, day_hours)
            if on_step is not None:
                on_step(s, last)
            consumed[s].record(compute)
            if s + 2 < steps:
                produce(s + 2)
        compute.wait_stream(ingest)
        return last

    def measure_copy(self, nbytes=4 << 30, reps=3):
        '''GB/s of a plain device-to-device copy kernel on this GPU
        (``mod16_measure_copy``): the measured reference point next to the
        nominal HBM peak.'''
        gbps = C.c_float(0)
        self.ctx.check(self.ctx.lib.mod16_measure_copy(self.ctx.handle, int(nbytes), int(reps),
                                                       C.byref(gbps)))
        return gbps.value

    def fold_ranks(self, gathered, world, diag):
        '''The rank-order fold behind the all-gather of the diagnostics vectors
        (``mod16_fold_diag``; ``mod16_amd.dist.allreduce_diag`` calls it): ``gathered`` is the
        ``(world, 8)`` float64 block, ``diag`` (8) receives sums in rank order and maxima. One
        kernel on the current stream.'''
        torch = _torch()
        self.ctx.check(self.ctx.lib.mod16_fold_diag(
            self.ctx.handle, self._check_tensor(gathered, torch.float64, world * 8, 'gathered'),
            int(world), self._check_tensor(diag, torch.float64, 8, 'diag'), self._stream()))
        return diag

    def check(self):
        '''Synchronise and raise deferred errors (IndexError for a class code
        >= 13, as the reference's numpy gather would).'''
        self.ctx.check_status(self._stream())

    def diagnostics(self, day, night, out=None):
        '''Deterministic reduction of a (day, night) pair on the device ->
        float64 tensor of 8 (``DIAG_FIELDS``), asynchronous.'''
        torch = _torch()
        n = day.numel()
        if out is None:
            out = torch.empty(8, dtype=torch.float64, device=self._dev())
        fn = self.ctx.lib.mod16_reduce_diag_f32 if self.np_dtype == np.float32 \
            else self.ctx.lib.mod16_reduce_diag_f64
        self.ctx.check(fn(
            self.ctx.handle, self._check_tensor(day, self.dtype, n, 'day'),
            self._check_tensor(night, self.dtype, n, 'night'), n, None,
            self._check_tensor(out, torch.float64, 8, 'out'), self._stream()))
        return out

    def time_kernel(self, cls, drivers, out_day, out_night, launches=10, diag=None):
        '''Mean milliseconds per launch of the fused kernel (with ``diag``: of
        the kernel that also reduces the diagnostics), measured with HIP
        events on the stream the kernel runs on (``mod16_time_et``).'''
        torch = _torch()
        n = cls.numel()
        cptr = self._check_tensor(cls, torch.uint8, n, 'cls')
        keep, dptr, dstride = self._marshal_drivers(drivers, n)
        ms = C.c_float(0)
        self.ctx.check(self.ctx.lib.mod16_time_et(
            self.ctx.handle, int(self.np_dtype == np.float32), cptr,
            _lib.ptr_array(dptr), _lib.i64_array(dstride), None, None, n,
            self._check_tensor(out_day, self.dtype, n, 'out_day'),
            self._check_tensor(out_night, self.dtype, n, 'out_night'), None,
            int(self.math),
            self._check_tensor(diag, torch.float64, 8, 'diag') if diag is not None else None,
            int(launches), self._stream(), C.byref(ms)))
        return ms.value


class ShardedSeries(object):
    '''
    ``RasterEngine.run_series_host`` over several GPUs of one node, in one process
    (``mod16_amd.multi``; SURVEY.md 8e for the PCIe-bound ingest: N links instead of one,
    no collective). The tiles of the raster are dealt over ``devices`` in order --
    device ``i`` holds tiles ``[t0_i, t1_i)`` in a two-slot ring of ``TiledRaster`` of
    its own, fed by its own host thread, ingest stream and engine -- and every step's
    kernel runs on each device over its tiles only. A pixel's result does not depend on
    the device list (``devices=[0]`` and ``[0, 0]`` give the same bits).

    Parameters
    ----------
    table : numpy.ndarray
        (13, 11) BPLUT table
    n : int
        Pixels of the raster (a multiple of the 16-byte vector width)
    devices : sequence of int
        GPU indices; one may be listed more than once
    form, dtype, math, trusted, tile
        As for ``RasterEngine`` / ``RasterEngine.alloc_tiled``; the defaults are the light
        input form (``FORM_RAW`` on float32: 58 bytes per pixel and step over PCIe)
    '''

    def __init__(self, table, n, devices, form=_lib.FORM_RAW, dtype='float32',
                 math=_lib.MATH_FAST, trusted=False, tile=None):
        from . import multi
        torch = _torch()
        devs = multi.device_list(devices)
        if devs is None:
            raise ValueError('devices is required')
        self.n, self.form = int(n), int(form)
        self.parts = []             # per device that holds tiles: dict(device, engine, ring, t0, t1, offset, n)
        probe = RasterEngine(table, device=devs[0], dtype=dtype, math=math, trusted=trusted)
        self.tile = int(tile) if tile else probe.TILE_BYTES // probe.np_dtype.itemsize
        self.ntiles = max(1, -(-self.n // self.tile))
        for i, (t0, count) in enumerate(multi.shards(self.ntiles, len(devs), 1)):
            if not count:
                continue
            lo, hi = t0 * self.tile, min(self.n, (t0 + count) * self.tile)
            with torch.cuda.device(devs[i]):
                eng = probe if (i == 0) else RasterEngine(table, device=devs[i], dtype=dtype, math=math,
                                                           trusted=trusted)
                ring = [eng.alloc_tiled(hi - lo, self.tile, self.form) for _ in range(2)]
                stream = torch.cuda.Stream(device=eng._dev())
            self.parts.append(dict(device=devs[i], engine=eng, ring=ring, stream=stream,
                                   t0=t0, t1=t0 + count, offset=lo, n=hi - lo))

    def run_host(self, host_steps, steps, day_hours=None, on_step=None):
        '''``run_series_host`` on every device at once. ``host_steps``: as there, records of
        page-locked CPU tensors covering the WHOLE raster (``ntiles * tile`` elements; ``None`` = the
        field stays as it is); each device copies its tiles' part. ``on_step(part, s, slot)``
        is called on the device's own host thread (compute stream current) behind step ``s``'s
        kernel, ``part`` being the entry of ``self.parts`` (``part['offset']`` = first pixel of
        the slot in the raster). Returns the slots of the last step, one per part. Errors of any
        device are raised here (first in device order) after all of them have finished.'''
        import threading
        torch = _torch()
        P = self.tile
        results, errors = [None] * len(self.parts), [None] * len(self.parts)

        def cut(rec, part):
            a, b = part['t0'] * P, part['t1'] * P
            return {key: [None if t is None else t[a:b] for t in rec.get(key, [])] for key in ('wide', 'bytes')}

        def work(k, part):
            try:
                torch.cuda.set_device(part['device'])
                mine = [cut(rec, part) for rec in host_steps]
                with torch.cuda.stream(part['stream']):
                    hook = (lambda s, slot: on_step(part, s, slot)) if on_step is not None else None
                    results[k] = part['engine'].run_series_host(part['ring'], mine, steps, day_hours, hook)
                    part['engine'].check()
            except BaseException as exc:
                errors[k] = exc

        threads = [threading.Thread(target=work, args=(k, p)) for k, p in enumerate(self.parts)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for e in errors:
            if e is not None:
                raise e
        return results

    def read(self, slots, out_index, into):
        '''Copies output ``out_index`` of the per-device ``slots`` (what ``run_host`` returned)
        into the 1-D CPU tensor ``into`` (n elements, ideally page-locked), every part at its
        pixel offset; synchronous.'''
        torch = _torch()
        for part, slot in zip(self.parts, slots):
            with torch.cuda.device(part['device']):
                flat = slot.flat(slot.outs[out_index], 0, part['n'])
                into[part['offset']:part['offset'] + part['n']].copy_(flat)
        return into
