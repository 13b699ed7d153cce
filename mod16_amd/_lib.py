'''
ctypes binding of ``libmod16hip.so`` (C ABI: ``include/mod16_hip.h``).

There is no CPU fallback: if the library is missing, or no MI355X is visible
when a computation is requested, the call fails loudly.
'''
import ctypes as C
import os
import threading
import weakref

import numpy as np

N_DRIVERS = 14
N_PARAMS = 11
N_CLASSES = 13
N_COMPONENTS = 6
HOST, DEVICE = 0, 1
BC_SCALAR, BC_DENSE, BC_ROW, BC_COL = 0, 1, 2, 3      # enum mod16_broadcast
MATH_FAST, MATH_EXACT, MATH_MIXED = 0, 1, 2
DOMAIN_TRUSTED = 4        # flag bit MOD16_DOMAIN_TRUSTED: or it into a math value (include/mod16_hip.h)
# enum mod16_form: (wide arrays, byte rasters, outputs) of each form of the forward run
(FORM_TOTALS, FORM_PET, FORM_COMPONENTS, FORM_TOTALS_COMPONENTS, FORM_RAW, FORM_RAW_TOTAL8,
 FORM_RAW_TOTAL8_HOURS) = range(7)
FORM_SHAPE = {FORM_TOTALS: (14, 1, 2), FORM_PET: (14, 1, 4), FORM_COMPONENTS: (14, 1, 6),
              FORM_TOTALS_COMPONENTS: (14, 1, 8), FORM_RAW: (14, 3, 2), FORM_RAW_TOTAL8: (14, 3, 3),
              FORM_RAW_TOTAL8_HOURS: (15, 3, 3)}

METHOD_MAX_IN = 13
(M_SVP, M_SVP_SLOPE, M_LHV, M_PSYCHROMETRIC, M_RADIATION_NET, M_AIR_DENSITY,
 M_AIR_PRESSURE, M_VPD, M_RHUMIDITY, M_POT_SOIL_EVAP, M_POT_TRANSPIRATION,
 M_EVAP_SOIL, M_EVAP_WET_CANOPY, M_RADIATION_SOIL, M_SOIL_HEAT_FLUX,
 M_SURFACE_CONDUCTANCE, M_TRANSPIRATION_DAY, M_TRANSPIRATION_NIGHT) = range(18)

OK = 0
ERR_ARG, ERR_HIP, ERR_CLASS_RANGE, ERR_NOMEM, ERR_NO_DEVICE, ERR_NO_BPLUT = \
    -1, -2, -3, -4, -5, -6

ABI_VERSION = 5
LIB_NAME = 'libmod16hip.so'
# MOD16_LIB: alternative build of the same library (kernel experiments only)
LIB_PATH = os.environ.get('MOD16_LIB') or os.path.join(
    os.path.dirname(os.path.abspath(__file__)), LIB_NAME)
# the same sources built with -DMOD16_EXPERIMENTS (launch-geometry overrides read from the
# environment at context creation): tests and tools only, see load_experiments()
EXP_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libmod16hip_exp.so')


class Mod16Error(RuntimeError):
    '''A failed call into libmod16hip (anything but a class-range error).'''

    def __init__(self, status, message):
        super().__init__('libmod16hip: %s (status %d)' % (message, status))
        self.status = status


_PP = C.POINTER(C.c_void_p)
_I64P = C.POINTER(C.c_int64)


class Layout(C.Structure):
    '''``mod16_layout`` (include/mod16_hip.h): pixel i of an array of a tiled
    raster lives at ``base[(i // tile) * row + (i % tile)]``.'''
    _fields_ = [('tile', C.c_int64), ('driver_row', C.c_int64),
                ('out_row', C.c_int64), ('cls_row', C.c_int64)]


_LAYP = C.POINTER(Layout)

# name -> (restype, argtypes); one entry per function declared in the header
PROTOTYPES = {
    'mod16_version': (C.c_int, []),
    'mod16_strerror': (C.c_char_p, [C.c_int]),
    'mod16_last_error': (C.c_char_p, [C.c_void_p]),
    'mod16_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'mod16_create': (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    'mod16_destroy': (C.c_int, [C.c_void_p]),
    'mod16_set_bplut_f64': (C.c_int, [C.c_void_p, C.c_void_p]),
    'mod16_et_f64': (C.c_int, [
        C.c_void_p, C.c_void_p, _PP, _I64P, _PP, _I64P, C.c_int64, C.c_void_p,
        C.c_void_p, _PP, C.c_uint, C.c_int, C.c_void_p]),
    'mod16_et_f32': (C.c_int, [
        C.c_void_p, C.c_void_p, _PP, _I64P, _PP, _I64P, C.c_int64, C.c_void_p,
        C.c_void_p, _PP, C.c_uint, C.c_int, C.c_void_p]),
    'mod16_et2_f64': (C.c_int, [
        C.c_void_p, C.c_void_p, C.c_int, _PP, _I64P, _PP, _I64P, C.c_int64, C.c_int64,
        C.c_void_p, C.c_void_p, _PP, C.c_uint, C.c_int, C.c_void_p]),
    'mod16_et2_f32': (C.c_int, [
        C.c_void_p, C.c_void_p, C.c_int, _PP, _I64P, _PP, _I64P, C.c_int64, C.c_int64,
        C.c_void_p, C.c_void_p, _PP, C.c_uint, C.c_int, C.c_void_p]),
    'mod16_host_tile_pixels': (C.c_int64, []),
    'mod16_et_hdiag_f64': (C.c_int, [
        C.c_void_p, C.c_void_p, _PP, _I64P, _PP, _I64P, C.c_int64, C.c_void_p,
        C.c_void_p, C.c_uint, C.c_void_p]),
    'mod16_et_hdiag_f32': (C.c_int, [
        C.c_void_p, C.c_void_p, _PP, _I64P, _PP, _I64P, C.c_int64, C.c_void_p,
        C.c_void_p, C.c_uint, C.c_void_p]),
    'mod16_fold_diag_host': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    'mod16_et_pet_f64': (C.c_int, [
        C.c_void_p, C.c_void_p, _PP, _I64P, _PP, _I64P, C.c_int64, C.c_void_p,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_int, C.c_void_p]),
    'mod16_et_pet_f32': (C.c_int, [
        C.c_void_p, C.c_void_p, _PP, _I64P, _PP, _I64P, C.c_int64, C.c_void_p,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_int, C.c_void_p]),
    'mod16_et_raw_f64': (C.c_int, [
        C.c_void_p, C.c_void_p, _PP, _I64P, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint,
        C.c_int, C.c_void_p]),
    'mod16_et_raw_f32': (C.c_int, [
        C.c_void_p, C.c_void_p, _PP, _I64P, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint,
        C.c_int, C.c_void_p]),
    'mod16_et_diag_f64': (C.c_int, [
        C.c_void_p, C.c_void_p, _PP, _I64P, C.c_int64, C.c_void_p, C.c_void_p,
        C.c_uint, C.c_void_p, C.c_void_p]),
    'mod16_et_diag_f32': (C.c_int, [
        C.c_void_p, C.c_void_p, _PP, _I64P, C.c_int64, C.c_void_p, C.c_void_p,
        C.c_uint, C.c_void_p, C.c_void_p]),
    'mod16_graph_et_diag_f64': (C.c_int, [
        C.c_void_p, C.c_void_p, _PP, _I64P, C.c_int64, C.c_void_p, C.c_void_p,
        C.c_uint, C.c_void_p, C.POINTER(C.c_void_p)]),
    'mod16_graph_et_diag_f32': (C.c_int, [
        C.c_void_p, C.c_void_p, _PP, _I64P, C.c_int64, C.c_void_p, C.c_void_p,
        C.c_uint, C.c_void_p, C.POINTER(C.c_void_p)]),
    'mod16_graph_launch': (C.c_int, [C.c_void_p, C.c_void_p]),
    'mod16_graph_destroy': (C.c_int, [C.c_void_p]),
    'mod16_method_f64': (C.c_int, [
        C.c_void_p, C.c_int, _PP, _I64P, _PP, _I64P, C.c_int64, _PP, C.c_double,
        C.c_double, C.c_int, C.c_void_p]),
    'mod16_method_f32': (C.c_int, [
        C.c_void_p, C.c_int, _PP, _I64P, _PP, _I64P, C.c_int64, _PP, C.c_float,
        C.c_float, C.c_int, C.c_void_p]),
    'mod16_et_static_f64': (C.c_int, [
        C.c_void_p, _PP, _I64P, _PP, _I64P, _PP, _I64P, C.c_int64, C.c_void_p,
        C.c_void_p, C.c_double, C.c_int, C.c_void_p]),
    'mod16_et_static_f32': (C.c_int, [
        C.c_void_p, _PP, _I64P, _PP, _I64P, _PP, _I64P, C.c_int64, C.c_void_p,
        C.c_void_p, C.c_float, C.c_int, C.c_void_p]),
    'mod16_et_static_batch_f64': (C.c_int, [
        C.c_void_p, _PP, _I64P, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_uint, C.c_int, C.c_void_p]),
    'mod16_et_static_batch_f32': (C.c_int, [
        C.c_void_p, _PP, _I64P, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_uint, C.c_int, C.c_void_p]),
    'mod16_static_batch_bind_f64': (C.c_int, [
        C.c_void_p, _PP, _I64P, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint, C.c_int,
        C.POINTER(C.c_void_p)]),
    'mod16_static_batch_bind_f32': (C.c_int, [
        C.c_void_p, _PP, _I64P, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint, C.c_int,
        C.POINTER(C.c_void_p)]),
    'mod16_static_batch_objective': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'mod16_static_batch_rows': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    'mod16_static_batch_info': (C.c_int, [C.c_void_p, _I64P, _I64P, _I64P]),
    'mod16_static_batch_time': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    'mod16_static_batch_destroy': (C.c_int, [C.c_void_p]),
    'mod16_check_status': (C.c_int, [C.c_void_p, C.c_void_p]),
    'mod16_reduce_diag_f64': (C.c_int, [
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
        C.c_void_p]),
    'mod16_reduce_diag_f32': (C.c_int, [
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
        C.c_void_p]),
    'mod16_synth_f64': (C.c_int, [
        C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
        _PP, C.c_void_p]),
    'mod16_synth_f32': (C.c_int, [
        C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
        _PP, C.c_void_p]),
    'mod16_et_tiled_f64': (C.c_int, [
        C.c_void_p, _LAYP, C.c_void_p, _PP, C.c_int64, C.c_void_p, C.c_void_p,
        C.c_uint, C.c_void_p, C.c_void_p]),
    'mod16_et_tiled_f32': (C.c_int, [
        C.c_void_p, _LAYP, C.c_void_p, _PP, C.c_int64, C.c_void_p, C.c_void_p,
        C.c_uint, C.c_void_p, C.c_void_p]),
    'mod16_graph_et_tiled_f64': (C.c_int, [
        C.c_void_p, _LAYP, C.c_void_p, _PP, C.c_int64, C.c_void_p, C.c_void_p,
        C.c_uint, C.c_void_p, C.POINTER(C.c_void_p)]),
    'mod16_graph_et_tiled_f32': (C.c_int, [
        C.c_void_p, _LAYP, C.c_void_p, _PP, C.c_int64, C.c_void_p, C.c_void_p,
        C.c_uint, C.c_void_p, C.POINTER(C.c_void_p)]),
    'mod16_synth_tiled_f64': (C.c_int, [
        C.c_void_p, _LAYP, C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
        _PP, C.c_void_p]),
    'mod16_synth_tiled_f32': (C.c_int, [
        C.c_void_p, _LAYP, C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
        _PP, C.c_void_p]),
    'mod16_form_shape': (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'mod16_et_form_tiled_f64': (C.c_int, [
        C.c_void_p, _LAYP, C.c_int, _PP, _PP, _PP, C.c_double, C.c_int64, C.c_uint, C.c_void_p]),
    'mod16_et_form_tiled_f32': (C.c_int, [
        C.c_void_p, _LAYP, C.c_int, _PP, _PP, _PP, C.c_double, C.c_int64, C.c_uint, C.c_void_p]),
    'mod16_time_et_tiled': (C.c_int, [
        C.c_void_p, C.c_int, _LAYP, C.c_void_p, _PP, C.c_int64, C.c_void_p, C.c_void_p,
        C.c_uint, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_float)]),
    'mod16_time_graph': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_float)]),
    'mod16_host_alloc': (C.c_int, [C.c_int64, C.POINTER(C.c_void_p)]),
    'mod16_host_free': (C.c_int, [C.c_void_p]),
    'mod16_measure_copy': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_float)]),
    'mod16_build_id': (C.c_char_p, []),
    'mod16_fold_diag': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'mod16_classify_f64': (C.c_int, [C.c_void_p, _PP, _I64P, C.c_int64, C.c_void_p, C.c_int, C.c_void_p,
                                     C.POINTER(C.c_int64), C.c_void_p]),
    'mod16_classify_f32': (C.c_int, [C.c_void_p, _PP, _I64P, C.c_int64, C.c_void_p, C.c_int, C.c_void_p,
                                     C.POINTER(C.c_int64), C.c_void_p]),
    'mod16_time_et': (C.c_int, [
        C.c_void_p, C.c_int, C.c_void_p, _PP, _I64P, _PP, _I64P, C.c_int64,
        C.c_void_p, C.c_void_p, _PP, C.c_uint, C.c_void_p, C.c_int, C.c_void_p,
        C.POINTER(C.c_float)]),
}

_lib = None


def _preload_torch_hip_runtime():
    '''PyTorch-ROCm wheels bundle their own HIP runtime (torch/lib/
    libamdhip64.so, SONAME libamdhip64.so.7). If libmod16hip.so pulled in the
    system copy first and torch were imported later, the process would hold two
    HIP/HSA runtimes and the second one could not open the GPU. Loading torch's
    copy first makes both resolve to the same runtime, in either import order.
    torch itself is not imported here.'''
    import importlib.util
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), 'lib', 'libamdhip64.so')
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def _declare(lib, tolerate_missing):
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            # an older experimental build given through MOD16_LIB (tools/kbench.py
            # A/B runs) may predate an entry point; the in-tree library may not
            if tolerate_missing:
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    if lib.mod16_version() != ABI_VERSION:
        raise ImportError('libmod16hip ABI version %d, expected %d'
                          % (lib.mod16_version(), ABI_VERSION))
    return lib


def load():
    '''Load libmod16hip.so (once) and declare its prototypes.'''
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            '%s is not built: run `python -c "import __graft_entry__ as g; '
            'g.build()"` (or mod16_amd/csrc/build.py) at the repo root. '
            'mod16_amd has no CPU fallback.' % LIB_PATH)
    _preload_torch_hip_runtime()
    _lib = _declare(C.CDLL(LIB_PATH), 'MOD16_LIB' in os.environ)
    return _lib


_exp_lib = None


def load_experiments():
    '''The experiments build of the library (``libmod16hip_exp.so``: the same sources with
    ``-DMOD16_EXPERIMENTS``), whose contexts read the launch-geometry overrides ``MOD16_NO_DMA``,
    ``MOD16_RUN_SHIFT``, ``MOD16_STATIC_BELOW``, ``MOD16_STREAM_BLOCKS``, ``MOD16_PITCH``,
    ``MOD16_GRID_MULT`` (and the fault injection ``MOD16_POISON_TICKET``) from the environment when they are created. For tests and tools: the
    product never calls this (``Context(device, experiments=True)`` is how a test asks for it).'''
    global _exp_lib
    if _exp_lib is None:
        if not os.path.exists(EXP_LIB_PATH):
            raise ImportError('%s is not built (mod16_amd/csrc/build.py builds it)' % EXP_LIB_PATH)
        load()          # the HIP runtime both resolve to
        _exp_lib = _declare(C.CDLL(EXP_LIB_PATH), False)
    return _exp_lib


def build_id():
    '''Digest of the sources and flags the loaded library was built from (``mod16_build_id``).'''
    return load().mod16_build_id().decode()


def device_count():
    n = C.c_int(0)
    load().mod16_device_count(C.byref(n))
    return n.value


_ARRAY_TYPES = {}


def _array_type(base, n):
    t = _ARRAY_TYPES.get((base, n))
    if t is None:
        t = _ARRAY_TYPES[(base, n)] = base * n
    return t


def ptr_array(ptrs):
    '''Python ints / None -> void*[len]'''
    return _array_type(C.c_void_p, len(ptrs))(*ptrs)


def i64_array(vals):
    try:
        return _array_type(C.c_int64, len(vals))(*vals)
    except TypeError:       # numpy integers
        return _array_type(C.c_int64, len(vals))(*[int(v) for v in vals])


class Context:
    '''One device context (``mod16_ctx``): BPLUT, staging tiles, workspace.'''

    def __init__(self, device=0, experiments=False):
        self.lib = load_experiments() if experiments else load()
        self.handle = C.c_void_p()
        self.device = device
        rc = self.lib.mod16_create(int(device), C.byref(self.handle))
        if rc != OK:
            self.handle = C.c_void_p()
            raise Mod16Error(rc, (
                'cannot create a context on device %d: %s -- mod16_amd needs '
                'an MI355X (gfx950); there is no CPU fallback'
                % (device, self.lib.mod16_strerror(rc).decode())))
        self._bplut_key = None

    def close(self):
        if getattr(self, 'handle', None) and self.handle.value:
            self.lib.mod16_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc == OK:
            return
        msg = self.lib.mod16_last_error(self.handle).decode() or \
            self.lib.mod16_strerror(rc).decode()
        if rc == ERR_CLASS_RANGE:
            # what numpy raises for params_dict[key][pft_map] with a code >= 13
            raise IndexError(msg)
        raise Mod16Error(rc, msg)

    def set_bplut(self, table):
        '''table: float64 [13][11], rows = PFT code, columns in
        MOD16.required_parameters order.'''
        table = np.ascontiguousarray(table, np.float64)
        if table.shape != (N_CLASSES, N_PARAMS):
            raise ValueError('BPLUT table must have shape (13, 11), got %r'
                             % (table.shape,))
        key = table.tobytes()
        if key != self._bplut_key:
            self.check(self.lib.mod16_set_bplut_f64(
                self.handle, table.ctypes.data))
            self._bplut_key = key

    def et(self, dtype, cls, drivers, dstride, params, pstride, n, out_day,
           out_night, out_sep, flags=MATH_FAST, where=HOST, stream=None):
        '''Thin wrapper of mod16_et_f64 / mod16_et_f32; every array argument
        is a raw address (int) or None.'''
        fn = self.lib.mod16_et_f32 if dtype == np.float32 \
            else self.lib.mod16_et_f64
        self.check(fn(
            self.handle, cls, ptr_array(drivers), i64_array(dstride),
            ptr_array(params) if params is not None else None,
            i64_array(pstride) if pstride is not None else None,
            int(n), out_day, out_night,
            ptr_array(out_sep) if out_sep is not None else None,
            int(flags), int(where), stream))

    def et2(self, dtype, cls, cls_kind, drivers, dkind, params, pkind, inner, n,
            out_day, out_night, out_sep, flags=MATH_FAST, where=HOST, stream=None):
        '''mod16_et2_f64 / mod16_et2_f32: as ``et`` with a broadcast kind
        (``BC_*``) per input instead of a 0 / 1 stride.'''
        fn = self.lib.mod16_et2_f32 if np.dtype(dtype) == np.float32 \
            else self.lib.mod16_et2_f64
        self.check(fn(
            self.handle, cls, int(cls_kind), ptr_array(drivers), i64_array(dkind),
            ptr_array(params) if params is not None else None,
            i64_array(pkind) if pkind is not None else None,
            int(inner), int(n), out_day, out_night,
            ptr_array(out_sep) if out_sep is not None else None,
            int(flags), int(where), stream))

    def method(self, dtype, method, inputs, istride, params, pstride, n, outs,
               alpha=1.26, tiny=1e-7, where=HOST, stream=None):
        '''Thin wrapper of mod16_method_f64 / _f32 (raw addresses or None).'''
        fn = self.lib.mod16_method_f32 if np.dtype(dtype) == np.float32 \
            else self.lib.mod16_method_f64
        self.check(fn(
            self.handle, int(method), ptr_array(inputs), i64_array(istride),
            ptr_array(params) if params is not None else None,
            i64_array(pstride) if pstride is not None else None,
            int(n), ptr_array(outs), float(alpha), float(tiny), int(where), stream))

    def check_status(self, stream=None):
        self.check(self.lib.mod16_check_status(self.handle, stream))


def _host_ram_bytes():
    try:
        return os.sysconf('SC_PAGE_SIZE') * os.sysconf('SC_PHYS_PAGES')
    except (ValueError, OSError, AttributeError):
        return 0


def _default_pinned_live():
    '''Default bound on page-locked result memory of THIS process: a quarter of the host's RAM,
    between 8 and 64 GiB -- divided by the number of processes the launcher put on the node
    (LOCAL_WORLD_SIZE, else WORLD_SIZE: one rank per GPU), so that eight ranks together stay within
    that quarter instead of pinning twice the RAM (never below 1 GiB per rank).'''
    try:
        ranks = max(1, int(os.environ.get('LOCAL_WORLD_SIZE') or os.environ.get('WORLD_SIZE') or 1))
    except ValueError:
        ranks = 1
    whole = min(64 << 30, max(8 << 30, _host_ram_bytes() // 4))
    return max(1 << 30, whole // ranks)


class _PinnedPool(object):
    '''Page-locked host blocks behind the result arrays of the numpy entry points.

    The reference returns fresh arrays; writing 1.5 GB of results into fresh
    pageable memory costs a page fault per 4 KiB (13 GB/s measured) where PCIe
    moves 57 GB/s, so result arrays of ``MIN_BYTES`` or more are numpy views of
    ``hipHostMalloc`` blocks. A block goes back to the pool when its array is
    garbage-collected and is handed out again for the next result of that size
    (a time loop allocates once). Two bounds: at most ``MAX_LIVE`` bytes are
    page-locked in total (default: a quarter of the host's RAM, between 8 and 64 GiB;
    ``MOD16_PINNED_LIVE``) -- beyond that ``empty`` returns plain numpy arrays and says
    so once (``warnings``; ``fallbacks`` counts them) -- and idle blocks are kept up to the
    larger of ``MAX_CACHED`` (1 GiB; ``MOD16_PINNED_CACHE``) and the sizes of the last eight
    results handed out, so the results of one step of a time loop always find their
    blocks again. ``trim()`` frees the idle blocks. Everything else about the arrays is
    ordinary (writeable, C-contiguous, own their memory through ``.base``).

    Locking: ``give`` runs from ``weakref.finalize``, i.e. possibly inside a garbage
    collection triggered while ``take`` / ``trim`` holds the lock on the same thread -- the lock
    is re-entrant, and ``hipHostFree`` (which synchronises the device) is only ever called
    after the lock has been released: a ``give`` that finds its own thread inside the lock leaves
    the blocks it wants freed on ``deferred``, and whoever holds the lock frees them on the way out.'''
    MIN_BYTES = 1 << 20
    MAX_CACHED = int(os.environ.get('MOD16_PINNED_CACHE', 1 << 30))
    MAX_LIVE = int(os.environ.get('MOD16_PINNED_LIVE', _default_pinned_live()))

    def __init__(self):
        import collections
        self.lock = threading.RLock()
        self.free = {}          # nbytes -> [address, ...]
        self.cached = 0         # idle bytes in `free`
        self.live = 0           # bytes allocated (handed out + idle)
        self.recent = collections.deque(maxlen=8)     # sizes of the last results handed out
        self.fallbacks = 0      # results that had to be plain numpy arrays
        self._warned = False
        self.deferred = []      # blocks a re-entrant give() wants freed (by the lock's holder, after release)
        self._inside = threading.local()    # depth of this thread inside `with self.lock`

    def _enter(self):
        self.lock.acquire()
        self._inside.n = getattr(self._inside, 'n', 0) + 1

    def _leave(self, doomed):
        '''Releases the lock; the outermost holder takes the deferred blocks along.'''
        self._inside.n -= 1
        if self._inside.n == 0 and self.deferred:
            doomed.extend(self.deferred)
            del self.deferred[:]
        self.lock.release()

    def _cache_bound(self):
        return min(self.MAX_LIVE, max(self.MAX_CACHED, sum(self.recent)))

    def _free_blocks(self, doomed):
        '''hipHostFree of blocks already taken off the books; called WITHOUT the lock.'''
        for addr in doomed:
            try:
                load().mod16_host_free(addr)
            except Exception:       # interpreter shutdown
                pass

    def _trim_locked(self, want, doomed):
        '''Takes idle blocks worth `want` bytes off the books (largest first) into `doomed`.'''
        for size in sorted(self.free, reverse=True):
            blocks = self.free[size]
            while blocks and want > 0:
                doomed.append(blocks.pop())
                self.cached -= size
                self.live -= size
                want -= size
        return want

    def take(self, nbytes):
        doomed = []
        self._enter()
        try:
            self.recent.append(nbytes)
            blocks = self.free.get(nbytes)
            hit = blocks.pop() if blocks else None
            room = True
            if hit is not None:
                self.cached -= nbytes
            else:
                if self.live + nbytes > self.MAX_LIVE:
                    self._trim_locked(self.live + nbytes - self.MAX_LIVE, doomed)
                    room = self.live + nbytes <= self.MAX_LIVE
                if room:
                    self.live += nbytes
        finally:
            self._leave(doomed)
        self._free_blocks(doomed)
        if hit is not None:
            return hit
        if not room:
            self._over_bound(nbytes)
            return None
        p = C.c_void_p()
        if load().mod16_host_alloc(nbytes, C.byref(p)) != OK or not p.value:
            with self.lock:         # no page-locked memory to be had (no GPU / the host's limit)
                self.live -= nbytes
            return None
        return p.value

    def _over_bound(self, nbytes):
        self.fallbacks += 1
        if not self._warned:
            self._warned = True
            import warnings
            warnings.warn(
                'mod16_amd: a result array of %.3f GB did not fit the page-locked pool (%.3f of %.3f GB '
                'in use; MOD16_PINNED_LIVE raises the bound): it is an ordinary numpy array and its '
                'device-to-host copy runs at the page-fault rate, not the PCIe rate'
                % (nbytes / 1e9, self.live / 1e9, self.MAX_LIVE / 1e9), RuntimeWarning, stacklevel=4)

    def trim(self, keep=0):
        '''Free idle blocks until at most ``keep`` bytes of them remain.'''
        doomed = []
        self._enter()
        try:
            self._trim_locked(self.cached - keep, doomed)
        finally:
            self._leave(doomed)
        self._free_blocks(doomed)

    def give(self, addr, nbytes):
        doomed = []
        reentrant = getattr(self._inside, 'n', 0) > 0      # a finalizer running inside take() / trim() of this thread
        self._enter()
        try:
            if self.cached + nbytes <= self._cache_bound():
                self.free.setdefault(nbytes, []).append(addr)
                self.cached += nbytes
            else:
                self.live -= nbytes
                (self.deferred if reentrant else doomed).append(addr)
        finally:
            self._leave(doomed)
        if doomed:
            self._free_blocks(doomed)

    def empty(self, shape, dtype):
        '''``numpy.empty(shape, dtype)``, page-locked when large enough.'''
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        if nbytes < self.MIN_BYTES or os.environ.get('MOD16_PINNED_RESULTS') == '0':
            return np.empty(shape, dtype)
        addr = self.take(nbytes)
        if addr is None:
            return np.empty(shape, dtype)
        buf = (C.c_char * nbytes).from_address(addr)
        weakref.finalize(buf, self.give, addr, nbytes)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)


pinned = _PinnedPool()
_local = threading.local()


def context(device=0):
    '''The calling thread's context on ``device`` (created on first use).

    One context per host thread and GPU: the reference's functions are pure and
    may be called from several threads at once; a context
    owns staging slabs (HOST mode: up to ``MOD16_HOST_THREADS`` = 8 of them, ~0.6 GB
    each for float64, allocated on the first numpy call of the thread and freed
    with it -- N threads on the numpy path hold N x 4.7 GB of HBM), streams, a
    BPLUT copy and a workspace, so threads that shared one would take turns (the library serialises calls on a ctx) and
    would see each other's ``set_bplut``. Contexts of finished threads are
    destroyed with the thread's locals.'''
    table = getattr(_local, 'contexts', None)
    if table is None:
        table = _local.contexts = {}
    ctx = table.get(device)
    if ctx is None:
        ctx = table[device] = Context(device)
    return ctx
