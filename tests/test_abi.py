"""The C-ABI library: it loads, exports every function include/mod16_hip.h
declares, and the ctypes prototypes cover exactly that set. No compute call is
made here (no GPU in the build container)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, 'include', 'mod16_hip.h')


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'MOD16_API\s+[\w\s\*]+?\b(mod16_\w+)\s*\(', text)))


@pytest.fixture(scope='module')
def lib():
    from mod16_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib


def test_header_declares_the_expected_surface():
    names = declared_functions()
    assert len(names) >= 16
    for must in ('mod16_create', 'mod16_destroy', 'mod16_set_bplut_f64', 'mod16_et_f64',
                 'mod16_et_f32', 'mod16_et_diag_f64', 'mod16_reduce_diag_f64',
                 'mod16_synth_f64', 'mod16_check_status', 'mod16_strerror'):
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    cdll = ctypes.CDLL(lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(cdll, name), 'libmod16hip.so does not export %s' % name


def test_ctypes_prototypes_match_the_header(lib):
    assert sorted(lib.PROTOTYPES) == declared_functions()
    loaded = lib.load()
    assert loaded.mod16_version() == lib.ABI_VERSION
    assert re.search(r"#define MOD16_ABI_VERSION %d\b" % lib.ABI_VERSION, open(HEADER).read())
    assert loaded.mod16_strerror(0) == b'ok'
    assert b'class' in loaded.mod16_strerror(lib.ERR_CLASS_RANGE)
    assert loaded.mod16_strerror(-99) == b'unknown status'


def test_enums_agree_with_python_constants(lib):
    text = open(HEADER).read()
    for name, value in (('MOD16_ERR_ARG', lib.ERR_ARG), ('MOD16_ERR_HIP', lib.ERR_HIP),
                        ('MOD16_ERR_CLASS_RANGE', lib.ERR_CLASS_RANGE),
                        ('MOD16_ERR_NO_DEVICE', lib.ERR_NO_DEVICE),
                        ('MOD16_ERR_NO_BPLUT', lib.ERR_NO_BPLUT)):
        assert re.search(r'%s\s*=\s*%d\b' % (name, value), text), name
    assert re.search(r'#define MOD16_N_DRIVERS 14', text) and lib.N_DRIVERS == 14
    assert re.search(r'#define MOD16_N_PARAMS 11', text) and lib.N_PARAMS == 11
    assert re.search(r'#define MOD16_N_CLASSES 13', text) and lib.N_CLASSES == 13


def test_no_device_fails_loudly(lib):
    """Without an MI355X the product path raises; it never computes on the CPU."""
    if lib.device_count() > 0:
        pytest.skip('a GPU is present')
    handle = ctypes.c_void_p()
    assert lib.load().mod16_create(0, ctypes.byref(handle)) == lib.ERR_NO_DEVICE
    assert not handle.value
    import mod16_amd
    model = mod16_amd.MOD16(dict.fromkeys(mod16_amd.MOD16.required_parameters, 1.0))
    with pytest.raises(lib.Mod16Error, match='no CPU fallback'):
        model.evapotranspiration(*([1.0] * 14))
    # the calibration interface likewise: one vector, batched, bound
    with pytest.raises(lib.Mod16Error, match='no CPU fallback'):
        mod16_amd.MOD16._et([1.0] * 11, *([1.0] * 14))
    with pytest.raises(lib.Mod16Error, match='no CPU fallback'):
        mod16_amd.MOD16._et_bind(*([1.0] * 14), observed=1.0)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under mod16_amd/ (the product) or
    tools/ (timing and profiling scripts) may import it; its only users are
    tests/, __graft_entry__.smoke() and bench.py's cpu_baseline / parity legs."""
    for sub in ('mod16_amd', 'tools'):
        for dirpath, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith(('.py', '.hip', '.hpp', '.h')):
                    text = open(os.path.join(dirpath, f)).read()
                    assert not re.search(r'^\s*(from|import)\s+oracle\b', text, flags=re.M), f


def test_no_kernel_spills_vector_registers(tmp_path):
    """hipcc 7.2 miscompiled a float32 kernel that needed ~400 vector registers
    and spilled inside divergent code (wrong values in a few percent of the
    pixels; tests/test_gpu_stream.py found it). No kernel of the library may
    spill vector registers -- checked on the gfx950 listing."""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    src = os.path.join(ROOT, 'mod16_amd', 'csrc', 'mod16_capi.hip')
    out = str(tmp_path / 'capi.s')
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S',
                           '--cuda-device-only', '-o', out, src],
                          stderr=subprocess.DEVNULL)
    text = open(out).read()
    kernels = re.findall(r'\.name:\s+(\S+)\n(?:.*\n){0,12}?\s*\.vgpr_spill_count:\s*(\d+)', text)
    assert len(kernels) > 50
    spilling = [name for name, count in kernels if int(count) > 0]
    assert not spilling, spilling
    # the pipeline kernel hides HBM latency behind a second wave per SIMD: every instance must fit two
    # (unified register file of 512 per lane: at most 256 each, accumulator registers included) -- round 5's
    # raw-driver instances with per-pixel hours had grown to 258 / 260, i.e. ONE wave
    regs = re.findall(r'\.name:\s+(\S*et_stream_kernel\S*)\n(?:.*\n){0,14}?\s*\.vgpr_count:\s*(\d+)', text)
    assert len(regs) >= 40, len(regs)
    fat = [(name, int(count)) for name, count in regs if int(count) > 256]
    assert not fat, fat
    # ... and none may touch scratch memory or park registers in the accumulator file
    # (how hipcc spilled that kernel: 183 v_accvgpr_write / 263 v_accvgpr_read, no
    # scratch): the stream kernels count their own vector-memory operations
    # (s_waitcnt vmcnt(NOUT)), which compiler-issued scratch traffic would break
    bodies = re.split(r'\n(_Z[^\n:]*):[^\n]*\n', text)
    assert len(bodies) > 100
    for name, body in zip(bodies[1::2], bodies[2::2]):
        body = body.split('.section')[0]
        assert not re.search(r'^\s*scratch_(load|store)', body, flags=re.M), name
        # (the kernels that revisit flagged pixels behind the pipeline -- reference-order
        # arithmetic of the raw-driver forms, three of them need 258 registers -- may park a
        # few; they hold no counted memory operations and run only for pixels outside the
        # domain of the production arithmetic, where tests/test_gpu_parity.py checks them
        # against the oracle value by value)
        if 'et_stream_redo_kernel' in name:
            assert len(re.findall(r'^\s*v_accvgpr_(read|write)', body, flags=re.M)) <= 16, name
            continue
        assert not re.search(r'^\s*v_accvgpr_(read|write)', body, flags=re.M), name


def test_form_table_matches_the_library():
    """_lib.FORM_SHAPE (what TiledRaster allocates for a form) against mod16_form_shape()
    (what mod16_et_form_tiled_* reads): same counts of wide arrays, byte rasters, outputs."""
    import ctypes as C
    from mod16_amd import _lib
    lib = _lib.load()
    nw, nb, no = C.c_int(), C.c_int(), C.c_int()
    for form, shape in _lib.FORM_SHAPE.items():
        assert lib.mod16_form_shape(form, C.byref(nw), C.byref(nb), C.byref(no)) == 0
        assert (nw.value, nb.value, no.value) == shape, form
    assert lib.mod16_form_shape(len(_lib.FORM_SHAPE), None, None, None) != 0


def test_shipped_library_reads_no_experiment_switch(lib):
    """Launch-geometry overrides and measurement paths exist only in -DMOD16_EXPERIMENTS builds
    (libmod16hip_exp.so, tools/): the shipped library names TWO environment variables, the
    documented knobs of its HOST mode (MOD16_HOST_THREADS, MOD16_SMALL_PIXELS), and the sources
    refuse to compile a measurement switch without the experiments define."""
    import shutil
    import subprocess

    def env_names(path):
        blob = open(path, 'rb').read()
        return set(m.decode() for m in re.findall(rb'MOD16_[A-Z0-9_]{3,}', blob))
    switches = {'MOD16_NO_DMA', 'MOD16_RUN_SHIFT', 'MOD16_STATIC_BELOW', 'MOD16_STREAM_BLOCKS',
                'MOD16_PITCH', 'MOD16_GRID_MULT', 'MOD16_POISON_TICKET', 'MOD16_POISON_BYTE'}
    shipped = env_names(lib.LIB_PATH)      # (the rest are enum names inside error messages)
    assert {'MOD16_HOST_THREADS', 'MOD16_SMALL_PIXELS'} <= shipped and not (shipped & switches), shipped
    assert not any(n.startswith(('MOD16_NO_', 'MOD16_EXPERIMENT', 'MOD16_TRIVIAL', 'MOD16_PRIO', 'MOD16_DYN'))
                   for n in shipped), shipped
    assert switches | {'MOD16_HOST_THREADS'} <= env_names(lib.EXP_LIB_PATH)
    capi = os.path.join(ROOT, 'mod16_amd', 'csrc', 'capi')       # the library's host side, every family of it
    src = ''.join(open(os.path.join(capi, f)).read() for f in sorted(os.listdir(capi)))
    product = re.sub(r'#ifdef MOD16_EXPERIMENTS.*?#endif', '', src, flags=re.S)
    assert re.findall(r'getenv\("(\w+)"\)', product) == ['MOD16_HOST_THREADS', 'MOD16_SMALL_PIXELS']
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if os.path.exists(hipcc):
        for switch in ('-DMOD16_NO_GUARD', '-DMOD16_TRIVIAL_BODY', '-DMOD16_DYN_RUN=4'):
            proc = subprocess.run([hipcc, '--offload-arch=gfx950', '-std=c++17', '-fsyntax-only', '--cuda-device-only',
                                   '-x', 'hip', switch, os.path.join(ROOT, 'mod16_amd', 'csrc', 'mod16_math.hpp')],
                                  capture_output=True, text=True)
            assert proc.returncode != 0 and 'MOD16_EXPERIMENTS' in proc.stderr, switch
