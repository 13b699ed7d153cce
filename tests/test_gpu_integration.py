"""The boundary documents, executed: the ctypes stub INTEGRATION.md section 2 prints for a
maintainer of the reference (cut out of the markdown and run verbatim against the reference's
own vectors), and the `mod16` alias package, through which the reference's import lines
(tests/tests.py:8, mod16/__init__.py:124) reach the GPU build unedited."""
import os
import re
import types

import numpy as np
import pytest

from conftest import ROOT
from oracle import mod16_oracle as oracle
from parity import assert_parity

SEP = ('canopy_day', 'soil_day', 'trans_day', 'canopy_night', 'soil_night', 'trans_night')


def stub_source():
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    section = text[text.index('## 2. Bind the C ABI inside the reference'):]
    blocks = re.findall(r'```python\n(.*?)```', section, flags=re.S)
    assert blocks and blocks[0].startswith('# mod16/_hip.py'), 'INTEGRATION.md section 2 lost its stub'
    return blocks[0]


def test_stub_is_where_the_document_says():
    src = stub_source()
    assert "C.CDLL('libmod16hip.so')" in src and 'def evapotranspiration(model, drivers, separate=False)' in src
    compile(src, 'INTEGRATION.md#stub', 'exec')


@pytest.mark.gpu
def test_integration_stub_runs_verbatim(golden):
    from mod16_amd import _lib
    _lib.load()          # maps libmod16hip.so (SONAME): the stub's CDLL('libmod16hip.so') finds it
    hip = types.ModuleType('mod16._hip')
    exec(compile(stub_source(), 'INTEGRATION.md#stub', 'exec'), hip.__dict__)

    class Model:         # what the stub reads of a reference MOD16 instance
        required_parameters = list(oracle.PARAM_NAMES)

    f1 = golden('f1_tests_scalars')
    m = Model()
    for k, v in zip(oracle.PARAM_NAMES, f1['params']):
        setattr(m, k, float(v))
    day, night = hip.evapotranspiration(m, [float(v) for v in f1['drivers']])
    assert_parity(np.asarray(day), f1['day'], 1e-9, 'F1 day')
    assert_parity(np.asarray(night), f1['night'], 1e-9, 'F1 night')
    sep = hip.evapotranspiration(m, [float(v) for v in f1['drivers']], separate=True)
    for name, got in zip(SEP, list(sep[0]) + list(sep[1])):
        assert_parity(np.asarray(got), f1[name], 1e-9, 'F1 ' + name)

    # F3: the reference's per-pixel parameter idiom, params_dict[key][pft_map] (notebook cell 32)
    f3 = golden('f3_random64_f64')
    m3 = Model()
    for j, k in enumerate(oracle.PARAM_NAMES):
        setattr(m3, k, f3['table'][:, j][f3['cls']])
    day, night = hip.evapotranspiration(m3, list(f3['drivers']))
    assert day.shape == (64, 64)
    assert_parity(day, f3['day'], 1e-9, 'F3 day')
    assert_parity(night, f3['night'], 1e-9, 'F3 night')
    sep = hip.evapotranspiration(m3, list(f3['drivers']), separate=True)
    for name, got in zip(SEP, list(sep[0]) + list(sep[1])):
        assert_parity(got, f3[name], 1e-9, 'F3 ' + name)


def test_reference_import_lines_resolve_to_the_gpu_build():
    """tests/tests.py:8 of the reference, and the other two modules of its forward-run surface."""
    from mod16 import MOD16, psychrometric_constant, radiation_net, svp_slope, latent_heat_vaporization
    from mod16.models import MOD16Collection61, PFT_ALL
    from mod16.utils import restore_bplut, BPLUT_FIELD_LOOKUP
    import mod16
    import mod16_amd
    assert MOD16 is mod16_amd.MOD16 and MOD16Collection61 is mod16_amd.models.MOD16Collection61
    assert restore_bplut is mod16_amd.utils.restore_bplut and len(PFT_ALL) and len(BPLUT_FIELD_LOOKUP) == 11
    assert mod16.PFT_VALID == (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12)
    assert mod16.STEFAN_BOLTZMANN == 5.67e-8 and mod16.SPECIFIC_HEAT_CAPACITY_AIR == 1013
    for fn in (psychrometric_constant, radiation_net, svp_slope, latent_heat_vaporization):
        assert fn.__module__ == 'mod16_amd'
    with pytest.raises(KeyError):
        MOD16({'tmin_close': 1.0})


@pytest.mark.gpu
def test_reference_test_preamble_through_the_alias(golden):
    """tests/tests.py:64-90 (test_et_vectorized) typed with the reference's import line."""
    from mod16 import MOD16, latent_heat_vaporization
    f1 = golden('f1_tests_scalars')
    model = MOD16(dict(zip(oracle.PARAM_NAMES, [float(v) for v in f1['params']])))
    day, night = model.evapotranspiration(*[float(v) for v in f1['drivers']])
    assert_parity(np.asarray(day), f1['day'], 1e-9, 'day')
    et = MOD16._et([float(v) for v in f1['params']], *[float(v) for v in f1['drivers']])
    assert abs(float(et) - float(f1['et_static'])) <= 1e-9 * abs(float(f1['et_static']))
    total = day * latent_heat_vaporization(293) + night * latent_heat_vaporization(290)
    assert round(float(total), 1) == 41.0
