"""SURVEY.md section 5 (sanitizers): the HOST half of libmod16hip -- its own source, compiled with
`hipcc --cuda-host-only -fsanitize=address,undefined` -- linked against a stand-in for the HIP
runtime (tests/host_asan/hip_stub.hip: device memory = host heap under AddressSanitizer, kernel
launches = shadows that replay the launch's address arithmetic against the allocation table) and
driven through the C ABI over ragged sizes, every form and layout, bad layouts, the global grid,
2^31 + 12344 pixels, graphs, the HOST-mode tiler and the resident calibration problem. Clean = no
sanitizer report, no address outside an allocation, nothing leaked; and a planted fault (a raster one
tile short) is reported. Sanitizers run on the CPU build only (the GPU pool refuses them)."""
import os
import subprocess

from conftest import ROOT


def test_host_half_of_the_library_is_clean_under_asan_and_ubsan(tmp_path):
    script = os.path.join(ROOT, 'tests', 'host_asan', 'build_and_run.sh')
    proc = subprocess.run(['bash', script, str(tmp_path)], capture_output=True, text=True, timeout=900)
    out = proc.stdout + proc.stderr
    assert proc.returncode == 0, out[-4000:]
    assert 'host_asan: ok' in out, out[-2000:]
    assert 'ERROR: AddressSanitizer' not in out and 'runtime error:' not in out and 'LeakSanitizer' not in out, out[-4000:]
    assert 'planted fault detected' in out, out[-2000:]
    # the shadows that carry the launch geometry ran, on both data types and on the big rasters
    assert 'et_stream_kernel' in out and 'et_stream_redo_kernel' in out and 'et_kernel' in out
    assert 'graph lifetime: done' in out
    for what in ('tiled rasters, float64', 'tiled rasters, float32', 'plain device arrays, float64',
                 'HOST mode, float64', 'HOST mode, float32'):
        assert what in out, what
