"""The N > 1 step of bench.py on the ONE GPU a test box has: MOD16_BENCH_FORCE_GROUP=1 makes a
`--gpus 1` run join an RCCL process group of one rank and take the multi-GPU code path unchanged
(init_process_group('nccl', device_id=...), the side stream, the produced / reduced events,
mod16_amd.dist.allreduce_diag -> all_gather_into_tensor + the library's rank-order fold, the MAX
all-reduce of the elapsed time, kernel_ms_by_rank, the parity leg's all-reduces). SURVEY.md 8(e):
"RCCL ... only for the final global-sum/reduction diagnostics".

Each run is a fresh child process started from here -- never a re-exec of a process that has
touched the GPU -- and this file sorts first so that the children are started before the test
process itself initialises the device."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

ARGS = ['--gpus', '1', '--rows', '5400', '--steps', '60', '--warmup', '5', '--no-cpu-baseline',
        '--no-configs', '--no-plain']


def run_bench(extra_env):
    env = dict(os.environ, OMP_NUM_THREADS='4', **extra_env)
    for key in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(key, None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + ARGS, env=env, cwd=ROOT,
                          capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, proc.stdout[-2000:]
    # rank 0's stdout is the ONE JSON line and nothing else (RCCL's version banner goes to stderr)
    assert [l for l in proc.stdout.splitlines() if l.strip()] == lines, proc.stdout[:1000]
    return json.loads(lines[0])


def test_one_rank_rccl_group_runs_the_multi_gpu_step():
    plain = run_bench({})
    forced = run_bench({'MOD16_BENCH_FORCE_GROUP': '1'})
    assert plain['process_group'] == {'backend': None, 'world_size': 1, 'forced_for_one_rank': False}
    assert forced['process_group'] == {'backend': 'nccl', 'world_size': 1, 'forced_for_one_rank': True}
    assert forced['ranks_seen'] == 1 and forced['n_gpus'] == 1
    assert len(forced['roofline']['kernel_ms_by_rank']) == 1
    # the diagnostics went through the gather and the rank-order fold: the same bits
    assert forced['diagnostics'] == plain['diagnostics']
    assert forced['diagnostics']['n_valid_day'] + forced['diagnostics']['n_nan_day'] == 5400 * 43200
    # parity summaries travelled through the group's all-reduces (MAX, SUM)
    for key in ('max_rel_err', 'masks_equal'):
        assert forced['parity'][key] == plain['parity'][key]
    full = forced['parity']['full_grid_fast_vs_exact_kernel']
    assert full == plain['parity']['full_grid_fast_vs_exact_kernel'] and full['pixels'] == 5400 * 43200
    assert full['nan_masks_equal'] and full['zero_mask_mismatches'] == 0
    # the collective hides behind the next step's kernel: same step time. (Two processes one after the
    # other on one device: 3 % allows for the clock the device holds at its power cap from run to run;
    # measured 0.1 % -- profiles/r04_one_rank_rccl_group_line.json: 20.04 vs 20.06 ms on the full grid.
    # The defect this guards against -- graph replays between two barriers that did nothing -- was a
    # factor of twelve.)
    assert abs(forced['ms_per_step'] / plain['ms_per_step'] - 1.0) < 0.03, (forced['ms_per_step'], plain['ms_per_step'])
    assert abs(forced['roofline']['kernel_ms'] / plain['roofline']['kernel_ms'] - 1.0) < 0.03
    # ... and clock-normalised -- cycles_per_step = kernel_ms (HIP events around the timed steps) x the
    # shader clock the device held under this load -- the two runs agree to 1.5 %: what a collective
    # that did NOT hide behind the kernel, or a step that had grown, would show beyond the device's
    # clock wander. (Needs the hwmon sensors; a box without them keeps the 3 % check only.)
    cyc_f, cyc_p = forced['roofline'].get('cycles_per_step'), plain['roofline'].get('cycles_per_step')
    if cyc_f and cyc_p:
        assert abs(cyc_f / cyc_p - 1.0) < 0.015, (cyc_f, cyc_p, forced['roofline']['sclk_mhz'], plain['roofline']['sclk_mhz'])
        assert forced['summary']['cycles_per_step'] == cyc_f
    # rank 0's stdout is the one JSON line (RCCL's banner goes to stderr)
    assert forced['process_group']['backend'] == 'nccl'
