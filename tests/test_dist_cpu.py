"""Tile sharding over ranks, on the CPU: band arithmetic, and a world_size-2
gloo run in which each rank computes its row band (with the oracle) and the
diagnostics vectors are all-reduced -- the N > 1 structure of bench.py."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from mod16_amd import dist as tiles


@pytest.mark.parametrize('rows,world', [(21600, 1), (21600, 2), (21600, 8), (10, 3), (7, 8)])
def test_bands_tile_the_raster(rows, world):
    edges = [tiles.band(rows, r, world) for r in range(world)]
    assert edges[0][0] == 0 and edges[-1][1] == rows
    for (a0, a1), (b0, b1) in zip(edges, edges[1:]):
        assert a1 == b0 and a0 <= a1
    sizes = [b - a for a, b in edges]
    assert max(sizes) - min(sizes) <= 1
    off = [tiles.pixel_range(rows, 43200, r, world) for r in range(world)]
    assert sum(n for _, n in off) == rows * 43200
    assert all(o == e[0] * 43200 for (o, _), e in zip(off, edges))
    with pytest.raises(ValueError):
        tiles.band(rows, world, world)


def test_global_grid_bands_are_vector_aligned():
    """Every band of the 43200 x 21600 grid starts on a 16-byte boundary and
    has an even pixel count for 1, 2, 4 and 8 GPUs (the fused path needs it)."""
    for world in (1, 2, 4, 8):
        for r in range(world):
            off, n = tiles.pixel_range(tiles.GLOBAL_ROWS, tiles.GLOBAL_COLS, r, world)
            assert off % 2 == 0 and n % 4 == 0
            assert n == tiles.GLOBAL_ROWS // world * tiles.GLOBAL_COLS


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_two_rank_gloo_bands_and_allreduce(tmp_path):
    out = tmp_path / 'result.json'
    env = dict(os.environ, MOD16_DIST_OUT=str(out), OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()),
           os.path.join(ROOT, 'tests', 'dist_worker.py')]
    proc = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    res = json.loads(out.read_text())
    assert res['world'] == 2
    np.testing.assert_allclose(res['reduced'][:2], res['global'][:2], rtol=1e-12)
    assert res['reduced'][2:] == res['global'][2:]
    assert res['bands_identical_to_global']


def test_bench_rank_environment():
    """`python bench.py --gpus N` starts its own ranks: what each child finds in its environment
    (the names torch.distributed.run exports, 127.0.0.1 for the rendezvous)."""
    sys.path.insert(0, ROOT)
    import bench
    env = bench.rank_env({'PATH': '/bin'}, 5, 8, 29511)
    assert env['RANK'] == '5' and env['LOCAL_RANK'] == '5' and env['WORLD_SIZE'] == '8'
    assert env['MASTER_ADDR'] == '127.0.0.1' and env['MASTER_PORT'] == '29511'
    assert env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and env['PATH'] == '/bin'


@pytest.mark.parametrize('ranks', [2, 8])
def test_bench_gpus_n_spawns_its_own_ranks(ranks):
    """`python bench.py --gpus N` typed as is (no torch.distributed.run around it), N = 2 and
    the 8 of a full node: the parent launches the ranks, the ranks count each other with a
    collective, reduce a diagnostics vector with the product's one-gather reduction and rank
    0's line comes back through the parent, whose exit code is the ranks'.
    MOD16_BENCH_PLUMBING=1 replaces the GPU work of a rank by nothing (no GPU in this
    container); launcher, rendezvous, collective and relay are the real ones."""
    env = dict(os.environ, MOD16_BENCH_PLUMBING='1', OMP_NUM_THREADS='1')
    for key in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(key, None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(ranks),
                           '--steps', '3', '--warmup', '1'],
                          env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, proc.stdout
    res = json.loads(lines[0])
    assert res['n_gpus'] == ranks and res['ranks_seen'] == ranks and res['steps'] == 3
    # rank r contributed [r, 1, ..., r]: sums in rank order, maxima over ranks
    assert res['diag_reduced'][0] == sum(range(ranks)) and res['diag_reduced'][1] == ranks
    assert res['diag_reduced'][6] == ranks - 1


def test_bench_under_torch_distributed_run():
    """The driver's launch line (python -m torch.distributed.run ... bench.py --gpus 2): the
    process is one of the ranks and does not spawn."""
    env = dict(os.environ, MOD16_BENCH_PLUMBING='1', OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()),
           os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1']
    proc = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1 and json.loads(lines[0])['ranks_seen'] == 2


def test_bench_failing_rank_fails_the_parent():
    """Ranks that die (here: the one-device rehearsal without a device) end the parent non-zero."""
    env = dict(os.environ, OMP_NUM_THREADS='1')
    for key in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(key, None)
    env['MOD16_BENCH_PLUMBING'] = '0'
    env['MOD16_BENCH_ONE_DEVICE'] = '1'
    env['CUDA_VISIBLE_DEVICES'] = env['HIP_VISIBLE_DEVICES'] = ''
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2',
                           '--no-cpu-baseline'], env=env, cwd=ROOT, capture_output=True,
                          text=True, timeout=300)
    assert proc.returncode != 0
    assert 'rank exit codes' in proc.stderr


@pytest.mark.parametrize('how', ['parent', 'rank'])
def test_bench_with_fewer_devices_than_ranks_says_so(how):
    """`bench.py --gpus N` on a node that shows fewer than N GPUs (this container shows none): one
    clear line and a non-zero exit -- from the parent before it starts a rank, and from a rank
    started by torch.distributed.run (the driver's launch line) before it touches a device. Fresh
    child processes only; nothing here initialises a GPU."""
    env = dict(os.environ, OMP_NUM_THREADS='1')
    for key in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MOD16_BENCH_PLUMBING', 'MOD16_BENCH_ONE_DEVICE'):
        env.pop(key, None)
    env['CUDA_VISIBLE_DEVICES'] = env['HIP_VISIBLE_DEVICES'] = ''
    if how == 'rank':
        env.update(RANK='1', LOCAL_RANK='1', WORLD_SIZE='4', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()))
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--no-cpu-baseline'],
                          env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert proc.returncode == 2, (proc.returncode, proc.stderr[-1000:])
    said = [l for l in proc.stderr.splitlines() if l.startswith('bench.py:')]
    assert len(said) == 1 and '--gpus 4 needs 4 visible GPUs, this node shows 0' in said[0], proc.stderr[-1000:]
    assert not [l for l in proc.stdout.splitlines() if l.startswith('{')]


def test_bench_sensors_are_optional():
    """The clock / power leg of the bench line reads hwmon files of the process's own card; a
    box without a device (this one) or without the files gets None, not an exception."""
    sys.path.insert(0, ROOT)
    import bench
    import torch
    if not torch.cuda.is_available():
        assert bench.device_sensors(torch) is None
