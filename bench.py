#!/usr/bin/env python3
"""
bench.py -- the MOD16 forward-run benchmark (BASELINE.json metric: pixels/s and
achieved HBM GB/s on the 43200 x 21600 global ET grid, float64).

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no torch.distributed environment the command starts its own
ranks: the parent -- which never touches a GPU -- starts the N ranks directly as
child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 /
MASTER_PORT in their environment; not through torch.distributed.run, whose
elastic agent holds the GPU open as one process more), relays rank 0's JSON
line and returns non-zero if any rank failed. Launched under
torch.distributed.run (RANK / WORLD_SIZE set: the driver's line) it is one of
the ranks. MOD16_BENCH_FORCE_GROUP=1 makes a --gpus 1 run take the N > 1 code
path unchanged -- an RCCL process group of one rank, the side stream, the event
hand-off and the all-gather of the diagnostics vector -- so that the collective
path runs on a one-GPU box (tests/test_a_gpu_group.py).

One "step" is one pass of the hot path over the (synthetic, already
HBM-resident) drivers of one time step: the fused ET kernel over this rank's row
band, which also reduces its outputs to the diagnostics vector (deterministic
two-level sum), and -- for N > 1 -- the RCCL all-reduce of that 8-double vector,
overlapped with the next step. The global grid is fixed and cut into N row bands
(one process per GPU), so total work is fixed: "scaling": "strong".

The raster is resident in the engine's tiled layout (mod16_layout: the 14 driver
arrays interleaved tile by tile); the same kernel on 16
plain arrays -- the layout the reference's arguments have -- is timed beside it
(`roofline.plain_arrays`).

Rank 0 prints ONE JSON line. Besides the contract fields it carries
  roofline      dominant kernel (fused ET) vs the 8 TB/s HBM peak; `achieved`
                = 129 B/pixel (float64) x pixels per launch / mean launch time,
                timed with HIP events on the launch stream; `frac_of_measured_copy`
                relates it to a copy kernel measured in this process;
  cpu_baseline  the numpy oracle (reference-shaped port) timed on this box's
                host cores on 1200 x 1200 tiles of the same synthetic workload
                (N = 1 only);
  parity        GPU outputs vs the oracle on 1200 x 1200 windows copied back,
                and the production kernel vs the reference-order kernel on
                every pixel;
  configs       (N = 1) the other BASELINE.json configurations, measured in the
                same run: one 1200 x 1200 tile, the 46-step series, float32.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E peak (MI355X_MICROARCH.md)
TILE = (1200, 1200)           # BASELINE.json configs[1], the CPU sample unit
SEED = 16
DIAG_NAMES = ('sum_day', 'sum_night', 'n_valid_day', 'n_valid_night',
              'n_nan_day', 'n_nan_night', 'max_day', 'max_night')


# ----------------------------------------------------------------- launcher
def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def rank_env(base, rank, gpus, port):
    """Environment of rank `rank` of an N-rank run on this node (what
    torch.distributed.run would export)."""
    env = dict(base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(gpus), LOCAL_WORLD_SIZE=str(gpus),
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    return env


def too_few_devices(gpus):
    """One line of text if this node shows fewer GPUs than the run has ranks (None otherwise, and
    for the rehearsals that need none or one: MOD16_BENCH_PLUMBING, MOD16_BENCH_ONE_DEVICE).
    Counting devices does not initialise the GPU (the parent of a multi-GPU run never does)."""
    if os.environ.get('MOD16_BENCH_PLUMBING') == '1' or os.environ.get('MOD16_BENCH_ONE_DEVICE') == '1':
        return None
    import torch
    visible = torch.cuda.device_count()
    if visible >= gpus:
        return None
    return ('bench.py: --gpus %d needs %d visible GPUs, this node shows %d (one rank per GPU; '
            'MOD16_BENCH_ONE_DEVICE=1 rehearses the ranks on one device over gloo)' % (gpus, gpus, visible))


def spawn(argv, gpus):
    """Parent of a multi-GPU run started as plain `python bench.py --gpus N`: starts the N
    ranks itself as child processes (never an exec: this process must stay clear of the GPU
    and simply waits), relays the one JSON line of rank 0 and returns non-zero if any rank
    failed. The ranks are started directly, not through torch.distributed.run: its elastic
    agent holds the GPU open as well (/dev/kfd, seen with MOD16_BENCH_CENSUS=1), which is one
    GPU process more than the ranks -- on the one-GPU boxes, which allow six, that silently
    ended the 6-rank rehearsal of round 2. Under torch.distributed.run
    (the driver's launch line) this function is not involved: the process is a rank."""
    short = too_few_devices(gpus)
    if short:
        sys.stderr.write(short + '\n')
        return 2
    port = free_port()
    script = os.path.join(ROOT, 'bench.py')
    procs = []
    for r in range(gpus):
        procs.append(subprocess.Popen(
            [sys.executable, '-u', script] + list(argv), cwd=ROOT, env=rank_env(os.environ, r, gpus, port),
            stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    import threading
    lines = []

    def relay():
        for out in procs[0].stdout:
            text = out.strip()
            if text.startswith('{') and '"metric"' in text:
                lines.append(text)
            else:
                sys.stderr.write(out)

    reader = threading.Thread(target=relay, daemon=True)
    reader.start()
    # wait for all ranks; once one has failed the others cannot finish a collective: give them
    # 30 s, then end them (these very children, by their handles)
    deadline = None
    while any(proc.poll() is None for proc in procs):
        if deadline is None and any(proc.poll() not in (None, 0) for proc in procs):
            deadline = time.monotonic() + 30
        if deadline is not None and time.monotonic() > deadline:
            for proc in procs:
                if proc.poll() is None:
                    proc.kill()
        time.sleep(0.1)
    codes = [proc.wait() for proc in procs]
    reader.join(timeout=10)
    line = lines[-1] if lines else None
    rc = next((c for c in codes if c != 0), 0)
    if rc != 0:
        sys.stderr.write('bench.py: rank exit codes %s\n' % codes)
    if line is not None and rc == 0:
        print(line, flush=True)
    elif rc == 0:
        sys.stderr.write('bench.py: the ranks exited without a result line\n')
        rc = 1
    return rc


def gpu_process_census():
    """Processes of this user that hold the GPU open (/dev/kfd), by command: the GPU boxes
    allow 6 at a time (a 7th ends the whole run without a message), and under
    torch.distributed.run the elastic agent can be one of them beside the ranks."""
    found = []
    for pid in os.listdir('/proc'):
        if not pid.isdigit():
            continue
        try:
            fds = os.listdir('/proc/%s/fd' % pid)
            if any(os.readlink('/proc/%s/fd/%s' % (pid, fd)) == '/dev/kfd' for fd in fds):
                with open('/proc/%s/cmdline' % pid, 'rb') as f:
                    cmd = f.read().replace(b'\0', b' ').decode(errors='replace').strip()
                found.append({'pid': int(pid), 'cmd': cmd[:120]})
        except OSError:
            continue
    return found


# -------------------------------------------------------------- CPU baseline
def cpu_tile_seconds(reps):
    """Worker of the CPU baseline: time `reps` oracle runs on one tile."""
    from oracle import mod16_oracle as oracle
    from oracle import synth
    from mod16_amd.utils import restore_bplut, bplut_table
    from mod16_amd.models import COLLECTION61_BPLUT
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    cls, drv = synth.drivers(TILE, seed=SEED + os.getpid() % 7)
    best = 1e30
    t_all = time.perf_counter()
    for _ in range(reps):
        t0 = time.perf_counter()
        oracle.evapotranspiration_raster(bplut, cls, *drv)
        best = min(best, time.perf_counter() - t0)
    return best, time.perf_counter() - t_all


def cpu_baseline(max_workers):
    """The oracle on host cores: one process (numpy's element-wise loops are
    single-threaded), then a pool with one tile per worker. Runs before this
    process touches the GPU; workers are spawned, never forked."""
    import multiprocessing as mp
    tile_px = TILE[0] * TILE[1]
    best1, _ = cpu_tile_seconds(3)
    cores = max(1, min(max_workers, os.cpu_count() or 1))
    out = {
        'value': tile_px / best1, 'unit': 'pixels/s', 'cores': 1, 'kind': 'port',
        'sample': '3 runs of a 1200x1200 float64 tile (best), numpy oracle incl. per-pixel BPLUT gather',
        'seconds_per_tile': best1,
    }
    if cores > 1:
        reps = 2
        with mp.get_context('spawn').Pool(cores) as pool:
            t0 = time.perf_counter()
            res = pool.map(cpu_tile_seconds, [reps] * cores)
            wall = time.perf_counter() - t0
        busy = max(r[1] for r in res)       # excludes interpreter start-up
        out['pool'] = {'value': cores * reps * tile_px / busy, 'cores': cores,
                       'sample': '%d workers x %d tiles of 1200x1200' % (cores, reps),
                       'wall_s': wall}
    # SURVEY 8d form (2): the same totals as fused, strength-reduced numpy (oracle/fused_numpy.py)
    from oracle import fused_numpy, mod16_oracle as oracle, synth
    from mod16_amd.utils import restore_bplut, bplut_table
    from mod16_amd.models import COLLECTION61_BPLUT
    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    cls, drv = synth.drivers(TILE, seed=SEED)
    best2 = 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        fused_numpy.evapotranspiration_raster(bplut, cls, *drv)
        best2 = min(best2, time.perf_counter() - t0)
    out['fused_numpy'] = {'value': tile_px / best2, 'unit': 'pixels/s', 'cores': 1, 'seconds_per_tile': best2,
                          'sample': '3 runs of a 1200x1200 float64 tile (best), shared per-period terms, '
                                    'conductance forms, one division per component'}
    # the same as flat scalars (the driver's record keeps scalars of this object)
    out['pool_px_s'] = _dig(out, 'pool', 'value')
    out['pool_cores'] = _dig(out, 'pool', 'cores')
    out['fused_px_s'] = out['fused_numpy']['value']
    out['global_grid_seconds_1core'] = 43200 * 21600 / out['value']
    try:
        with open('/proc/cpuinfo') as f:
            models = [l.split(':', 1)[1].strip() for l in f if l.startswith('model name')]
        out['cpu_model'] = models[0] if models else None
        out['host_logical_cpus'] = os.cpu_count()
    except OSError:
        pass
    return out


def device_sensors(torch):
    """Directory of the hwmon files (power1_input in uW, power1_cap, freq1_input = shader clock
    in Hz; read-only) of the card THIS process computes on, found by its PCI address -- a box
    shows the sensors of all the GPUs of its host."""
    import ctypes
    import glob
    try:
        # the HIP runtime this process already uses (PyTorch-ROCm wheels bundle their own copy:
        # a second runtime in the process could not open the GPU)
        own = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')
        hip = ctypes.CDLL(own if os.path.exists(own) else 'libamdhip64.so')
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, torch.cuda.current_device()) != 0:
            return None
        bus = buf.value.decode().lower()
    except (OSError, AttributeError, RuntimeError):       # no HIP runtime / no device: no sensors
        return None
    for path in glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_input'):
        if bus in os.path.realpath(path.split('/hwmon')[0]).lower():
            return os.path.dirname(path)
    return None


def device_under_load(torch, step, nsteps, sample):
    """Shader clock and package power while the timed step runs, OUTSIDE the timed region:
    `nsteps` more steps are enqueued and the sensors read until they are done. Every rank runs
    the steps (they hold the collective); `sample` says whether this rank reads the sensors. Why
    it is in the line: the float64 step runs the chip into its package power cap and the shader
    clock comes down to ~1.7 GHz (profiles/r03_power_probe.jsonl) -- the step
    time follows that clock, and it differs from device to device.

    Order (round 5): the steps are enqueued FIRST and the watcher thread opens the sensor files
    only once 40 % of them have run, so the load is steady when the first read arrives. Round 4
    started the thread in front of the steps; its rocprofv3 trace has a 3.4 ms idle gap in front of
    the first extra launch (eng.check(), reading the step times back, starting the thread) and the
    two launches behind it at 21.4 and 27.4 ms against 20.1: with the reads
    moved away from the gap the next trace tells the two suspects apart."""
    hw = device_sensors(torch) if sample else None

    def read(name):
        try:
            with open(os.path.join(hw, name)) as f:
                return float(f.read().split()[0])
        except (OSError, ValueError, IndexError):
            return None
    clocks, stop = [], threading.Event()
    t0 = time.perf_counter()
    lead = max(1, int(0.4 * nsteps))
    ready = torch.cuda.Event()

    def watch():        # a thread: with N > 1 a step may block in its collective
        ready.synchronize()             # the first `lead` steps are done: the load is steady
        while not stop.is_set():
            c, w = read('freq1_input'), read('power1_input')
            if c is not None and w is not None:
                clocks.append((time.perf_counter() - t0, c * 1e-6, w * 1e-6))
            time.sleep(0.02)
    for k in range(nsteps):
        step()
        if k + 1 == lead:
            ready.record()
    watcher = threading.Thread(target=watch, daemon=True) if hw is not None else None
    if watcher is not None:
        watcher.start()
    torch.cuda.synchronize()
    stop.set()
    if watcher is None:
        return None
    watcher.join()
    total = time.perf_counter() - t0
    kept = clocks
    if not kept:
        return None
    cap = read('power1_cap')
    return {'sclk_mhz': sum(x[1] for x in kept) / len(kept), 'sclk_mhz_min': min(x[1] for x in kept),
            'sclk_mhz_max': max(x[1] for x in kept), 'power_w': sum(x[2] for x in kept) / len(kept),
            'power_cap_w': cap * 1e-6 if cap else None, 'samples': len(kept), 'seconds': total,
            'first_sample_s': kept[0][0], 'source': hw,
            'note': 'hwmon sensors of this device read while %d more steps run behind the timed region '
                    '(first read after %d of them); a float64 step at the power cap runs at the clock the cap leaves'
                    % (nsteps, lead)}


def pmc_traffic(pixels_per_launch, dtype, layout, build_id, profiles_dir=None):
    """(HBM bytes per launch, source file, note) of the dominant kernel from the committed
    rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, collected separately by
    tools/run_profiles.sh, see the files). Counters cannot be read from inside this process,
    so a record applies only when it was measured on THIS build of the library (its
    `build_id`, the digest of sources and flags that mod16_build_id() returns) launching the
    same kernel on the same pixel count and layout; otherwise the figure is None and the note
    says why -- a kernel change without fresh PMC passes never prints a stale number."""
    import glob
    profiles_dir = profiles_dir or os.path.join(ROOT, 'profiles')
    shape_only = None
    for path in sorted(glob.glob(os.path.join(profiles_dir, '*pmc_hbm_traffic*.json')), reverse=True):
        try:
            with open(path) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        if rec.get('pixels_per_launch') == pixels_per_launch and rec.get('dtype') == dtype \
                and rec.get('layout', 'plain') == layout:
            if rec.get('build_id') == build_id:
                return rec['traffic_bytes_per_launch'], os.path.relpath(path, ROOT), \
                    'PMC passes of build %s (git %s)' % (build_id, rec.get('git_commit'))
            shape_only = shape_only or (os.path.relpath(path, ROOT), rec.get('build_id'))
    if shape_only:
        return None, None, ('no PMC record of this build (%s): the newest record of this shape, %s, was measured '
                            'on build %s -- run tools/run_profiles.sh' % (build_id, shape_only[0], shape_only[1]))
    return None, None, 'no PMC record for this pixel count / dtype / layout'


# ------------------------------------------------------------------ the rank
def plumbing_rank(args, rank, world):
    """MOD16_BENCH_PLUMBING=1 (tests/test_dist_cpu.py): the launcher, the
    rendezvous and the result line without a GPU -- every rank joins a gloo
    group, the ranks are counted with an all-reduce and rank 0 prints a line
    whose measurements are null."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from mod16_amd import dist as tiles
    seen = torch.ones(1, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(seen)
    # the product's reduction of the diagnostics vector (one gather, rank order)
    diag = torch.tensor([rank, 1, 0, 0, 0, 0, rank, -rank], dtype=torch.float64)
    tiles.allreduce_diag(diag)
    if rank == 0:
        print(json.dumps({'metric': 'plumbing rehearsal, no measurement', 'value': None,
                          'n_gpus': world, 'ranks_seen': int(seen.item()),
                          'diag_reduced': diag.tolist(),
                          'steps': args.steps, 'warmup': args.warmup}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def compare_on_device(torch, got, ref, thresholds):
    """Masks and relative error of two device tensors; bounded temporaries."""
    res = {'nan_masks_equal': True, 'zero_mask_mismatches': 0, 'max_rel_err': 0.0,
           'max_abs_err_over_max_value': 0.0}
    res.update({'n_rel_err_gt_' + t: 0 for t in thresholds})
    step = 1 << 27
    scale = 0.0
    for lo in range(0, ref.numel(), step):
        scale = max(scale, float(torch.nan_to_num(ref[lo:lo + step]).abs().max()))
    for lo in range(0, ref.numel(), step):
        a, b = got[lo:lo + step], ref[lo:lo + step]
        res['nan_masks_equal'] &= bool(torch.equal(torch.isnan(a), torch.isnan(b)))
        res['zero_mask_mismatches'] += int(((a == 0) != (b == 0)).sum())
        err = (a.double() - b.double()).abs_()
        res['max_abs_err_over_max_value'] = max(res['max_abs_err_over_max_value'],
                                                float(torch.nan_to_num(err).max()) / scale)
        err = torch.nan_to_num_(err.div_(b.double().abs_()), nan=0.0, posinf=0.0)
        res['max_rel_err'] = max(res['max_rel_err'], float(err.max()))
        for t in thresholds:
            res['n_rel_err_gt_' + t] += int((err > float(t)).sum())
        del err
    return res


def merge_compare(dst, src):
    dst['nan_masks_equal'] &= src['nan_masks_equal']
    for k, v in src.items():
        if k.startswith('n_') or k == 'zero_mask_mismatches':
            dst[k] += v
        elif k.startswith('max_'):
            dst[k] = max(dst[k], v)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100,
                    help='timed steps (default 100: a 2 s timed region on one MI355X)')
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--rows', type=int, default=21600, help='global raster rows')
    ap.add_argument('--cols', type=int, default=43200, help='global raster columns')
    ap.add_argument('--dtype', default='float64', choices=['float64', 'float32'])
    ap.add_argument('--math', default='fast', choices=['fast', 'mixed'],
                    help="'mixed': the mixed-precision form for --dtype float32 (configs[4])")
    ap.add_argument('--layout', default='tiled', choices=['tiled', 'plain'],
                    help="raster layout of the timed steps: the engine's tiled layout or 16 plain arrays")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true')
    ap.add_argument('--no-plain', action='store_true',
                    help='skip timing the same kernel on 16 plain arrays (profiled runs: one layout per kernel symbol)')
    ap.add_argument('--plain', action='store_true',
                    help='N > 1: time the plain-array layout as well (default: N = 1 only -- its set-up tries '
                         'seven slab spacings of up to 12 GiB of slack per rank, which a scaling run has no use for)')
    ap.add_argument('--no-sensors', action='store_true',
                    help="skip the 2 s of extra steps behind the timed region during which the device's clock and power are read")
    ap.add_argument('--no-configs', action='store_true',
                    help='skip the other BASELINE.json configurations (1200x1200 tile, series, float32)')
    ap.add_argument('--no-ingest', action='store_true',
                    help='N > 1: skip the host-ingest series every rank streams beside its band')
    ap.add_argument('--no-host-call', action='store_true',
                    help='skip the numpy-in / numpy-out call over the GPUs of the run (rank 0, devices=range(N))')
    ap.add_argument('--cpu-workers', type=int, default=16)
    ap.add_argument('--series-steps', type=int, default=46)
    args = ap.parse_args()

    in_group = 'RANK' in os.environ and 'WORLD_SIZE' in os.environ
    # MOD16_BENCH_FORCE_GROUP=1: the N > 1 step (process group, side stream, collective) also
    # for ONE rank -- this process becomes rank 0 of a group of one (it has not touched the GPU)
    force_group = os.environ.get('MOD16_BENCH_FORCE_GROUP') == '1'
    if force_group and not in_group and args.gpus == 1:
        os.environ.update(rank_env({}, 0, 1, free_port()))
        in_group = True
    if not in_group and args.gpus > 1:
        # the parent of a multi-GPU run: no torch.cuda, no mod16_amd, no GPU
        return spawn(sys.argv[1:], args.gpus)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    args.gpus = world
    if os.environ.get('MOD16_BENCH_PLUMBING') == '1':
        return plumbing_rank(args, rank, world)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_workers)     # before any GPU initialisation

    import numpy as np
    import torch
    import torch.distributed as dist
    from mod16_amd import _lib
    from mod16_amd import dist as tiles
    from mod16_amd.raster import RasterEngine
    from mod16_amd.utils import restore_bplut, bplut_table
    from mod16_amd.models import COLLECTION61_BPLUT

    # rehearsal on a 1-GPU box: MOD16_BENCH_ONE_DEVICE=1 puts every rank on
    # cuda:0 and uses gloo (RCCL refuses two ranks on one device)
    rehearsal = os.environ.get('MOD16_BENCH_ONE_DEVICE') == '1'
    if rehearsal:
        local_rank = 0
    short = too_few_devices(max(world, 1) if not rehearsal else 1)
    if short:                      # (under torch.distributed.run every rank says so and leaves)
        sys.stderr.write(short + '\n')
        return 2
    torch.cuda.set_device(local_rank)
    grouped = world > 1 or force_group
    backend = None
    seen = torch.ones(1, dtype=torch.float64, device='cpu' if rehearsal else 'cuda')
    if grouped:
        backend = 'gloo' if rehearsal else 'nccl'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # RCCL prints a version banner on STDOUT when its first communicator comes up; rank 0's
        # stdout carries exactly one JSON line, so file descriptor 1 points at stderr while the
        # group is created and its first collective runs
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if rehearsal:
                dist.init_process_group('gloo', rank=rank, world_size=world)
            else:
                dist.init_process_group('nccl', rank=rank, world_size=world,
                                        device_id=torch.device('cuda', local_rank))
            # how many ranks the collective layer really has (RCCL for N > 1)
            dist.all_reduce(seen)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    ranks_seen = int(seen.item())

    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    bplut = None
    math = {'fast': _lib.MATH_FAST, 'mixed': _lib.MATH_MIXED}[args.math]
    eng = RasterEngine(table, device=local_rank, dtype=args.dtype, math=math)
    offset, n = tiles.pixel_range(args.rows, args.cols, rank, world)
    total = args.rows * args.cols
    bpp = eng.bytes_per_pixel

    def fence():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the raster of the timed steps, resident before the timed region
    layout_info = {'layout': args.layout}
    if args.layout == 'tiled':
        ras = eng.synth_tiled(eng.alloc_tiled(n), seed=SEED, step=0, pixel_offset=offset)
        layout_info.update(tile_pixels=ras.tile, tile_bytes_per_field=ras.tile * eng.np_dtype.itemsize,
                           note='[tile][14 drivers][tile pixels] + [tile][day, night][tile pixels]; '
                                'every array a 2-D strided view (mod16_layout)')
        cls = drv = day = night = None
    else:
        ras = None
        (cls, drv, day, night), tune = eng.alloc_raster_tuned(n)
        eng.synth(n, seed=SEED, step=0, pixel_offset=offset, out=(cls, drv))
        layout_info.update(slab=tune, note='16 plain arrays in one slab, spacing chosen by measurement at set-up')
    # ET + diagnostics in one pass, one HIP graph launch per step. Two diagnostics
    # vectors: the all-reduce of step s runs on a side stream under the kernel of
    # step s + 1 (N > 1); the kernel of step s + 2, which reuses the vector, waits.
    diags = [torch.zeros(8, dtype=torch.float64, device='cuda') for _ in range(2)]
    if ras is not None:
        steps_bound = [eng.bind_tiled(ras, d) for d in diags]
    else:
        steps_bound = [eng.bind(cls, drv, day, night, d, graph=True) for d in diags]
    main_stream = torch.cuda.current_stream()
    comm_stream = torch.cuda.Stream() if grouped else None
    produced = [torch.cuda.Event() for _ in range(2)]
    reduced = [torch.cuda.Event() for _ in range(2)]
    counter = [0]

    def step(events=None):
        k = counter[0] & 1
        counter[0] += 1
        if comm_stream is None:
            steps_bound[k]()
            return
        main_stream.wait_event(reduced[k])          # the reduction that last used this vector
        if events is not None:
            events[0].record(main_stream)
        steps_bound[k]()
        if events is not None:
            events[1].record(main_stream)
        produced[k].record(main_stream)
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(produced[k])
            tiles.allreduce_diag(diags[k], engine=eng)
            reduced[k].record(comm_stream)

    for _ in range(args.warmup):
        step()
    fence()
    # MOD16_BENCH_CENSUS=1: who holds the GPU open while all ranks are up
    census = gpu_process_census() if rank == 0 and os.environ.get('MOD16_BENCH_CENSUS') == '1' else None
    # the dominant kernel is timed INSIDE the timed steps: an event pair on the launch stream
    # around every step's graph launch (counter reset + pipeline kernel + the kernel that revisits
    # flagged pixels + the fixed-order sum), so kernel_ms <= ms_per_step by construction
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    bare_step = step if comm_stream is None else None
    t0 = time.perf_counter()
    for s_ in range(args.steps):
        if bare_step is not None:
            ev[s_][0].record(main_stream)
            bare_step()
            ev[s_][1].record(main_stream)
        else:
            step(ev[s_])
    fence()
    elapsed = time.perf_counter() - t0
    step_ms = [a.elapsed_time(b) for a, b in ev]
    eng.check()
    if grouped:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kernel_ms = sum(step_ms) / len(step_ms)
    kernel_ms_min = min(step_ms)
    achieved = bpp * n / (kernel_ms * 1e-3) / 1e9
    torch.cuda.synchronize()
    # clock and power under this load (same count of extra steps on every rank: `elapsed` is the maximum over them)
    under_load = None
    if not args.no_sensors:
        under_load = device_under_load(torch, step, max(20, int(2.0 / max(elapsed / args.steps, 1e-4))), rank == 0)
        fence()
    # load balance: every rank's kernel time (its band is 1/N of the grid)
    rank_kernel_ms = [kernel_ms]
    if grouped:
        kt = torch.zeros(world, dtype=torch.float64, device='cpu' if rehearsal else 'cuda')
        kt[rank] = kernel_ms
        dist.all_reduce(kt)
        rank_kernel_ms = kt.tolist()
    diag = diags[(counter[0] - 1) & 1]      # the last step's vector, already reduced over the ranks by that step
    diag_host = diag.cpu().numpy()

    def window(field_plain, field_tiled, lo, hi):
        return ras.flat(field_tiled, lo, hi) if ras is not None else field_plain[lo:hi]

    parity = None
    if not args.no_parity:
        from oracle import mod16_oracle as oracle
        bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
    if rank == 0 and not args.no_parity:
        # 1200 x 1200-pixel windows of this band (start, two inside, end), inputs
        # AND outputs copied back; the oracle runs on exactly those input bits
        m = min(n, TILE[0] * TILE[1])
        starts = sorted(set([0, (n // 3) // 4 * 4, (2 * n // 3) // 4 * 4, n - m]))
        worst, masks = 0.0, True
        for s0 in starts:
            h_cls = window(cls, ras.cls if ras else None, s0, s0 + m).cpu().numpy()
            h_drv = [window(drv[k] if drv else None, ras.drivers[k] if ras else None, s0, s0 + m)
                     .cpu().numpy() for k in range(14)]
            # float32 data: the kernel widens, computes in float64 and rounds
            # once, so the checker is the float64 oracle on the widened inputs
            want = oracle.evapotranspiration_raster(
                bplut, h_cls, *[d.astype(np.float64) for d in h_drv])
            want = [w.astype(h_drv[0].dtype) for w in want]
            gots = (window(day, ras.day if ras else None, s0, s0 + m).cpu().numpy(),
                    window(night, ras.night if ras else None, s0, s0 + m).cpu().numpy())
            for got, ref in zip(gots, want):
                masks = masks and bool(np.array_equal(np.isnan(got), np.isnan(ref))
                                       and np.array_equal(got == 0, ref == 0))
                ok = np.isfinite(ref) & (ref != 0)
                worst = max(worst, float(np.max(
                    np.abs(got[ok].astype(np.float64) - ref[ok]) / np.abs(ref[ok]))))
        parity = {'pixels': int(m * len(starts)), 'tiles': len(starts), 'max_rel_err': worst,
                  'masks_equal': masks, 'rtol_north_star': 1e-5,
                  'against': 'numpy oracle on the same input bits'}
    if not args.no_parity:
        # every pixel of the band on the device, chunk by chunk: the production kernel
        # against the kernel that keeps the reference's operation order (IEEE divide /
        # pow; float32 + mixed: against the float64 arithmetic rounded once)
        ref_math = _lib.MATH_FAST if args.math == 'mixed' else _lib.MATH_EXACT
        ref_eng = RasterEngine(table, device=local_rank, dtype=args.dtype, math=ref_math)
        thresholds = ('1e-6', '1e-5', '1e-4', '1e-3') if args.math == 'mixed' else ('1e-9', '1e-5')
        full = None
        chunk = 1 << 27
        for lo in range(0, n, chunk):
            hi = min(n, lo + chunk)
            c_cls = window(cls, ras.cls if ras else None, lo, hi)
            c_drv = [window(drv[k] if drv else None, ras.drivers[k] if ras else None, lo, hi)
                     for k in range(14)]
            rday, rnight = ref_eng.run(c_cls, c_drv)
            ref_eng.check()
            for got, ref in ((window(day, ras.day if ras else None, lo, hi), rday),
                             (window(night, ras.night if ras else None, lo, hi), rnight)):
                res = compare_on_device(torch, got, ref, thresholds)
                if full is None:
                    full = res
                else:
                    merge_compare(full, res)
            del c_cls, c_drv, rday, rnight
        full['pixels'] = int(n)
        if grouped:         # every rank's band counts
            mx = torch.tensor([full['max_rel_err'], full['max_abs_err_over_max_value'],
                               -float(full['nan_masks_equal'])], dtype=torch.float64, device='cuda')
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            keys = ['zero_mask_mismatches', 'pixels'] + ['n_rel_err_gt_' + t for t in thresholds]
            sm = torch.tensor([full[k] for k in keys], dtype=torch.float64, device='cuda')
            dist.all_reduce(sm, op=dist.ReduceOp.SUM)
            full.update(max_rel_err=float(mx[0]), max_abs_err_over_max_value=float(mx[1]),
                        nan_masks_equal=bool(mx[2] == -1))
            full.update({k: int(v) for k, v in zip(keys, sm.tolist())})
        if parity is not None:
            key = 'full_grid_mixed_vs_float64_arithmetic' if args.math == 'mixed' \
                else 'full_grid_fast_vs_exact_kernel'
            parity[key] = full
        del ref_eng

    # the same kernel on 16 plain arrays (the layout of the reference's arguments)
    plain = None
    copy_gbps = None
    if args.layout == 'tiled' and not args.no_plain and (world == 1 or args.plain):
        steps_bound = ras = None
        torch.cuda.empty_cache()
        try:
            (pcls, pdrv, pday, pnight), tune = eng.alloc_raster_tuned(n)
        except RuntimeError as exc:
            # no silent fall-back: say so, loudly, and measure the back-to-back layout
            sys.stderr.write('bench.py: WARNING: no room for the slab-spacing candidates (%s); '
                             'plain arrays measured back to back\n' % str(exc)[:200])
            torch.cuda.empty_cache()
            pcls, pdrv, pday, pnight = eng.alloc_raster(n)
            tune = {'chosen_extra_bytes': 0, 'tuning_failed': str(exc)[:200]}
        eng.synth(n, seed=SEED, step=0, pixel_offset=offset, out=(pcls, pdrv))
        pstep = eng.bind(pcls, pdrv, pday, pnight, diags[1], graph=True)
        pstep()
        p_ms = pstep.time(max(3, min(args.steps, 20)))
        plain = {'kernel_ms': p_ms, 'achieved': bpp * n / (p_ms * 1e-3) / 1e9,
                 'frac': bpp * n / (p_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 'slab': tune}
        del pstep, pcls, pdrv, pday, pnight
        torch.cuda.empty_cache()
    if rank == 0:
        copy_gbps = eng.measure_copy(4 << 30, 3)

    # real-data ingest in the light input form (every rank has its own PCIe link): N > 1 runs it on every
    # rank beside the resident band; N = 1 runs it with the other configurations below (with the oracle)
    ingest = None
    if world > 1 and not args.no_ingest:
        n_in = (args.rows * args.cols // 32) // 8192 * 8192
        fence()
        mine = ingest_raw_series(torch, np, _lib, RasterEngine, table, local_rank, _lib.MATH_MIXED, n_in, args.series_steps)
        rates = torch.zeros(world, 2, dtype=torch.float64, device='cpu' if rehearsal else 'cuda')
        rates[rank, 0], rates[rank, 1] = mine['pixels_per_s'], mine['h2d_GBps']
        dist.all_reduce(rates)
        ingest = dict(mine, pixels_per_s=float(rates[:, 0].sum()), h2d_GBps=float(rates[:, 1].sum()),
                      pixels_per_s_by_rank=rates[:, 0].tolist(), h2d_GBps_by_rank=rates[:, 1].tolist(),
                      note='every rank streams its own series at the same time; sums over the ranks')
    # the numpy drop-in itself over the GPUs of this run (mod16_amd.multi: one host thread, context
    # and PCIe link per device, no collective): rank 0 alone calls it with devices=range(N) while
    # the other ranks wait at the fence. A failure here is recorded, never raised: the scaling
    # line of an 8-GPU run must not depend on it.
    host_call = None
    if not args.no_host_call and not args.no_configs:
        fence()
        if rank == 0:
            try:
                if bplut is None and not args.no_parity:
                    from oracle import mod16_oracle as oracle
                    bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
                # (a launcher may show a rank its own GPU only: then the leg runs on what this rank sees)
                visible = torch.cuda.device_count()
                devs = [0] * world if rehearsal else [torch.cuda.current_device()] if visible < world else list(range(world))
                host_call = numpy_in_numpy_out(np, torch, eng, table, devs, bplut)
            except Exception as exc:        # noqa: BLE001 -- recorded in the line
                host_call = {'error': '%s: %s' % (type(exc).__name__, str(exc)[:300])}
        fence()
    configs = None
    if rank == 0 and world == 1 and not args.no_configs and args.dtype == 'float64':
        # the other configurations need the whole card: drop this raster first
        steps_bound = ras = cls = drv = day = night = None
        torch.cuda.empty_cache()
        if bplut is None:
            from oracle import mod16_oracle as oracle
            bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
        configs = other_configs(args, torch, np, _lib, RasterEngine, table, bplut)
        ingest = configs['c4_series_float64']['host_ingest_raw']['float32_mixed']

    if rank == 0:
        traffic, traffic_source, traffic_note = pmc_traffic(n, args.dtype, args.layout, _lib.build_id())
        value = total * args.steps / elapsed
        line = {
            'metric': 'pixels/sec, fused Penman-Monteith ET forward run (day+night), 43200x21600 global grid',
            'value': value, 'unit': 'pixels/s', 'n_gpus': world, 'ranks_seen': ranks_seen,
            'process_group': {'backend': backend, 'world_size': world, 'forced_for_one_rank': bool(force_group and world == 1)},
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f64' if args.dtype == 'float64' else 'f32', 'data': 'synthetic',
            'config': {
                'workload': '%dx%d global ET grid, one timestep, %s, %d row band(s) of %d-%d rows'
                            % (args.cols, args.rows, args.dtype, world,
                               args.rows // world, -(-args.rows // world)),
                'pixels': total, 'pixels_per_gpu': n, 'parallelism': 'tile-dp%d' % world,
                'math': args.math, 'bplut': os.path.basename(COLLECTION61_BPLUT),
                'step': 'fused ET kernel with in-kernel diagnostics + the kernel that revisits pixels outside the fast '
                        "arithmetic's domain + fixed-order final sum (one HIP graph launch) + one all-gather(8 doubles), "
                        'reduced in rank order, overlapped with the next step on a side stream',
                'raster_layout': layout_info,
            },
            # (order: the long per-configuration objects first, what the contract asks for last --
            # the driver's record keeps the tail of stdout)
            'configs': configs,
            'ingest': ingest,
            'numpy_in_numpy_out': host_call,
            'diagnostics': dict(zip(DIAG_NAMES, [float(v) for v in diag_host])),
            'parity': parity,
            'cpu_baseline': cpu,
            'roofline': {
                'bound': 'hbm', 'kernel': 'et_stream_kernel<%s, %s> (LDS-DMA, dynamic runs, in-kernel diagnostics)'
                                          % (args.dtype, 'totals, mixed precision' if args.math == 'mixed' else 'totals'),
                'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBPS,
                'traffic': traffic, 'traffic_source': traffic_source, 'traffic_note': traffic_note,
                'traffic_unit': 'bytes per launch', 'library_build_id': _lib.build_id(),
                'bytes_per_pixel': bpp, 'pixels_per_launch': n,
                'kernel_ms': kernel_ms, 'kernel_ms_min': kernel_ms_min,
                'kernel_ms_note': 'mean / min over the timed steps themselves (an event pair around each '
                                  "step's graph launch on the launch stream)",
                'kernel_ms_by_rank': rank_kernel_ms,
                'kernel_pixels_per_s': n / (kernel_ms * 1e-3),
                'measured_copy_GBps': copy_gbps,
                'frac_of_measured_copy': achieved / copy_gbps if copy_gbps else None,
                'device_under_load': under_load,
                'limited_by': ('package power cap: %.0f of %.0f W while the step runs, shader clock %.0f MHz'
                               % (under_load['power_w'], under_load['power_cap_w'], under_load['sclk_mhz'])
                               if under_load and under_load.get('power_cap_w')
                               and under_load['power_w'] >= 0.97 * under_load['power_cap_w'] else None),
                'plain_arrays': plain,
            },
        }
        if census is not None:
            line['gpu_process_census'] = census
        # the scalars the driver's record must carry, repeated flat: in `roofline` (the driver keeps
        # its scalars) and as the LAST key of the line (the driver keeps the tail of stdout)
        line['roofline'].update(roofline_scalars(line))
        line['summary'] = build_summary(line)
        print(json.dumps(line), flush=True)
    if grouped:
        dist.destroy_process_group()
    return 0


def _dig(d, *path):
    for key in path:
        if not isinstance(d, dict) or d.get(key) is None:
            return None
        d = d[key]
    return d


def roofline_scalars(line):
    """plain_frac, sclk_mhz, power_w, cycles_per_step as scalars of `roofline`. cycles_per_step
    = kernel_ms x shader clock under load: the device-independent cost of the step (a build that
    needs fewer cycles is faster on every device; the clock is what the power cap leaves)."""
    roof = line.get('roofline') or {}
    sclk = _dig(roof, 'device_under_load', 'sclk_mhz')
    kms = roof.get('kernel_ms')
    return {'plain_frac': _dig(roof, 'plain_arrays', 'frac'), 'plain_ms': _dig(roof, 'plain_arrays', 'kernel_ms'),
            'sclk_mhz': sclk, 'power_w': _dig(roof, 'device_under_load', 'power_w'),
            'cycles_per_step': kms * sclk * 1e3 if kms and sclk else None}


def build_summary(line):
    """Flat object of scalars, the LAST key of the line: every configuration's headline number
    in a few hundred bytes, so that the tail of stdout the driver keeps carries them all
    (tests/test_host_logic.py parses the last 6000 bytes of a full line)."""
    roof, cfg = line.get('roofline') or {}, line.get('configs') or {}
    c2, c4, c5 = (cfg.get(k) or {} for k in ('c2_1200x1200_float64', 'c4_series_float64', 'c5_global_grid_float32'))
    c1 = cfg.get('c1_single_site') or {}
    f64, f32 = cfg.get('forms_float64') or {}, cfg.get('forms_float32_mixed') or {}
    par = line.get('parity') or {}
    full = par.get('full_grid_fast_vs_exact_kernel') or par.get('full_grid_mixed_vs_float64_arithmetic') or {}
    host = line.get('numpy_in_numpy_out') or {}
    out = {
        'n_gpus': line.get('n_gpus'), 'ranks_seen': line.get('ranks_seen'), 'ms_per_step': line.get('ms_per_step'),
        'gpx_s': line['value'] / 1e9 if line.get('value') else None,
        'kernel_ms': roof.get('kernel_ms'), 'kernel_ms_min': roof.get('kernel_ms_min'), 'frac': roof.get('frac'),
        'traffic_bytes_per_pixel': roof['traffic'] / roof['pixels_per_launch'] if roof.get('traffic') else None,
        'copy_GBps': roof.get('measured_copy_GBps'),
        'sclk_mhz': roof.get('sclk_mhz'), 'power_w': roof.get('power_w'), 'cycles_per_step': roof.get('cycles_per_step'),
        'plain_ms': roof.get('plain_ms'), 'plain_frac': roof.get('plain_frac'),
        'c1_scalar_call_us': _dig(c1, 'scalars', 'call_us'), 'c1_scalar_numpy_us': _dig(c1, 'scalars', 'numpy_oracle_us'),
        'c1_year_call_us': _dig(c1, 'site_year_365', 'call_us'), 'c1_year_numpy_us': _dig(c1, 'site_year_365', 'numpy_oracle_us'),
        'c1_et_static_call_us': _dig(c1, 'et_static_365x30', 'call_us'), 'c1_et_static_numpy_us': _dig(c1, 'et_static_365x30', 'numpy_oracle_us'),
        'c2_us': c2.get('tile_us_per_launch'), 'c2_us_no_diag': c2.get('tile_us_per_launch_without_diagnostics'),
        'c2_batch64_us_per_tile': c2.get('batch64_us_per_tile'),
        'c4_ms_per_step': c4.get('ms_per_step'),
        'c4_gpx_s': c4['pixels_per_s'] / 1e9 if c4.get('pixels_per_s') else None,
        'c4_step_over_sum_of_kernels': c4.get('step_over_sum_of_kernels'),
        'c5_mixed_ms': c5.get('mixed_ms'), 'c5_mixed_frac': c5.get('mixed_frac'),
        'c5_fast_ms': c5.get('fast_float64_arithmetic_ms'), 'c5_fast_frac': c5.get('fast_float64_arithmetic_frac'),
        'c5_masks_equal': _dig(c5, 'mixed_vs_float64_arithmetic_full_grid', 'nan_masks_equal'),
        'c5_max_abs_err_over_max': _dig(c5, 'mixed_vs_float64_arithmetic_full_grid', 'max_abs_err_over_max_value'),
        'pet_f64_frac': _dig(f64, 'potential_et', 'frac'), 'sep_f64_frac': _dig(f64, 'components', 'frac'),
        'raw_f64_frac': _dig(f64, 'raw_drivers', 'frac'), 'raw8_f64_frac': _dig(f64, 'raw_drivers_total8', 'frac'),
        'raw_mixed_frac': _dig(f32, 'raw_drivers', 'frac'), 'raw8_mixed_frac': _dig(f32, 'raw_drivers_total8', 'frac'),
        'raw_f64_max_rel': _dig(f64, 'raw_drivers', 'parity', 'max_rel_err_vs_oracle'),
        'ingest_gpx_s': _dig(line, 'ingest', 'pixels_per_s') / 1e9 if _dig(line, 'ingest', 'pixels_per_s') else None,
        'ingest_h2d_GBps': _dig(line, 'ingest', 'h2d_GBps'),
        'host_call_gpx_s': host['pixels_per_s'] / 1e9 if host.get('pixels_per_s') else None,
        'host_call_pinned_gpx_s': _dig(host, 'pinned_drivers', 'pixels_per_s') / 1e9 if _dig(host, 'pinned_drivers', 'pixels_per_s') else None,
        'host_call_devices': host.get('n_devices'), 'host_call_one_device_gpx_s':
            host['one_device_pixels_per_s'] / 1e9 if host.get('one_device_pixels_per_s') else None,
        'host_call_bits_equal_one_device': host.get('bits_equal_one_device'), 'host_call_error': host.get('error'),
        'n2_resident_frac': _dig(cfg, 'n2_calibration', 'fast_resident', 'frac_of_valu_issue_peak'),
        'n2_resident_gpu_frac': _dig(cfg, 'n2_calibration', 'fast_resident', 'frac_of_valu_issue_peak_gpu_part'),
        'parity_max_rel': par.get('max_rel_err'), 'parity_masks_equal': par.get('masks_equal'),
        'full_grid_max_rel': full.get('max_rel_err'), 'full_grid_masks_equal': full.get('nan_masks_equal'),
        'full_grid_n_gt_1e-5': full.get('n_rel_err_gt_1e-5'),
        'cpu_px_s_1core': _dig(line, 'cpu_baseline', 'value'),
        'cpu_pool_px_s': _dig(line, 'cpu_baseline', 'pool', 'value'), 'cpu_pool_cores': _dig(line, 'cpu_baseline', 'pool', 'cores'),
        'cpu_fused_px_s': _dig(line, 'cpu_baseline', 'fused_numpy', 'value'),
        'store_gpx_s': _dig(cfg, 'store_on_disk', 'pixels_per_s') / 1e9 if _dig(cfg, 'store_on_disk', 'pixels_per_s') else None,
        'class_device_f64_frac': _dig(cfg, 'class_surface_device', 'one_pft_float64', 'frac'),
        'class_device_f32_frac': _dig(cfg, 'class_surface_device', 'one_pft_float32', 'frac'),
        'class_gather_f64_frac': _dig(cfg, 'class_surface_device', 'pft_gather_float64', 'frac'),
        'class_gather_f32_frac': _dig(cfg, 'class_surface_device', 'pft_gather_float32', 'frac'),
        'class_device_f32_mixed_frac': _dig(cfg, 'class_surface_device', 'one_pft_mixed_float32', 'frac'),
        'c5_trusted_frac': c5.get('mixed_trusted_frac'),
        'c5_n_gt_1e-4': _dig(c5, 'mixed_vs_float64_arithmetic_full_grid', 'n_rel_err_gt_1e-4'),
        'c5_n_gt_1e-3': _dig(c5, 'mixed_vs_float64_arithmetic_full_grid', 'n_rel_err_gt_1e-3'),
        'c5_max_rel_err': _dig(c5, 'mixed_vs_float64_arithmetic_full_grid', 'max_rel_err'),
        'build_id': roof.get('library_build_id'),
    }
    return out


def numpy_in_numpy_out(np, torch, eng, table, devices, bplut, tiles_per_device=16):
    """The numpy drop-in over the GPUs of the run: evapotranspiration_raster(..., devices=range(N),
    diagnostics=True) on host arrays of `tiles_per_device` staging tiles per device (weak scaling:
    33.5 M float64 pixels = 4.3 GB of drivers per device), wall time of the whole call -- every
    byte crosses PCIe, one link per device. Beside it the same call on ONE device over one
    device's share (the one-link rate), the bit-equality of the two on that share, and with
    `bplut` the oracle on a 320 k-pixel window across the first cut."""
    import mod16_amd
    from mod16_amd import multi
    tile = multi.host_tile()
    per_dev = tiles_per_device * tile
    n = per_dev * len(devices)
    h_cls = np.empty(n, np.uint8)
    h_drv = [np.empty(n, np.float64) for _ in range(14)]
    for lo in range(0, n, 1 << 24):         # generated on the device chunk by chunk, copied down
        m = min(1 << 24, n - lo)
        c, d = eng.synth(m, seed=SEED, step=3, pixel_offset=lo)
        h_cls[lo:lo + m] = c.cpu().numpy()
        for k in range(14):
            h_drv[k][lo:lo + m] = d[k].cpu().numpy()
        del c, d
    torch.cuda.empty_cache()
    call = lambda hi, devs: mod16_amd.evapotranspiration_raster(
        table, h_cls[:hi], *[x[:hi] for x in h_drv], devices=devs, diagnostics=True)
    call(min(n, 2 * tile * len(devices)), devices)          # contexts, slabs, pinned result blocks
    best, res = 1e30, None
    for _ in range(2):
        res = None                                           # (its blocks go back to the pool)
        t0 = time.perf_counter()
        res = call(n, devices)
        best = min(best, time.perf_counter() - t0)
    out = {'pixels': n, 'n_devices': len(devices), 'devices': list(devices), 'visible_devices': torch.cuda.device_count(), 'seconds': best,
           'pixels_per_s': n / best, 'pcie_GBps_both_directions': 129.0 * n / best / 1e9,
           'diagnostics': [float(v) for v in res[2]],
           'note': 'mod16_amd.evapotranspiration_raster(table, cls, *drivers, devices=range(N), diagnostics=True) '
                   'on host float64 arrays, %d staging tiles of %d pixels per device; best of 2 calls' % (tiles_per_device, tile)}
    # the same call with the drivers in page-locked memory (mod16_amd.pinned_empty): the host-to-device
    # copies are pure DMA, no staging copy on the host's threads -- what keeps eight links fed
    try:
        p_cls = mod16_amd.pinned_empty(n, np.uint8)
        p_drv = [mod16_amd.pinned_empty(n, np.float64) for _ in range(14)]
        p_cls[:] = h_cls
        for a, b in zip(p_drv, h_drv):
            a[:] = b
        pinned_call = lambda: mod16_amd.evapotranspiration_raster(table, p_cls, *p_drv, devices=devices, diagnostics=True)
        pinned_call()
        t0 = time.perf_counter()
        pres = pinned_call()
        dt = time.perf_counter() - t0
        out['pinned_drivers'] = {'pixels_per_s': n / dt, 'seconds': dt,
                                 'bits_equal_pageable_call': bool(np.array_equal(pres[0], res[0], equal_nan=True)
                                                                  and np.array_equal(pres[2], res[2])),
                                 'page_locked': bool(p_drv[0].base is not None)}
        del pres, p_cls, p_drv
    except Exception as exc:        # noqa: BLE001 -- recorded
        out['pinned_drivers'] = {'error': '%s: %s' % (type(exc).__name__, str(exc)[:200])}
    if len(devices) > 1:
        call(2 * tile, devices[:1])
        t0 = time.perf_counter()
        one = call(per_dev, devices[:1])
        out['one_device_pixels_per_s'] = per_dev / (time.perf_counter() - t0)
        out['bits_equal_one_device'] = bool(np.array_equal(one[0], res[0][:per_dev], equal_nan=True)
                                            and np.array_equal(one[1], res[1][:per_dev], equal_nan=True))
        del one
    if bplut is not None:
        from oracle import mod16_oracle as oracle
        lo = max(0, per_dev - 160000) if len(devices) > 1 else n // 2
        hi = min(n, lo + 320000)
        with np.errstate(all='ignore'):
            want = oracle.evapotranspiration_raster(bplut, h_cls[lo:hi], *[x[lo:hi] for x in h_drv])
        worst, masks = 0.0, True
        for got, ref in zip(res[:2], want):
            got = got[lo:hi]
            masks = masks and bool(np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(got == 0, ref == 0))
            ok = np.isfinite(ref) & (ref != 0)
            worst = max(worst, float(np.max(np.abs(got[ok] - ref[ok]) / np.abs(ref[ok]))))
        out['parity'] = {'pixels': hi - lo, 'window_start': lo, 'max_rel_err_vs_oracle': worst, 'masks_equal': masks}
    return out


def single_site_config(np, oracle):
    """BASELINE.json configs[0]: ONE call of the drop-in, numpy in -> numpy out, on the scalars of the
    reference's flux-tower test (tests/tests.py:21-54, golden fixture F1), on a year of that site
    (365 values per driver) and on a 100 x 100 window -- wall clock of MOD16.evapotranspiration()
    (best and median of 300; HOST mode: calls this small go through the library's copy-free path)
    next to the numpy oracle on this box's host cores for the same call, each result checked
    against the oracle's."""
    import mod16_amd
    f1 = np.load(os.path.join(ROOT, 'tests', 'golden', 'f1_tests_scalars.npz'))
    params = dict(zip(mod16_amd.MOD16.required_parameters, (float(v) for v in f1['params'])))
    model = mod16_amd.MOD16(params)
    rng = np.random.default_rng(SEED)
    out = {}

    def timed(fn, reps):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[0] * 1e6, ts[len(ts) // 2] * 1e6
    for name, shape in (('scalars', ()), ('site_year_365', (365,)), ('window_100x100', (100, 100))):
        drv = [float(v) * (1 + 0.01 * rng.uniform(-1, 1, shape)) if shape else float(v) for v in f1['drivers']]
        got = model.evapotranspiration(*drv)
        want = oracle.evapotranspiration(params, *drv)
        err = max(float(np.max(np.abs(np.asarray(g) - np.asarray(w)) / np.abs(np.asarray(w)))) for g, w in zip(got, want))
        best, med = timed(lambda: model.evapotranspiration(*drv), 300)
        obest, omed = timed(lambda: oracle.evapotranspiration(params, *drv), 30)
        out[name] = {'pixels': int(np.prod(shape, dtype=np.int64)) if shape else 1, 'call_us': best, 'call_us_median': med,
                     'numpy_oracle_us': obest, 'numpy_oracle_us_median': omed, 'max_rel_err_vs_oracle': err}
    # the calibration interface a sampler calls once per draw (reference mod16/__init__.py:162-193): a year of 30 sites
    plist = [params[k] for k in mod16_amd.MOD16.required_parameters]
    drv = [float(v) * (1 + 0.01 * rng.uniform(-1, 1, (365, 30))) for v in f1['drivers']]
    got = mod16_amd.MOD16._et(plist, *drv)
    want = oracle.et_static(plist, *drv)
    ok = np.isfinite(want) & (want != 0)
    err = float(np.max(np.abs(got[ok] - want[ok]) / np.abs(want[ok])))
    best, med = timed(lambda: mod16_amd.MOD16._et(plist, *drv), 100)
    obest, omed = timed(lambda: oracle.et_static(plist, *drv), 10)
    out['et_static_365x30'] = {'pixels': 365 * 30, 'call_us': best, 'call_us_median': med, 'numpy_oracle_us': obest,
                               'numpy_oracle_us_median': omed, 'max_rel_err_vs_oracle': err}
    out['note'] = ('wall clock of one MOD16.evapotranspiration() call (numpy in, numpy out, PCIe and Python included); '
                   'round 4 took 124 us for the scalars -- slower than the reference\'s numpy (86 us)')
    return out


def other_configs(args, torch, np, _lib, RasterEngine, table, bplut):
    """The BASELINE.json configurations beside the headline one, measured in the same
    process on the same GPU (N = 1): configs[1] one 1200 x 1200 float64 tile per launch
    (and 64 tiles per launch), configs[3] the 46-step series streamed through a two-slot
    ring, configs[4] the global grid in float32 (mixed precision and float64 arithmetic).
    Each with its own parity summary."""
    from oracle import mod16_oracle as oracle
    out = {}
    out['c1_single_site'] = single_site_config(np, oracle)
    eng = RasterEngine(table, dtype='float64')
    diag = torch.zeros(8, dtype=torch.float64, device='cuda')

    # ---- configs[1]: 1200 x 1200, one timestep, float64, device resident
    m = TILE[0] * TILE[1]
    c2 = {}
    for name, tiles_per_launch in (('tile', 1), ('batch64', 64)):
        r = eng.synth_tiled(eng.alloc_tiled(m * tiles_per_launch), seed=SEED)
        # a step this short is issued directly: a graph replay's fixed cost (10-20 us)
        # would show; HIP events around 200 back-to-back launches (mod16_time_et_tiled)
        eng.time_tiled(r, 20, diag)
        us = min(eng.time_tiled(r, 200, diag) for _ in range(3)) * 1e3
        c2[name + '_us_per_launch'] = us
        c2[name + '_us_per_launch_without_diagnostics'] = min(eng.time_tiled(r, 200) for _ in range(3)) * 1e3
        c2[name + '_us_per_tile'] = us / tiles_per_launch
        c2[name + '_GBps'] = 129.0 * m * tiles_per_launch / us / 1e3
        if tiles_per_launch == 1:
            eng.check()
            want = oracle.evapotranspiration_raster(
                bplut, r.flat(r.cls).cpu().numpy(), *[r.flat(d).cpu().numpy() for d in r.drivers])
            errs = []
            masks = True
            for got, ref in zip((r.flat(r.day).cpu().numpy(), r.flat(r.night).cpu().numpy()), want):
                masks = masks and bool(np.array_equal(np.isnan(got), np.isnan(ref))
                                       and np.array_equal(got == 0, ref == 0))
                ok = np.isfinite(ref) & (ref != 0)
                errs.append(float(np.max(np.abs(got[ok] - ref[ok]) / np.abs(ref[ok]))))
            c2['parity'] = {'max_rel_err_vs_oracle': max(errs), 'masks_equal': masks, 'pixels': m}
        del r
    c2['target_us_survey'] = 33.0
    c2['note'] = ('*_us_per_launch = the launch PERIOD of 200 back-to-back direct launches (HIP events); the kernel itself takes '
                  'about 5.5 us less -- the constant dispatch + completion gap between two launches on a stream '
                  '(profiles/r05_c2_kernel_vs_launch_gap.json: kernel 38-39 us with diagnostics, 35.6 without)')
    out['c2_1200x1200_float64'] = c2
    torch.cuda.empty_cache()

    # ---- configs[3]: the global grid, 46 8-day steps streamed through HBM (two-slot
    # ring; the on-device generator stands in for the ingest stage on a second stream)
    n = args.rows * args.cols
    ring = [eng.alloc_tiled(n), eng.alloc_tiled(n)]
    eng.run_series_tiled(n, 2, seed=SEED, ring=ring)
    torch.cuda.synchronize()
    lo, hi = (n // 2) // 4 * 4, (n // 2) // 4 * 4 + 320000
    grabbed = {}

    def grab(s, r):
        if s in (0, args.series_steps // 2, args.series_steps - 1):
            grabbed[s] = ([r.flat(d, lo, hi) for d in r.drivers], r.flat(r.cls, lo, hi),
                          r.flat(r.day, lo, hi), r.flat(r.night, lo, hi))

    t0 = time.perf_counter()
    sdiag = eng.run_series_tiled(n, args.series_steps, seed=SEED, on_step=grab, ring=ring)[0]   # ([1]: a ring slot)
    torch.cuda.synchronize()
    t_series = time.perf_counter() - t0
    eng.check()
    # the two kernels of a series step ALONE, on this stream, by events: both stream through the same
    # HBM, so a step cannot beat their sum; it should not be far above it either (the pipeline kernel
    # enqueued right behind a cross-stream wait once cost 45 ms where the sum is 39: RasterEngine._gate)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    for k in range(3):
        eng.synth_tiled(ring[0], seed=SEED, step=k)
    ev[1].record()
    for k in range(3):
        eng.run_tiled(ring[0], diag=diag)
    ev[2].record()
    torch.cuda.synchronize()
    synth_ms, et_ms = ev[0].elapsed_time(ev[1]) / 3, ev[1].elapsed_time(ev[2]) / 3
    worst, masks = 0.0, True
    for s, (d, c, gd, gn) in grabbed.items():
        want = oracle.evapotranspiration_raster(bplut, c.cpu().numpy(), *[x.cpu().numpy() for x in d])
        for got, ref in zip((gd.cpu().numpy(), gn.cpu().numpy()), want):
            masks = masks and bool(np.array_equal(np.isnan(got), np.isnan(ref))
                                   and np.array_equal(got == 0, ref == 0))
            ok = np.isfinite(ref) & (ref != 0)
            worst = max(worst, float(np.max(np.abs(got[ok] - ref[ok]) / np.abs(ref[ok]))))
    out['c4_series_float64'] = {
        'steps': args.series_steps, 'seconds': t_series,
        'pixels_per_s': n * args.series_steps / t_series,
        'ms_per_step': 1e3 * t_series / args.series_steps,
        'generator_kernel_ms': synth_ms, 'et_kernel_ms': et_ms,
        'step_over_sum_of_kernels': 1e3 * t_series / args.series_steps / (synth_ms + et_ms),
        'note': 'includes producing every step\'s 14 driver arrays on the device '
                '(113 B/pixel written by the generator on a second stream): 242 B/pixel of traffic per step',
        'GBps_total_traffic': 242.0 * n * args.series_steps / t_series / 1e9,
        'sum_day_first_last': [float(sdiag[0, 0]), float(sdiag[-1, 0])],
        'parity': {'steps_checked': sorted(grabbed), 'pixels_per_step': hi - lo,
                   'max_rel_err_vs_oracle': worst, 'masks_equal': masks},
    }
    del ring, grabbed, sdiag
    torch.cuda.empty_cache()
    out['c4_series_float64']['host_ingest'] = series_from_host(torch, eng, args)
    n_in = (args.rows * args.cols // 32) // 8192 * 8192
    out['c4_series_float64']['host_ingest_raw'] = {
        name: ingest_raw_series(torch, np, _lib, RasterEngine, table, torch.cuda.current_device(), math, n_in,
                                args.series_steps, bplut)
        for name, math in (('float32_fast', _lib.MATH_FAST), ('float32_mixed', _lib.MATH_MIXED))}

    # ---- configs[4]: the global grid in float32, mixed precision vs float64 tolerance
    c5 = {}
    e_mixed = RasterEngine(table, dtype='float32', math=_lib.MATH_MIXED)
    e_fast = RasterEngine(table, dtype='float32', math=_lib.MATH_FAST)
    r = e_mixed.synth_tiled(e_mixed.alloc_tiled(n), seed=SEED)
    ref = e_fast.alloc_tiled(n)
    ref.slab.copy_(r.slab)
    e_trusted = RasterEngine(table, dtype='float32', math=_lib.MATH_MIXED, trusted=True)
    # (the guarded form runs LAST on its raster: its values are the ones compared below)
    for name, e, ras in (('mixed_trusted', e_trusted, r), ('mixed', e_mixed, r), ('fast_float64_arithmetic', e_fast, ref)):
        step = e.bind_tiled(ras, diag)
        step()
        ms = min(step.time(10) for _ in range(2))
        e.check()
        c5[name + '_ms'] = ms
        c5[name + '_GBps'] = 65.0 * n / ms / 1e6
        c5[name + '_frac'] = 65.0 * n / ms / 1e6 / HBM_PEAK_GBPS
        del step
    full = None
    for got, want in ((r.day, ref.day), (r.night, ref.night)):
        res = compare_on_device(torch, r.flat(got), ref.flat(want), ('1e-6', '1e-5', '1e-4', '1e-3'))
        if full is None:
            full = res
        else:
            merge_compare(full, res)
    full['pixels'] = n
    c5['mixed_vs_float64_arithmetic_full_grid'] = full
    c5['note'] = ('tolerance check, not a 1e-5 guarantee: the mixed form keeps every NaN / exact-zero decision of '
                  'the float64 arithmetic and bounds the ABSOLUTE error; relative error exceeds 1e-5 on the '
                  'counted share of (small) values')
    out['c5_global_grid_float32'] = c5
    del r, ref, e_mixed, e_fast, e_trusted, e, ras, got, want, res      # (loop variables hold the rasters too)
    import gc
    gc.collect()                 # bound steps and their launch closures are reference cycles
    torch.cuda.empty_cache()
    if torch.cuda.memory_allocated() > n * 40:
        raise RuntimeError('a raster of an earlier configuration is still alive (%.1f GB allocated)'
                           % (torch.cuda.memory_allocated() / 1e9))
    # ---- the other forms of the forward run (N1, N3, separate=True) and the calibration path (N2)
    out['forms_float64'] = forms_config(torch, np, _lib, RasterEngine, table, bplut, 'float64', _lib.MATH_FAST, 10800)
    out['forms_float32_mixed'] = forms_config(torch, np, _lib, RasterEngine, table, bplut, 'float32', _lib.MATH_MIXED, 10800)
    out['n2_calibration'] = n2_config(np, _lib)
    out['class_surface_device'] = class_surface_config(torch, np, RasterEngine, table)
    out['store_on_disk'] = store_config(torch, np, table)
    return out


def class_surface_config(torch, np, RasterEngine, table, n=10800 * 21600):
    """The reference's own signature on rasters resident in HBM (tools/classbench.py):
    MOD16(params).evapotranspiration(*device tensors), HIP events around 10 calls --
      one_pft      scalar parameters of one plant functional type: the production pipeline behind a class
                   raster of ones, on separately allocated tensors;
      pft_gather   the reference's multi-class idiom (notebook cell 32): MOD16({k: bplut[k][pft_map]}) with
                   eleven per-pixel parameter TENSORS -- recognised as a gather of the table's rows and run
                   through the same pipeline with a class raster built on the device (round 6)."""
    import mod16_amd
    names = mod16_amd.MOD16.required_parameters
    out = {'pixels': n}
    for dtype in ('float64', 'float32'):
        eng = RasterEngine(table, dtype=dtype)
        cls, drv = eng.synth(n, seed=SEED)
        bpp = 129 if dtype == 'float64' else 65
        models = {'one_pft': mod16_amd.MOD16(dict(zip(names, (float(v) for v in table[7]))))}
        tt = torch.from_numpy(table).cuda().to(drv[0].dtype)
        models['pft_gather'] = mod16_amd.MOD16({k: tt[:, j][cls.long()] for j, k in enumerate(names)})
        del tt
        if dtype == 'float32':      # the opt-in mixed-precision form on the class surface (model.math)
            from mod16_amd import _lib
            models['one_pft_mixed'] = mod16_amd.MOD16(dict(zip(names, (float(v) for v in table[7]))))
            models['one_pft_mixed'].math = _lib.MATH_MIXED
        for name, model in models.items():
            res = model.evapotranspiration(*drv)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                res = model.evapotranspiration(*drv)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            out['%s_%s' % (name, dtype)] = {'ms': ms, 'GBps': bpp * n / ms / 1e6, 'frac': bpp * n / ms / 1e6 / HBM_PEAK_GBPS,
                                            'bytes_per_pixel': bpp}
            del res
        del models, model, drv, cls, eng
        mod16_amd.release_device_cache()
        torch.cuda.empty_cache()
    return out


def store_config(torch, np, table, pixels=43200 * 432, steps=2):
    """SURVEY.md 8f N4 with a number in the line: a raster time series on disk (/dev/shm when there is
    one: the page cache either way) run through mod16_amd.io.run_store -- file -> page-locked buffer ->
    H2D -> fused raw-driver kernel -> D2H -> file, four pipelines of three readers. float32 raw drivers,
    `steps` x `pixels`; the set-up (fields written once) is not measured."""
    import shutil
    import tempfile
    from mod16_amd import io
    base = '/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK) else tempfile.gettempdir()
    root = tempfile.mkdtemp(prefix='mod16_bench_store_', dir=base)
    try:
        store = io.RasterStore.create(root, steps, pixels, np.dtype(np.float32))
        g = torch.Generator(device='cuda').manual_seed(SEED)
        ranges = {0: (-100, 0), 1: (-50, 0), 2: (0, 360), 4: (0.1, 0.22), 5: (255, 305), 6: (250, 300),
                  8: (245, 298), 9: (0.001, 0.02), 10: (0.001, 0.02), 11: (70000, 101340), 12: (70000, 101340)}
        rnd = lambda lo, hi: (torch.rand(pixels, generator=g, device='cuda') * (hi - lo) + lo).cpu().numpy()
        for t in range(steps):
            for idx, name in io.DYNAMIC_FIELDS:
                store.array(name, 'r+')[t] = rnd(*ranges[idx])
            store.array(io.FPAR, 'r+')[t] = torch.randint(0, 101, (pixels,), generator=g, device='cuda', dtype=torch.uint8).cpu().numpy()
            store.array(io.LAI, 'r+')[t] = torch.randint(0, 71, (pixels,), generator=g, device='cuda', dtype=torch.uint8).cpu().numpy()
        store.array('MERRA2/T10M_annual', 'r+')[:] = rnd(265, 300)
        store.array('state/elevation_m', 'r+')[:] = rnd(-50, 3500)
        store.array(io.PFT, 'r+')[:] = torch.randint(1, 11, (pixels,), generator=g, device='cuda', dtype=torch.uint8).cpu().numpy()
        del store
        best = None
        for _ in range(2):
            rep = io.run_store(table, root, workers=4, readers=3)
            if best is None or rep['pixels_per_s'] > best['pixels_per_s']:
                best = rep
        keep = {k: best[k] for k in ('pixels_per_s', 'file_GBps', 'wall_s', 'setup_s') if k in best}
        keep.update(pixels=pixels, steps=steps, dtype='float32', where=base,
                    note='mod16_amd.io.run_store: 4 pipelines x 3 readers, raw float32 drivers + uint8 fPAR / LAI in, '
                         'two float32 rasters out; files in the page cache; best of 2 runs')
        return keep
    except Exception as exc:                      # (no room in /dev/shm, ...): recorded, never raised
        return {'error': '%s: %s' % (type(exc).__name__, exc)}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def series_from_host(torch, eng, args):
    """What the series sustains when the drivers of every step come from HOST memory instead of
    the on-device generator (a real ingest): a band of 1/16 of the grid, 6 steps, the 14 driver
    arrays of a step copied from page-locked host memory into the ring slot's tiled raster on a
    second stream (2-D copies: tile-wide rows into the raster's pitch) while the kernel works on
    the previous step. Bound: PCIe (113 B/pixel up per step), not the kernel."""
    n = (args.rows * args.cols // 16) // 4096 * 4096
    steps = 6
    ring = [eng.alloc_tiled(n), eng.alloc_tiled(n)]
    eng.synth_tiled(ring[0], seed=SEED)
    eng.synth_tiled(ring[1], seed=SEED, step=1)
    host = [torch.empty((n // ring[0].tile, ring[0].tile), dtype=eng.dtype, pin_memory=True) for _ in range(14)]
    for k in range(14):
        host[k].copy_(ring[0].drivers[k])
    torch.cuda.synchronize()
    compute = torch.cuda.current_stream()
    ingest = torch.cuda.Stream()
    diag = torch.zeros(steps, 8, dtype=torch.float64, device='cuda')
    filled = [torch.cuda.Event() for _ in range(steps)]
    consumed = [torch.cuda.Event() for _ in range(steps)]

    def produce(s):
        with torch.cuda.stream(ingest):
            if s >= 2:
                ingest.wait_event(consumed[s - 2])
            for k in range(14):
                ring[s % 2].drivers[k].copy_(host[k], non_blocking=True)
            filled[s].record(ingest)

    ingest.wait_stream(compute)
    t0 = time.perf_counter()
    produce(0)
    produce(1)
    for s in range(steps):
        compute.wait_event(filled[s])
        eng.run_tiled(ring[s % 2], diag=diag[s])
        consumed[s].record(compute)
        if s + 2 < steps:
            produce(s + 2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.check()
    same = bool(torch.equal(diag[0], diag[-1]))          # the same host data every step
    return {'pixels': n, 'steps': steps, 'seconds': dt, 'pixels_per_s': n * steps / dt,
            'h2d_GBps': 14 * eng.np_dtype.itemsize * n * steps / dt / 1e9,
            'diagnostics_identical_across_steps': same,
            'note': 'drivers of every step from page-locked host memory (14 x %d B/pixel up per step, overlapped '
                    'with the kernel of the previous step); the class raster stays resident' % eng.np_dtype.itemsize}


def ingest_raw_series(torch, np, _lib, RasterEngine, table, device, math, n, steps, bplut=None):
    """A series whose drivers come from HOST memory in the light input form (the
    reference's own pre-processing inputs, calibration.py:380-423): per step 14 float32 raw fields
    (radiation, temperatures, QV10M, PS, elevation) + uint8 fPAR / LAI = 58 bytes per pixel copied
    from page-locked memory straight into a two-slot ring of FORM_RAW tiled rasters (tile-wide rows
    into the slot's pitch) on a second stream, under the kernel of the previous step; the class
    raster stays resident. Three distinct host records take turns. With `bplut` the oracle checks a
    320 k-pixel window of the first, the middle and the last step. Bound: this rank's PCIe link."""
    eng = RasterEngine(table, device=device, dtype='float32', math=math)
    ring = [eng.alloc_tiled(n, form=_lib.FORM_RAW) for _ in range(2)]
    npad = ring[0].ntiles * ring[0].tile
    g = torch.Generator(device='cuda').manual_seed(7)
    pin = lambda x: torch.empty(x.shape, dtype=x.dtype, pin_memory=True).copy_(x)
    host, cls = [], None
    for k in range(3):
        c, drv = eng.synth(npad, seed=SEED, step=k)
        u = lambda lo, hi: torch.empty(npad, dtype=torch.float32, device='cuda').uniform_(lo, hi, generator=g)
        raw = drv[:9] + [u(0.001, 0.02), u(0.001, 0.02), u(7e4, 1.0134e5), u(7e4, 1.0134e5), u(0, 3500)]
        fpar = torch.randint(0, 101, (npad,), dtype=torch.uint8, device='cuda', generator=g)
        lai = torch.randint(0, 71, (npad,), dtype=torch.uint8, device='cuda', generator=g)
        fill = torch.rand(npad, device='cuda', generator=g) < 0.01
        fpar[fill] = 255
        lai[fill] = 250
        host.append({'wide': [pin(x) for x in raw], 'bytes': [None, pin(fpar), pin(lai)]})
        if cls is None:
            cls = c                          # land cover is static
        del c, drv, raw, fpar, lai, fill
    for r in ring:
        r.bytes[0].copy_(cls.view(r.ntiles, r.tile))
    torch.cuda.empty_cache()
    lo = (n // 2) // 8192 * 8192
    hi = min(n, lo + 320000)
    checked = sorted(set([0, steps // 2, steps - 1]))
    grabbed = {}

    def grab(s, slot):
        if s in checked:
            grabbed[s] = (slot.flat(slot.outs[0], lo, hi), slot.flat(slot.outs[1], lo, hi))

    eng.run_series_host(ring, host, 2)          # warm-up: both slots, both streams
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run_series_host(ring, host, steps, on_step=grab)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.check()
    bpp = 14 * 4 + 2
    res = {'pixels': n, 'steps': steps, 'seconds': dt, 'pixels_per_s': n * steps / dt,
           'bytes_per_pixel_and_step_over_pcie': bpp, 'h2d_GBps': bpp * npad * steps / dt / 1e9,
           'math': 'mixed' if math == _lib.MATH_MIXED else 'fast (float64 arithmetic)',
           'host_records': len(host), 'form': 'FORM_RAW (14 float32 raw fields + uint8 fPAR, LAI; class raster resident)'}
    if bplut is not None:
        from oracle import mod16_oracle as oracle
        worst, worst_abs, masks, n_gt, n_all = 0.0, 0.0, True, 0, 0
        mixed = math == _lib.MATH_MIXED
        tiny = float(np.finfo(np.float32).tiny)
        h_cls = cls[lo:hi].cpu().numpy()
        for s_, outs in grabbed.items():
            rec = host[s_ % len(host)]
            with np.errstate(all='ignore'):
                want = oracle.evapotranspiration_raw(
                    bplut, h_cls, [x[lo:hi].numpy().astype(np.float64) for x in rec['wide']],
                    rec['bytes'][1][lo:hi].numpy(), rec['bytes'][2][lo:hi].numpy())
            for o, w in zip(outs, want):
                got = o.cpu().numpy().astype(np.float64)
                w = w.astype(np.float32).astype(np.float64)
                if mixed:
                    got = np.where(np.abs(got) < tiny, 0, got)
                    w = np.where(np.abs(w) < tiny, 0, w)
                masks = masks and bool(np.array_equal(np.isnan(got), np.isnan(w)) and np.array_equal(got == 0, w == 0))
                ok = np.isfinite(w) & (w != 0)
                err = np.abs(got[ok] - w[ok])
                rel = err / np.abs(w[ok])
                worst = max(worst, float(rel.max()))
                worst_abs = max(worst_abs, float(err.max() / np.abs(w[ok]).max()))
                n_gt += int(np.count_nonzero(rel > 1e-5))
                n_all += int(rel.size)
        res['parity'] = {'steps_checked': sorted(grabbed), 'pixels_per_step': hi - lo, 'masks_equal': masks,
                         'max_rel_err_vs_oracle': worst}
        if mixed:
            res['parity'].update({'max_abs_err_over_max_value': worst_abs, 'within_abs_bound_1e-6': worst_abs <= 1e-6,
                                  'fraction_rel_err_gt_1e-5': n_gt / max(n_all, 1)})
        else:
            res['parity']['within_rtol_1e-6'] = worst <= 1e-6
    del ring, host, grabbed, cls
    torch.cuda.empty_cache()
    return res


def forms_config(torch, np, _lib, RasterEngine, table, bplut, dtype, math, rows):
    """The other forms of the forward run on the tiled layout (SURVEY.md 8f N1 / N3 and
    separate=True): potential ET, six components, totals + components, raw drivers without /
    with the 8-day total. Per form: ms per launch (events on the launch stream), share of the
    HBM peak at the form's algorithmic bytes per pixel, and the numpy oracle on a window of
    320 k pixels (inputs and outputs copied back)."""
    from oracle import mod16_oracle as oracle
    n = rows * 43200
    eng = RasterEngine(table, dtype=dtype, math=math)
    esz = eng.np_dtype.itemsize
    cls, drv = eng.synth(n, seed=SEED)
    g = torch.Generator(device='cuda').manual_seed(1)
    u = lambda lo, hi: torch.empty(n, dtype=eng.dtype, device='cuda').uniform_(lo, hi, generator=g)
    raw = drv[:9] + [u(0.001, 0.02), u(0.001, 0.02), u(7e4, 1.0134e5), u(7e4, 1.0134e5), u(0, 3500)]
    fpar = torch.randint(0, 101, (n,), dtype=torch.uint8, device='cuda', generator=g)
    lai = torch.randint(0, 71, (n,), dtype=torch.uint8, device='cuda', generator=g)
    fill = torch.rand(n, device='cuda', generator=g) < 0.01
    fpar[fill] = 255
    lai[fill] = 250
    hours = u(8, 16)
    lo = (n // 2) // 8192 * 8192
    hi = lo + 320000
    mixed = math == _lib.MATH_MIXED
    rtol = None if mixed else (1e-8 if dtype == 'float64' else 1e-6)
    cases = [('potential_et', _lib.FORM_PET, 14 * esz + 1 + 4 * esz),
             ('components', _lib.FORM_COMPONENTS, 14 * esz + 1 + 6 * esz),
             ('totals_and_components', _lib.FORM_TOTALS_COMPONENTS, 14 * esz + 1 + 8 * esz),
             ('raw_drivers', _lib.FORM_RAW, 14 * esz + 3 + 2 * esz),
             ('raw_drivers_total8', _lib.FORM_RAW_TOTAL8_HOURS, 15 * esz + 3 + 3 * esz)]
    out = {'pixels': n, 'layout': 'tiled', 'parity_rtol': rtol,
           'parity_note': 'largest relative error against the numpy oracle on 320 k pixels; masks = NaN and '
                          'exact-zero masks identical' + (' (float32 subnormals count as zero); the mixed form is held to an '
                                                          'absolute bound, 1e-6 of the largest value, and the share of values '
                                                          'off by more than 1e-5 relative is reported' if mixed else '')}

    def f64(ts):
        return [t.cpu().numpy().astype(np.float64) for t in ts]

    for name, form, bpp in cases:
        r = eng.alloc_tiled(n, form=form)
        is_raw = form >= _lib.FORM_RAW
        wide = (raw + [hours])[:len(r.wide)] if is_raw else drv
        for dst, src in zip(r.wide, wide):
            r.put(dst, src)
        for dst, src in zip(r.bytes, [cls, fpar, lai]):
            r.put(dst, src)
        eng.run_form_tiled(r)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            eng.run_form_tiled(r)
        e1.record()
        torch.cuda.synchronize()
        eng.check()
        ms = e0.elapsed_time(e1) / 5
        h_cls = cls[lo:hi].cpu().numpy()
        with np.errstate(all='ignore'):
            if is_raw:
                want = oracle.evapotranspiration_raw(
                    bplut, h_cls, f64([x[lo:hi] for x in raw]), fpar[lo:hi].cpu().numpy(), lai[lo:hi].cpu().numpy(),
                    day_hours=hours[lo:hi].cpu().numpy().astype(np.float64) if form == _lib.FORM_RAW_TOTAL8_HOURS else None)
            else:
                h_drv = f64([x[lo:hi] for x in drv])
                params = oracle.gather_params(bplut, h_cls)
                sep = oracle.evapotranspiration(params, *h_drv, separate=True)
                tot = oracle.evapotranspiration(params, *h_drv)
                if form == _lib.FORM_PET:
                    want = list(tot) + list(oracle.potential_et(params, *h_drv))
                elif form == _lib.FORM_COMPONENTS:
                    want = list(sep[0]) + list(sep[1])
                else:
                    want = list(tot) + list(sep[0]) + list(sep[1])
        worst, masks = 0.0, True
        worst_abs, n_gt, n_all = 0.0, 0, 0
        tiny = float(np.finfo(np.float32).tiny)
        for o, w in zip(r.outs, want):
            got = r.flat(o, lo, hi).cpu().numpy().astype(np.float64)
            w = w.astype(eng.np_dtype).astype(np.float64)
            if mixed:
                got = np.where(np.abs(got) < tiny, 0, got)
                w = np.where(np.abs(w) < tiny, 0, w)
            masks = masks and bool(np.array_equal(np.isnan(got), np.isnan(w)) and np.array_equal(got == 0, w == 0))
            ok = np.isfinite(w) & (w != 0)
            err = np.abs(got[ok] - w[ok])
            rel = err / np.abs(w[ok])
            worst = max(worst, float(np.max(rel)))
            worst_abs = max(worst_abs, float(np.max(err) / np.abs(w[ok]).max()))
            n_gt += int(np.count_nonzero(rel > 1e-5))
            n_all += int(rel.size)
        parity = {'max_rel_err_vs_oracle': worst, 'masks_equal': masks}
        if mixed:       # the mixed form bounds the ABSOLUTE error: a tolerance check
            parity.update({'max_abs_err_over_max_value': worst_abs, 'within_abs_bound_1e-6': worst_abs <= 1e-6,
                           'fraction_rel_err_gt_1e-5': n_gt / max(n_all, 1)})
        else:
            parity['within_rtol'] = worst <= rtol
        out[name] = {'ms': ms, 'bytes_per_pixel': bpp, 'GBps': bpp * n / ms / 1e6,
                     'frac': bpp * n / ms / 1e6 / HBM_PEAK_GBPS, 'parity': parity}
        del r
        torch.cuda.empty_cache()
    return out


def n2_config(np, _lib):
    """the calibration path MOD16._et evaluated for 2048 parameter vectors
    (the sampler's loop, reference calibration.py:907, sensitivity.py:95) over 100 k pixels in
    one launch, reference-order and FAST arithmetic; numpy in, the objective (sse, count) out.
    Bound: the float64 vector pipe -- instructions per pixel-draw from the listing
    (profiles/r03_isa_mix_calibration_kernels.txt) against 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz."""
    import mod16_amd
    from oracle import mod16_oracle as oracle
    from oracle import synth
    n, ndraw = 100000, 2048
    rng = np.random.default_rng(0)
    _, drv = synth.drivers((n,), seed=1)
    lo = np.array([-10, 5, 400, 2000, 0.01, 0.01, 1e-6, 0.001, 20, 60, 50.0])
    hi = np.array([-6, 15, 1000, 5000, 0.12, 0.12, 1e-4, 0.01, 70, 120, 800.0])
    params = rng.uniform(lo, hi, (ndraw, 11))
    M = mod16_amd.MOD16
    with np.errstate(all='ignore'):
        want = [oracle.et_static(list(params[d]), *drv) for d in (0, 1, ndraw - 1)]
    obs = want[0] + rng.normal(0, 5, n)
    res = {'pixels': n, 'draws': ndraw, 'valu_peak_lane_instructions_per_s': 256 * 4 * 16 * 2.4e9}
    for name, math, per_draw in (('reference_order', _lib.MATH_EXACT, N2_INSTR['reference_order']),
                                 ('fast_unbound', _lib.MATH_FAST, N2_INSTR['fast'])):
        M._et_batch(params[:4], *drv, observed=obs, math=math)
        best = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            sse, cnt = M._et_batch(params, *drv, observed=obs, math=math)
            best = min(best, time.perf_counter() - t0)
        rows = M._et_batch(params[[0, 1, ndraw - 1]], *drv, math=math)
        worst, masks = 0.0, True
        for got, w in zip(rows, want):
            masks = masks and bool(np.array_equal(np.isnan(got), np.isnan(w)) and np.array_equal(got == 0, w == 0))
            ok = np.isfinite(w) & (w != 0)
            worst = max(worst, float(np.max(np.abs(got[ok] - w[ok]) / np.abs(w[ok]))))
        rate = n * ndraw / best
        res[name] = {'seconds': best, 'pixel_draws_per_s': rate,
                     'valu_instructions_per_pixel_draw': per_draw,
                     'frac_of_valu_issue_peak': rate * per_draw / res['valu_peak_lane_instructions_per_s'] if per_draw else None,
                     'parity': {'max_rel_err_vs_oracle': worst, 'masks_equal': masks, 'draws_checked': 3},
                     'objective_first_draw': float(sse[0] / cnt[0]),
                     'role': ('the bit-identical path: the reference\'s operation order, IEEE divide and pow' if math == _lib.MATH_EXACT else
                              'the ONE-SHOT call (drivers uploaded, workspace sized and the kernels launched one by one on '
                              'every call): kept as the comparison point -- a loop over parameter vectors uses '
                              'fast_resident (MOD16._et_bind), which is the product number')}
    # the same problem RESIDENT on the device (MOD16._et_bind): drivers up once, per evaluation the
    # parameters up and (sse, count) down around one graph launch -- what an MCMC chain would hold
    prob = M._et_bind(*drv, observed=obs, max_draws=ndraw)
    s_res, c_res = prob.objective(params)
    best = 1e30
    for _ in range(5):
        t0 = time.perf_counter()
        s_res, c_res = prob.objective(params)
        best = min(best, time.perf_counter() - t0)
    gpu_ms = prob.gpu_milliseconds(10)
    rows = prob.rows(params[[0, 1, ndraw - 1]])
    worst, masks = 0.0, True
    for got, w in zip(rows, want):
        masks = masks and bool(np.array_equal(np.isnan(got), np.isnan(w)) and np.array_equal(got == 0, w == 0))
        ok = np.isfinite(w) & (w != 0)
        worst = max(worst, float(np.max(np.abs(got[ok] - w[ok]) / np.abs(w[ok]))))
    unbound_rows = M._et_batch(params[[0, 1, ndraw - 1]], *drv, math=_lib.MATH_FAST)
    s_unb, c_unb = M._et_batch(params, *drv, observed=obs, math=_lib.MATH_FAST)
    rate = n * ndraw / best
    per_draw = N2_INSTR['fast_resident']
    res['fast_resident'] = {
        'seconds': best, 'pixel_draws_per_s': rate, 'gpu_ms': gpu_ms,
        'pixel_draws_per_s_gpu_part': n * ndraw / (gpu_ms * 1e-3),
        'valu_instructions_per_pixel_draw': per_draw,
        'frac_of_valu_issue_peak': rate * per_draw / res['valu_peak_lane_instructions_per_s'],
        'frac_of_valu_issue_peak_gpu_part': n * ndraw / (gpu_ms * 1e-3) * per_draw / res['valu_peak_lane_instructions_per_s'],
        'parity': {'max_rel_err_vs_oracle': worst, 'masks_equal': masks, 'draws_checked': 3,
                   'rows_bit_identical_to_unbound_call': bool(np.array_equal(rows, unbound_rows, equal_nan=True)),
                   'counts_equal_unbound_objective': bool(np.array_equal(c_res, c_unb)),
                   'sse_max_rel_diff_vs_unbound_objective': float(np.max(np.abs(s_res - s_unb) / np.abs(s_unb)))},
        'pixels_outside_fast_domain': prob.n_outside_domain,
        'objective_first_draw': float(s_res[0] / c_res[0]),
        'note': 'MOD16._et_bind: drivers resident; an evaluation = 180 KB of parameters up, one graph launch '
                '(parameter preparation, fused evaluation + residual reduction, the whole-array g_surf switch, '
                'final sums), 32 KB down; wall-clock of the Python call, best of 5'}
    prob.close()
    res['note'] = ('numpy in -> (sse, count) out: the 11 MB of drivers go up once per call, 32 KB come back; '
                   'bound = the float64 vector pipe, not HBM (each pixel is read once for 2048 draws)')
    return res


# VALU instructions per pixel-draw of the batched calibration kernels' inner loop (static count of
# the gfx950 listing, tools/isa_count.py; profiles/r03_isa_mix_calibration_kernels.txt)
N2_INSTR = {'reference_order': 2325, 'fast': 180, 'fast_resident': 180}   # ('fast': the one-shot call, key fast_unbound in the line)   # (resident: 166 in the draw loop + 21 / 16 per reduction pass + ~380 / 32 of per-pixel preparation)


if __name__ == '__main__':
    sys.exit(main())
