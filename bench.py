#!/usr/bin/env python3
"""
bench.py -- the MOD16 forward-run benchmark (BASELINE.json metric: pixels/s and
achieved HBM GB/s on the 43200 x 21600 global ET grid, float64).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" is one pass of the hot path over the (synthetic, already
HBM-resident) drivers of one time step: the fused ET kernel over this rank's
row band, which also reduces its outputs to the diagnostics vector
(deterministic two-level sum), and -- for N > 1 -- the RCCL all-reduce of that
8-double vector. The global
grid is fixed and cut into N row bands (one process per GPU), so total work is
fixed: "scaling": "strong".

Rank 0 prints ONE JSON line. Besides the contract fields it carries
  roofline      dominant kernel (fused ET) vs the 8 TB/s HBM peak; `achieved`
                = 129 B/pixel (float64) x pixels per launch / mean launch time,
                timed with HIP events on the launch stream (mod16_time_et);
  cpu_baseline  the numpy oracle (reference-shaped port) timed on this box's
                host cores on 1200 x 1200 tiles of the same synthetic workload
                (N = 1 only);
  parity        GPU outputs vs the oracle on a 1200 x 1200 tile copied back.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E peak (MI355X_MICROARCH.md)
PMC_TRAFFIC = [os.path.join(ROOT, 'profiles', f) for f in
               ('r01j_pmc_hbm_traffic.json', 'r01i_pmc_hbm_traffic_float32.json')]
TILE = (1200, 1200)           # BASELINE.json configs[1], the CPU sample unit
SEED = 16


def cpu_tile_seconds(reps):
    """Worker of the CPU baseline: time `reps` oracle runs on one tile."""
    import numpy as np
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
    single-threaded), then a pool with one tile per core. Runs before this
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
    out['global_grid_seconds_1core'] = 43200 * 21600 / out['value']
    try:
        with open('/proc/cpuinfo') as f:
            models = [l.split(':', 1)[1].strip() for l in f if l.startswith('model name')]
        out['cpu_model'] = models[0] if models else None
        out['host_logical_cpus'] = os.cpu_count()
    except OSError:
        pass
    return out


def pmc_traffic(pixels_per_launch, dtype):
    """(HBM bytes per launch, source file) of the dominant kernel from the committed
    rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, collected separately, see the
    files); counters cannot be read from inside this process, so a figure applies
    only when this run launches the same kernel on the same pixel count. The float32
    record was taken with the mixed-precision form; the FAST form moves the same bytes."""
    for path in PMC_TRAFFIC:
        try:
            with open(path) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        if rec.get('pixels_per_launch') == pixels_per_launch and rec.get('dtype') == dtype:
            return rec['traffic_bytes_per_launch'], os.path.relpath(path, ROOT)
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--rows', type=int, default=21600, help='global raster rows')
    ap.add_argument('--cols', type=int, default=43200, help='global raster columns')
    ap.add_argument('--dtype', default='float64', choices=['float64', 'float32'])
    ap.add_argument('--math', default='fast', choices=['fast', 'exact', 'mixed'],
                    help="'mixed': the mixed-precision form for --dtype float32 (configs[4])")
    ap.add_argument('--no-graph', action='store_true',
                    help='enqueue the step kernel by kernel instead of replaying its HIP graph')
    ap.add_argument('--no-tune', action='store_true',
                    help='arrays of the raster slab back to back instead of the measured best spacing')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true')
    ap.add_argument('--cpu-workers', type=int, default=16)
    ap.add_argument('--time-steps', type=int, default=0,
                    help='also time a streamed series of this many steps (BASELINE configs[3])')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed.run '
                     '--nproc-per-node %d' % (args.gpus, args.gpus))
        args.gpus = world

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
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if rehearsal:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local_rank))

    table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
    math = {'fast': _lib.MATH_FAST, 'exact': _lib.MATH_EXACT, 'mixed': _lib.MATH_MIXED}[args.math]
    eng = RasterEngine(table, device=local_rank, dtype=args.dtype, math=math)
    offset, n = tiles.pixel_range(args.rows, args.cols, rank, world)
    total = args.rows * args.cols

    # one slab; the spacing between its arrays is chosen by measurement at set-up
    # (outside the timed region; RasterEngine.alloc_raster_tuned)
    if args.no_tune:
        cls, drv, day, night = eng.alloc_raster(n)
        layout = {'chosen_extra_bytes': 0}
    else:
        try:
            (cls, drv, day, night), layout = eng.alloc_raster_tuned(n)
        except RuntimeError as exc:      # no room for the candidates' slack: back-to-back layout
            torch.cuda.empty_cache()
            cls, drv, day, night = eng.alloc_raster(n)
            layout = {'chosen_extra_bytes': 0, 'tuning_failed': str(exc)[:200]}
    eng.synth(n, seed=SEED, step=0, pixel_offset=offset, out=(cls, drv))
    # ET + diagnostics in one pass. Two diagnostics vectors: the all-reduce of
    # step s runs on a side stream under the kernel of step s + 1 (for N > 1;
    # the kernel of step s + 2, which reuses the vector, waits for it).
    diags = [torch.zeros(8, dtype=torch.float64, device='cuda') for _ in range(2)]
    launches = [eng.bind(cls, drv, day, night, d, graph=not args.no_graph) for d in diags]
    diag, launch = diags[0], launches[0]
    main_stream = torch.cuda.current_stream()
    comm_stream = torch.cuda.Stream() if world > 1 else None
    produced = [torch.cuda.Event() for _ in range(2)]
    reduced = [torch.cuda.Event() for _ in range(2)]
    counter = [0]

    def step():
        k = counter[0] & 1
        counter[0] += 1
        if comm_stream is None:
            launches[k]()
            return
        main_stream.wait_event(reduced[k])          # the all-reduce that last used this vector
        launches[k]()
        produced[k].record(main_stream)
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(produced[k])
            tiles.allreduce_diag(diags[k])
            reduced[k].record(comm_stream)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    eng.check()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant kernel alone, HIP events on its own stream
    kernel_ms = eng.time_kernel(cls, drv, day, night, launches=max(3, min(args.steps, 20)),
                                diag=diag)
    bpp = eng.bytes_per_pixel
    achieved = bpp * n / (kernel_ms * 1e-3) / 1e9
    torch.cuda.synchronize()
    tiles.allreduce_diag(diag)          # the timing launches left this rank's band in it
    diag_host = diag.cpu().numpy()

    parity = None
    if rank == 0 and not args.no_parity:
        from oracle import mod16_oracle as oracle
        # 1200 x 1200-pixel samples of this band (start, two inside, end), inputs
        # AND outputs copied back; the oracle runs on exactly those input bits
        m = min(n, TILE[0] * TILE[1])
        starts = sorted(set([0, (n // 3) // 4 * 4, (2 * n // 3) // 4 * 4, n - m]))
        bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
        worst, masks = 0.0, True
        for s0 in starts:
            h_cls = cls[s0:s0 + m].cpu().numpy()
            h_drv = [d[s0:s0 + m].cpu().numpy() for d in drv]
            # float32 data: the kernel widens, computes in float64 and rounds
            # once, so the checker is the float64 oracle on the widened inputs
            want = oracle.evapotranspiration_raster(
                bplut, h_cls, *[d.astype(np.float64) for d in h_drv])
            want = [w.astype(h_drv[0].dtype) for w in want]
            for got, ref in ((day[s0:s0 + m].cpu().numpy(), want[0]),
                             (night[s0:s0 + m].cpu().numpy(), want[1])):
                masks = masks and bool(np.array_equal(np.isnan(got), np.isnan(ref))
                                       and np.array_equal(got == 0, ref == 0))
                ok = np.isfinite(ref) & (ref != 0)
                worst = max(worst, float(np.max(
                    np.abs(got[ok].astype(np.float64) - ref[ok]) / np.abs(ref[ok]))))
        parity = {'pixels': int(m * len(starts)), 'tiles': len(starts), 'max_rel_err': worst,
                  'masks_equal': masks, 'rtol_north_star': 1e-5,
                  'against': 'numpy oracle on the same input bits'}
    if not args.no_parity and args.math == 'mixed':
        # configs[4], every pixel of the band: the mixed-precision form against the
        # float64 arithmetic on the same float32 rasters (FAST: float64 result
        # rounded once), on the device
        ref_eng = RasterEngine(table, device=local_rank, dtype=args.dtype, math=_lib.MATH_FAST)
        rday, rnight = ref_eng.run(cls, drv)
        ref_eng.check()
        full = {'pixels': int(n), 'nan_masks_equal': True, 'zero_mask_mismatches': 0, 'max_rel_err': 0.0,
                'max_abs_err_over_max_value': 0.0, 'n_rel_err_gt_1e-6': 0, 'n_rel_err_gt_1e-5': 0,
                'n_rel_err_gt_1e-4': 0, 'n_rel_err_gt_1e-3': 0}
        for got, ref in ((day, rday), (night, rnight)):
            full['nan_masks_equal'] &= bool(torch.equal(torch.isnan(got), torch.isnan(ref)))
            full['zero_mask_mismatches'] += int(((got == 0) != (ref == 0)).sum())
            scale = float(torch.nan_to_num(ref).abs().max())
            err = (got.double() - ref.double()).abs_()
            full['max_abs_err_over_max_value'] = max(full['max_abs_err_over_max_value'],
                                                     float(torch.nan_to_num(err).max()) / scale)
            err = torch.nan_to_num_(err.div_(ref.double().abs_()), nan=0.0, posinf=0.0)
            full['max_rel_err'] = max(full['max_rel_err'], float(err.max()))
            for thr in ('1e-6', '1e-5', '1e-4', '1e-3'):
                full['n_rel_err_gt_' + thr] += int((err > float(thr)).sum())
            del err
        del rday, rnight
        if parity is not None:
            parity['full_grid_mixed_vs_float64_arithmetic'] = full
    if not args.no_parity and args.math == 'fast':
        # every pixel of the band: the production kernel against the kernel that
        # keeps the reference's operation order (IEEE divide / pow), on the device
        exact = RasterEngine(table, device=local_rank, dtype=args.dtype, math=_lib.MATH_EXACT)
        eday, enight = exact.run(cls, drv)
        exact.check()
        full = {'pixels': int(n), 'nan_masks_equal': True, 'zero_masks_equal': True,
                'max_rel_err': 0.0, 'n_rel_err_gt_1e-9': 0, 'n_rel_err_gt_1e-5': 0}
        for got, ref in ((day, eday), (night, enight)):
            full['nan_masks_equal'] &= bool(torch.equal(torch.isnan(got), torch.isnan(ref)))
            full['zero_masks_equal'] &= bool(torch.equal(got == 0, ref == 0))
            err = (got - ref).abs_().div_(ref.abs())
            err = torch.nan_to_num_(err, nan=0.0, posinf=0.0)     # masked pixels: 0/0, x/0
            full['max_rel_err'] = max(full['max_rel_err'], float(err.max()))
            full['n_rel_err_gt_1e-9'] += int((err > 1e-9).sum())
            full['n_rel_err_gt_1e-5'] += int((err > 1e-5).sum())
            del err
        del eday, enight
        if world > 1:
            t = torch.tensor([full['max_rel_err'], -float(full['nan_masks_equal']),
                              -float(full['zero_masks_equal'])], dtype=torch.float64, device='cuda')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            c = torch.tensor([full['n_rel_err_gt_1e-9'], full['n_rel_err_gt_1e-5'], n],
                             dtype=torch.float64, device='cuda')
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            full.update(max_rel_err=float(t[0]), nan_masks_equal=bool(t[1] == -1),
                        zero_masks_equal=bool(t[2] == -1), pixels=int(c[2]))
            full['n_rel_err_gt_1e-9'], full['n_rel_err_gt_1e-5'] = int(c[0]), int(c[1])
        if parity is not None:
            parity['full_grid_fast_vs_exact_kernel'] = full

    series = None
    if args.time_steps > 0:
        # configs[3]: drivers of step s+1 produced on a second stream into a
        # two-slot ring while the kernel works on step s (producer = the
        # on-device generator standing in for an ingest stage)
        # everything that still refers into the slab: the bound launch keeps its
        # tensors alive, the parity loops leave views behind
        got = ref = h_cls = h_drv = None
        del cls, drv, day, night, launch, launches, step
        torch.cuda.empty_cache()
        bufs = eng.alloc_series(n, layout['chosen_extra_bytes'])
        eng.run_series(n, 2, seed=SEED, pixel_offset=offset, buffers=bufs)   # warm-up
        fence()
        t0 = time.perf_counter()
        sdiag, _, _ = eng.run_series(n, args.time_steps, seed=SEED, pixel_offset=offset,
                                     buffers=bufs)
        if world > 1:
            for s in range(args.time_steps):
                tiles.allreduce_diag(sdiag[s])
        fence()
        t_series = time.perf_counter() - t0
        eng.check()
        if world > 1:
            t = torch.tensor([t_series], dtype=torch.float64, device='cuda')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            t_series = float(t.item())
        series = {'steps': args.time_steps, 'seconds': t_series,
                  'pixels_per_s': total * args.time_steps / t_series,
                  'note': 'includes producing every step\'s 14 driver arrays on the device '
                          '(113 B/pixel written by the generator on a second stream)',
                  'sum_day_first_last': [float(sdiag[0, 0]), float(sdiag[-1, 0])]}

    if rank == 0:
        traffic, traffic_source = pmc_traffic(n, args.dtype) if args.math != 'exact' else (None, None)
        value = total * args.steps / elapsed
        line = {
            'metric': 'pixels/sec, fused Penman-Monteith ET forward run (day+night), 43200x21600 global grid',
            'value': value, 'unit': 'pixels/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f64' if args.dtype == 'float64' else 'f32', 'data': 'synthetic',
            'config': {
                'workload': '%dx%d global ET grid, one timestep, %s, %d row band(s) of %d-%d rows'
                            % (args.cols, args.rows, args.dtype, world,
                               args.rows // world, -(-args.rows // world)),
                'pixels': total, 'pixels_per_gpu': n, 'parallelism': 'tile-dp%d' % world,
                'math': args.math, 'bplut': os.path.basename(COLLECTION61_BPLUT),
                'step': 'fused ET kernel with in-kernel diagnostics + fixed-order final sum (one HIP graph launch) + '
                        'all-reduce(8 doubles) overlapped with the next step on a side stream',
                'slab_layout': dict(layout, note='rank 0; set-up, not timed: spacing between the arrays '
                                                 'of the raster slab chosen by measurement'),
            },
            'roofline': {
                'bound': 'hbm', 'kernel': 'et_stream_kernel<%s, %s> (LDS-DMA, dynamic runs, in-kernel diagnostics)'
                                          % (args.dtype, 'totals, mixed precision' if args.math == 'mixed' else 'totals'), 'achieved': achieved,
                'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBPS,
                'traffic': traffic, 'traffic_source': traffic_source, 'traffic_unit': 'bytes per launch',
                'bytes_per_pixel': bpp, 'pixels_per_launch': n,
                'kernel_ms': kernel_ms, 'kernel_pixels_per_s': n / (kernel_ms * 1e-3),
            },
            'cpu_baseline': cpu,
            'series': series,
            'parity': parity,
            'diagnostics': dict(zip(
                ('sum_day', 'sum_night', 'n_valid_day', 'n_valid_night',
                 'n_nan_day', 'n_nan_night', 'max_day', 'max_night'),
                [float(v) for v in diag_host])),
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
