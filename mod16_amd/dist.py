'''
Tile sharding of a global raster over the GPUs of one node.

The ET computation is an independent map over pixels: no halo, no exchange.
The row-major raster is cut into contiguous row bands, one per rank (one
process per GPU); the only collective is one all-gather of the 8-double
diagnostics vector per step (RCCL over xGMI on GPUs, gloo in the CPU tests),
reduced in rank order by every rank. Outputs
stay sharded.
'''

import threading

GLOBAL_ROWS = 21600
GLOBAL_COLS = 43200


def band(rows, rank, world):
    '''Row range [r0, r1) of ``rank``: bands differ by at most one row and
    tile [0, rows) exactly, in rank order.'''
    if not 0 <= rank < world:
        raise ValueError('rank %d outside world of %d' % (rank, world))
    base, extra = divmod(rows, world)
    r0 = rank * base + min(rank, extra)
    return r0, r0 + base + (1 if rank < extra else 0)


def pixel_range(rows, cols, rank, world):
    '''(pixel_offset, n_pixels) of the band of ``rank`` in the flattened
    row-major raster.'''
    r0, r1 = band(rows, rank, world)
    return r0 * cols, (r1 - r0) * cols


_GATHER = threading.local()       # per thread: (device, world) -> the gather buffer


def allreduce_diag(diag, group=None, engine=None):
    '''In-place reduction of a diagnostics vector (``raster.DIAG_FIELDS``) over
    the ranks: sums and counts [0:6] added, maxima [6:8] maximised. ``diag`` is
    a float64 tensor of 8 on the device the process group's backend serves.

    ONE collective per call -- an all-gather of the 8 doubles (64 bytes per
    rank: latency-bound whatever the algorithm) -- and the reduction itself is
    done by every rank on the gathered ``(world, 8)`` block in RANK ORDER, so
    the global sums are the same bits on every rank and from run to run,
    whatever order the collective library moves the pieces in. With ``engine`` (a
    ``RasterEngine`` on the tensor's device) that fold is one library kernel
    (``mod16_fold_diag``) behind the gather; without, a handful of torch operations
    (CPU tensors under gloo). A group of ONE still runs the collective: whatever the
    process group is, is what a step executes (``bench.py`` rehearses the N > 1 step on
    one GPU that way).'''
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return diag
    world = dist.get_world_size(group)
    key = (diag.device, world)
    table = getattr(_GATHER, 'buffers', None)
    if table is None:
        table = _GATHER.buffers = {}
    buf = table.get(key)
    if buf is None:
        buf = table[key] = torch.empty(world * 8, dtype=torch.float64, device=diag.device)
    dist.all_gather_into_tensor(buf, diag, group=group)
    if engine is not None and diag.is_cuda:
        return engine.fold_ranks(buf, world, diag)
    buf = buf.view(world, 8)
    acc = buf[0, 0:6].clone()
    for r in range(1, world):          # fixed order: rank 0 + rank 1 + ...
        acc += buf[r, 0:6]
    diag[0:6] = acc
    diag[6:8] = buf[:, 6:8].max(dim=0).values
    return diag
