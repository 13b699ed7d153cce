'''
Tile sharding of a global raster over the GPUs of one node.

The ET computation is an independent map over pixels: no halo, no exchange.
The row-major raster is cut into contiguous row bands, one per rank (one
process per GPU); the only collective is an all-reduce of the 8-double
diagnostics vector (RCCL over xGMI on GPUs, gloo in the CPU tests). Outputs
stay sharded.
'''

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


def allreduce_diag(diag, group=None):
    '''In-place all-reduce of a diagnostics vector (``raster.DIAG_FIELDS``):
    sums and counts [0:6] with SUM, maxima [6:8] with MAX. ``diag`` is a
    float64 tensor of 8 on the device the process group's backend serves.'''
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or \
            dist.get_world_size(group) == 1:
        return diag
    dist.all_reduce(diag[0:6], op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(diag[6:8], op=dist.ReduceOp.MAX, group=group)
    return diag
