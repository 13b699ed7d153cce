"""A dynamically scheduled launch that does not process its raster must not pass for one that did
(round 4's defect: graph replays between two RCCL barriers found their ticket
counter in use, claimed nothing and left the previous step's outputs and diagnostics in place --
plausible numbers, a step twelve times too fast, status OK).

Two checks. The functional one: a captured step replayed K times back to back with the drivers
changed between the replays -- every replay's outputs and diagnostics are the ones of ITS drivers.
The loud one: every run's diagnostics partial carries the launch's serial number and what runs
behind the pipeline kernel counts the runs that carry it (kStatusIncomplete); a launch whose ticket
counter was poisoned (fault injection of the experiments build) is reported by check(), guarded
and trusted, and the launch after it is whole again."""
import pytest

pytestmark = pytest.mark.gpu

N = 32 * 1024 * 1024          # dynamic schedule: 262144 pieces, 16384+ runs, more than 8 per wave


@pytest.fixture(scope='module')
def env():
    import torch
    from mod16_amd.raster import RasterEngine
    from mod16_amd.utils import restore_bplut, bplut_table
    from mod16_amd.models import COLLECTION61_BPLUT
    return torch, RasterEngine, bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)


@pytest.mark.parametrize('layout', ['tiled', 'plain'])
def test_replays_of_a_captured_step_follow_their_drivers(env, layout):
    torch, RasterEngine, table = env
    eng = RasterEngine(table)
    K = 6
    diag = torch.zeros(8, dtype=torch.float64, device='cuda')
    if layout == 'tiled':
        ras = eng.alloc_tiled(N)
        step = eng.bind_tiled(ras, diag)
        fill = lambda k: eng.synth_tiled(ras, seed=40, step=k)
        outs = lambda: (ras.flat(ras.day, 0, 1 << 20), ras.flat(ras.night, N - (1 << 20), N))
    else:
        cls, drv, day, night = eng.alloc_raster(N)
        step = eng.bind(cls, drv, day, night, diag, graph=True)
        fill = lambda k: eng.synth(N, seed=40, step=k, out=(cls, drv))
        outs = lambda: (day[:1 << 20].clone(), night[N - (1 << 20):].clone())
    # the reference: every step on its own, synchronised, direct launches (no graph)
    want = []
    for k in range(K):
        fill(k)
        if layout == 'tiled':
            eng.run_tiled(ras, diag=diag)
        else:
            eng.run(cls, drv, day, night, diag=diag)
        eng.check()
        want.append((diag.clone(), ) + outs())
    # K replays queued back to back, the generator between them, nothing waited for
    torch.cuda.synchronize()
    got = []
    for k in range(K):
        fill(k)
        step()
        got.append((diag.clone(), ) + outs())
    torch.cuda.synchronize()
    eng.check()
    for k in range(K):
        for a, b in zip(got[k], want[k]):
            assert torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)), (layout, k)
        assert got[k][0][2] + got[k][0][4] == N          # every pixel counted: valid + NaN
    for k in range(1, K):                                # ... and the steps do differ
        assert not torch.equal(got[k][0], got[k - 1][0])


@pytest.mark.parametrize('trusted', [False, True])
def test_a_launch_that_found_its_ticket_in_use_is_reported(env, trusted, monkeypatch):
    torch, RasterEngine, table = env
    from mod16_amd import _lib
    good = RasterEngine(table, trusted=trusted)
    monkeypatch.setenv('MOD16_POISON_TICKET', '3')      # read when the context is created: its 3rd dynamic launch
    eng = RasterEngine(table, trusted=trusted, experiments=True)
    monkeypatch.delenv('MOD16_POISON_TICKET')
    ras = eng.synth_tiled(eng.alloc_tiled(N), seed=41)
    diag = torch.zeros(8, dtype=torch.float64, device='cuda')
    ref = torch.zeros(8, dtype=torch.float64, device='cuda')
    good.run_tiled(ras, diag=ref)
    good.check()
    day_ref = ras.flat(ras.day)
    # two whole launches first: the workspace then holds the partials of a COMPLETE launch of the
    # same shape -- what the poisoned launch leaves in place for the runs it does not reach
    for _ in range(2):
        eng.run_tiled(ras, diag=diag)
        eng.check()
        assert torch.equal(diag, ref)
    ras.day.zero_()
    eng.run_tiled(ras, diag=diag)                        # its ticket counter is poisoned: processes its first runs only
    with pytest.raises(_lib.Mod16Error, match='only part of its raster'):
        eng.check()
    assert (ras.flat(ras.day) == 0).sum() > N // 2       # most of the raster untouched: what went unnoticed in round 4
    eng.run_tiled(ras, diag=diag)                        # the poisoned launch left the counter at zero: whole again
    eng.check()
    assert torch.equal(diag, ref)
    assert torch.equal(torch.nan_to_num(ras.flat(ras.day), nan=-7.0), torch.nan_to_num(day_ref, nan=-7.0))
    # (the numpy path -- HOST mode -- stages tiles of 2^21 pixels, which are dealt out statically:
    # it never reads a ticket counter)


@pytest.mark.parametrize('byte', [0x3f, 0xff])
def test_no_ticket_value_sends_the_kernel_outside_its_raster(env, byte, monkeypatch):
    """Round 5's first fault injection wrote 0x3f3f... into the ticket: (nwaves + ticket) << run_shift
    overflowed into a negative base, the loop guard held, and the kernel read and STORED through a
    wild piece index (a GPU memory fault). The kernel now compares the ticket unsigned with the
    raster's pieces before it forms a base: whatever 64 bits sit in the counter, the launch ends,
    writes nothing outside its raster and is reported as incomplete. (0xff...: -1.)"""
    torch, RasterEngine, table = env
    from mod16_amd import _lib
    monkeypatch.setenv('MOD16_POISON_TICKET', '2')
    monkeypatch.setenv('MOD16_POISON_BYTE', str(byte))
    eng = RasterEngine(table, experiments=True)
    monkeypatch.delenv('MOD16_POISON_TICKET')
    monkeypatch.delenv('MOD16_POISON_BYTE')
    # the raster between two guard bands of the same allocation class: a wild store of the kind
    # round 5 saw would most likely land in a neighbour of the raster's slab
    before = torch.full((1 << 22,), 7.25, dtype=torch.float64, device='cuda')
    ras = eng.synth_tiled(eng.alloc_tiled(N), seed=43)
    after = torch.full((1 << 22,), 7.25, dtype=torch.float64, device='cuda')
    diag = torch.zeros(8, dtype=torch.float64, device='cuda')
    ref = torch.zeros(8, dtype=torch.float64, device='cuda')
    eng.run_tiled(ras, diag=ref)
    eng.check()
    day_ref = ras.flat(ras.day)
    ras.day.zero_()
    eng.run_tiled(ras, diag=diag)                        # the poisoned one
    torch.cuda.synchronize()                             # it ENDS
    if byte == 0xff:
        # -1: the wave that draws it stops claiming, the counter continues at 0 and the others take
        # every run -- a complete launch
        eng.check()
        assert torch.equal(diag, ref)
    else:
        with pytest.raises(_lib.Mod16Error, match='only part of its raster'):
            eng.check()
        assert (ras.flat(ras.day) == 0).sum() > N // 2
    assert bool((before == 7.25).all()) and bool((after == 7.25).all())
    eng.run_tiled(ras, diag=diag)                        # the counter is at zero again: whole
    eng.check()
    assert torch.equal(diag, ref)
    assert torch.equal(torch.nan_to_num(ras.flat(ras.day), nan=-7.0), torch.nan_to_num(day_ref, nan=-7.0))


def test_a_graph_whose_context_is_gone_is_refused_not_replayed(env):
    """a captured graph's kernels read the context's tables and write its status
    word. mod16_destroy marks the context's live graphs dead: a replay returns MOD16_ERR_ARG (nothing
    is launched -- the raster keeps the last good step), mod16_graph_destroy still frees the graph;
    a graph destroyed before its context leaves the context's list."""
    torch, RasterEngine, table = env
    from mod16_amd import _lib
    eng = RasterEngine(table)
    n = 1 << 22
    ras = eng.alloc_tiled(n)
    eng.synth_tiled(ras, seed=7)
    diag = torch.zeros(8, dtype=torch.float64, device='cuda')
    first, second = eng.bind_tiled(ras, diag), eng.bind_tiled(ras, diag)
    first()
    torch.cuda.synchronize()
    eng.ctx.check(0)
    want = ras.flat(ras.day, 0, 1 << 16).clone()
    import ctypes
    stream = ctypes.c_void_p(torch.cuda.current_stream(eng.device).cuda_stream)
    lib = eng.ctx.lib
    assert lib.mod16_graph_destroy(first._graph.handle) == _lib.OK          # graph first
    first._graph.handle = None
    eng.ctx.close()                                           # then the context, under a live graph
    ras.day.zero_()
    torch.cuda.synchronize()
    assert lib.mod16_graph_launch(second._graph.handle, stream) == _lib.ERR_ARG
    torch.cuda.synchronize()
    assert not ras.day.any()                                  # nothing ran
    assert lib.mod16_graph_destroy(second._graph.handle) == _lib.OK
    second._graph.handle = None
    # a fresh context on the same raster: whole again
    eng2 = RasterEngine(table)
    eng2.run_tiled(ras, diag)
    torch.cuda.synchronize()
    eng2.check()
    assert torch.equal(ras.flat(ras.day, 0, 1 << 16).view(torch.int64), want.view(torch.int64))     # (bits: NaN pixels too)
