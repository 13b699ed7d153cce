import os, sys, time, socket
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mode = sys.argv[1]
os.environ.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1')
with socket.socket() as s:
    s.bind(('127.0.0.1', 0)); os.environ['MASTER_PORT'] = str(s.getsockname()[1])
import torch, torch.distributed as dist
from mod16_amd import _lib, dist as tiles
from mod16_amd.raster import RasterEngine
from mod16_amd.utils import restore_bplut, bplut_table
from mod16_amd.models import COLLECTION61_BPLUT
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
eng = RasterEngine(table)
n = 5400 * 43200
ras = eng.synth_tiled(eng.alloc_tiled(n), seed=16)
diags = [torch.zeros(8, dtype=torch.float64, device='cuda') for _ in range(2)]
bound = [eng.bind_tiled(ras, d) for d in diags]
main = torch.cuda.current_stream()
comm = torch.cuda.Stream()
produced = [torch.cuda.Event() for _ in range(2)]
reduced = [torch.cuda.Event() for _ in range(2)]
cnt = [0]
def step(ev=None, poison=False):
    k = cnt[0] & 1; cnt[0] += 1
    if 'nowait' not in mode:
        main.wait_event(reduced[k])
    if poison: ras.day.fill_(-1.0)
    if ev: ev[0].record(main)
    if 'direct' in mode: eng.run_tiled(ras, diag=diags[k])
    else: bound[k]()
    if ev: ev[1].record(main)
    produced[k].record(main)
    if 'nocomm' not in mode:
        with torch.cuda.stream(comm):
            comm.wait_event(produced[k])
            if 'gather' in mode: tiles.allreduce_diag(diags[k], engine=eng)
            reduced[k].record(comm)
for _ in range(3): step()
dist.barrier()
torch.cuda.synchronize()
def loop(tag, poison):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
    t0 = time.perf_counter()
    for i in range(8): step(ev[i], poison=poison)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(mode, tag, 'wall ms/step %.3f' % (1e3 * dt / 8), 'event ms', ['%.2f' % a.elapsed_time(b) for a, b in ev],
          'poisoned left', int((ras.day == -1.0).sum()), flush=True)
loop('plain', False)
ras.day.fill_(-1.0); torch.cuda.synchronize()
loop('after-one-poison', False)
loop('poison-each', True)
dist.destroy_process_group()
