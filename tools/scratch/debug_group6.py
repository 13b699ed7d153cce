import os, sys, time, socket
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mode = sys.argv[1]
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
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
main = torch.cuda.Stream() if 'own' in mode else torch.cuda.current_stream()
comm = torch.cuda.Stream()
produced = [torch.cuda.Event() for _ in range(2)]
reduced = [torch.cuda.Event() for _ in range(2)]
cnt = [0]
def step(ev=None):
    k = cnt[0] & 1; cnt[0] += 1
    with torch.cuda.stream(main):
        if 'nowait' not in mode: main.wait_event(reduced[k])
        if ev: ev[0].record(main)
        if 'direct' in mode: eng.run_tiled(ras, diag=diags[k])
        else: bound[k]()
        if ev: ev[1].record(main)
        if 'noprod' not in mode: produced[k].record(main)
    if 'nocomm' not in mode:
        with torch.cuda.stream(comm):
            if 'noprod' not in mode: comm.wait_event(produced[k])
            reduced[k].record(comm)
for _ in range(3): step()
if 'earlybarrier' in mode:
    dist.barrier()
torch.cuda.synchronize()
for trail in ('none', 'barrier', 'streamsync', 'barrier'):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(NS)]
    if 'nopoison' not in mode:
        ras.day.fill_(-1.0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(NS): step(ev[i])
    with torch.cuda.stream(main):
        if trail == 'barrier': dist.barrier()
        elif trail == 'streamsync': main.synchronize()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    left = int((ras.day == -1.0).sum())
    print(mode, 'trail=%s' % trail, 'to trail end %.2f ms, device sync +%.2f ms' % (1e3 * (t1 - t0), 1e3 * (t2 - t1)),
          'event ms', ['%.2f' % a.elapsed_time(b) for a, b in ev], 'poisoned left', left, flush=True)
dist.destroy_process_group()
