import os, sys, time, socket
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mode = sys.argv[1]     # none | gather | gather_fold | events_only
os.environ.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1')
with socket.socket() as s:
    s.bind(('127.0.0.1', 0)); os.environ['MASTER_PORT'] = str(s.getsockname()[1])
import torch, torch.distributed as dist
from mod16_amd import _lib, dist as tiles
from mod16_amd.raster import RasterEngine
from mod16_amd.utils import restore_bplut, bplut_table
from mod16_amd.models import COLLECTION61_BPLUT
torch.cuda.set_device(0)
if mode != 'none':
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
eng = RasterEngine(table)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5400 * 43200
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
    main.wait_event(reduced[k])
    if poison:
        ras.day.fill_(-1.0)
    if ev: ev[0].record(main)
    bound[k]()
    if ev: ev[1].record(main)
    produced[k].record(main)
    with torch.cuda.stream(comm):
        comm.wait_event(produced[k])
        if mode in ('gather', 'barrier_gather'):
            tiles.allreduce_diag(diags[k])
        elif mode in ('gather_fold', 'barrier_gather_fold'):
            tiles.allreduce_diag(diags[k], engine=eng)
        reduced[k].record(comm)
for _ in range(3): step()
if 'barrier' in mode:
    dist.barrier()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
t0 = time.perf_counter()
for i in range(12): step(ev[i])
if 'barrier' in mode:
    dist.barrier()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
t1 = time.perf_counter(); x = float(diags[0].sum().item()); t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
print('after: item +%.2f ms, device sync +%.2f ms' % (1e3 * (t2 - t1), 1e3 * (t3 - t2)), flush=True)
print(mode, 'wall ms/step %.3f' % (1e3 * dt / 12), 'event ms', ['%.2f' % a.elapsed_time(b) for a, b in ev], flush=True)
# does a step really overwrite its outputs?
for i in range(4):
    step(poison=True)
    torch.cuda.synchronize()
    print(mode, 'poisoned pixels left', int((ras.day == -1.0).sum()), 'diag n_valid', float(diags[(cnt[0] - 1) & 1][2]), flush=True)
if mode != 'none':
    dist.destroy_process_group()
