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
own = torch.cuda.Stream()
cnt = [0]
def run(use_own):
    ctx = torch.cuda.stream(own) if use_own else torch.cuda.stream(torch.cuda.default_stream())
    with ctx:
        main = torch.cuda.current_stream()
        for _ in range(3):
            bound[cnt[0] & 1](); cnt[0] += 1
        torch.cuda.synchronize()
        for trail in ('none', 'barrier', 'streamsync', 'barrier'):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
            t0 = time.perf_counter()
            for i in range(8):
                ev[i][0].record(main); bound[cnt[0] & 1](); cnt[0] += 1; ev[i][1].record(main)
            if trail == 'barrier': dist.barrier()
            elif trail == 'streamsync': main.synchronize()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            x = float(diags[0][0].item())
            t3 = time.perf_counter()
            print(mode, 'own-stream' if use_own else 'null-stream', 'trail=%s' % trail,
                  'to trail end %.2f ms, device sync +%.2f ms, item +%.2f ms' % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)),
                  'event ms', ['%.2f' % a.elapsed_time(b) for a, b in ev], flush=True)
run(False)
run(True)
dist.destroy_process_group()
