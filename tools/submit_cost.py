import sys, time, torch
sys.path.insert(0, '.')
import kraken_amd
from kraken_amd.engine import RecognitionEngine
from kraken_amd.specs import BENCH_A, bench_codec
torch.manual_seed(0)
m = kraken_amd.TorchVGSLModel(vgsl=BENCH_A, codec=bench_codec()).to('cuda')
m.nn.set_precision('bf16x3')
eng = RecognitionEngine(m, device=0, max_batch=256, max_width=1200, slots=3)
x = torch.rand(256, 1, 48, 1200, device='cuda')
for _ in range(6):
    eng.submit(x); eng.collect()
ts, tc = [], []
for _ in range(30):
    t0 = time.perf_counter(); eng.submit(x); t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter(); eng.collect(); t3 = time.perf_counter()
    ts.append(t1 - t0); tc.append(t3 - t2)
ts.sort(); tc.sort()
print(f'submit (enqueue of one 256-line batch): median {1e6*ts[15]:.0f} us, min {1e6*ts[0]:.0f} us; collect after completion: median {1e6*tc[15]:.0f} us')
