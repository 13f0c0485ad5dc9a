import sys, torch, numpy as np
sys.path.insert(0, '.')
import kraken_amd
from kraken_amd.engine import RecognitionEngine
from kraken_amd.specs import bench_codec
from oracle.torch_port import CpuRecognizer
SPEC = ('[1,120,0,1 Cr3,13,32 Do0.1,2 Mp2,2 Cr3,13,32 Do0.1,2 Mp2,2 Cr3,9,64 Do0.1,2 Mp2,2 Cr3,9,64 Do0.1,2 S1(1x0)1,3 '
        'Lbx200 Do0.1,2 Lbx200 Do0.1,2 Lbx200 Do O1c256]')
torch.manual_seed(0)
m = kraken_amd.TorchVGSLModel(vgsl=SPEC, codec=bench_codec()).to('cuda')
m.nn.set_precision('bf16x3')
N, W = 256, 1200
x = torch.rand(N, 1, 120, W, generator=torch.Generator().manual_seed(1)).cuda()
eng = RecognitionEngine(m, device=0, max_batch=N, max_width=W, slots=1)
eng.set_profiling(True)
best = None
for r in range(5):
    eng.submit(x); eng.collect()
    t = [(n_, ms) for n_, ms, _ in eng.layer_times()[0]]
    if r: best = t if best is None else [(a[0], min(a[1], b[1])) for a, b in zip(best, t)]
print('height 120 alone: total %.3f ms |' % sum(ms for _, ms in best), ' '.join('%s=%.3f' % kv for kv in best), flush=True)
eng.close()
eng = RecognitionEngine(m, device=0, max_batch=N, max_width=W, slots=3)
import time
xs = [torch.rand(N, 1, 120, W, device='cuda') for _ in range(3)]
for i in range(12):
    if eng.free_slots() == 0: eng.collect()
    eng.submit(xs[i % 3])
while eng.free_slots() < 3: eng.collect()
torch.cuda.synchronize(); t0 = time.perf_counter(); K = 60
for i in range(K):
    if eng.free_slots() == 0: m.codec.decode_strings(eng.collect()[0])
    eng.submit(xs[i % 3])
while eng.free_slots() < 3: m.codec.decode_strings(eng.collect()[0])
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('height 120 pipelined: %.1f lines/s, %.3f ms/step' % (N * K / dt, 1e3 * dt / K), flush=True)
y, _ = m.nn(x[:4])
want, _ = CpuRecognizer(m.layer_specs, {k: v.cpu() for k, v in m.state_dict().items()}).forward(x[:4].cpu())
print('max |d logit| vs the CPU oracle (4 lines): %.2e' % (y.cpu() - want).abs().max().item())
