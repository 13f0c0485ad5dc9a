"""Dev tool (GPU): launch groups (alone) and pipelined lines/s of ANY recognition spec on 256 synthetic lines -- which kernels a
real-world spec lands on.  python tools/spec_probe.py "<vgsl spec>" [--w 1200] [--n 256] [--precision bf16x3]"""
import sys
import time

sys.path.insert(0, '.')
import torch  # noqa: E402

import kraken_amd  # noqa: E402
from kraken_amd.engine import RecognitionEngine  # noqa: E402
from kraken_amd.specs import bench_codec  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith('--')]
opts = {a.split('=')[0][2:]: (a.split('=') + ['1'])[1] for a in sys.argv[1:] if a.startswith('--')}
spec = args[0]
N, W = int(opts.get('n', 256)), int(opts.get('w', 1200))
torch.manual_seed(0)
m = kraken_amd.TorchVGSLModel(vgsl=spec, codec=bench_codec()).to('cuda')
m.nn.set_precision(opts.get('precision', 'bf16x3'))
_, c, h, _ = m.input
x = torch.rand(N, c, h, W, device='cuda')
eng = RecognitionEngine(m, device=0, max_batch=N, max_width=W, slots=1)
eng.set_profiling(True)
best = None
for r in range(5):
    eng.submit(x)
    eng.collect()
    t = [(n_, ms) for n_, ms, _ in eng.layer_times()[0]]
    if r:
        best = t if best is None else [(a[0], min(a[1], b[1])) for a, b in zip(best, t)]
print(spec)
print('  alone: total %.3f ms |' % sum(ms for _, ms in best), ' '.join('%s=%.3f' % kv for kv in best), flush=True)
eng.close()
eng = RecognitionEngine(m, device=0, max_batch=N, max_width=W, slots=3)
xs = [torch.rand(N, c, h, W, device='cuda') for _ in range(3)]
for i in range(12):
    if eng.free_slots() == 0:
        eng.collect()
    eng.submit(xs[i % 3])
while eng.free_slots() < 3:
    eng.collect()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 40
for i in range(K):
    if eng.free_slots() == 0:
        m.codec.decode_strings(eng.collect()[0])
    eng.submit(xs[i % 3])
while eng.free_slots() < 3:
    m.codec.decode_strings(eng.collect()[0])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('  pipelined: %.1f lines/s, %.3f ms/step (precision %s)' % (N * K / dt, 1e3 * dt / K, {0: 'f32', 1: 'bf16', 2: 'bf16x3'}.get(m.nn.precision, m.nn.precision)), flush=True)
